// Generalised convolution as implicit GEMM on the gfx950 matrix cores.
//
//   rows    M = N*Do*Ho*Wo output points (channels-last, so one row's K slice for one filter tap is a
//             contiguous run of input channels -> 16-byte coalesced gathers, zero filled when a tap
//             falls outside the input)
//   columns   = output channels (weights pre-packed [cout_pad][k_pad], k = tap*Cin + ci)
//   K         = ntaps*Cin, walked in 128-byte steps (32 fp32 / 64 bf16 elements per row)
//
// One kernel serves conv2d/conv3d (1 phase) and stride-2 transposed convs (one phase per output
// parity, blockIdx.y), with the folded-BN / residual / ReLU epilogue of lt_hip.h.
//
// CDNA4 mapping: 256 threads = 4 wave64; global -> VGPR -> LDS staging with the loads for step k+1 in
// flight under the MFMAs of step k (one barrier per step, two LDS buffers); LDS rows are 128 B with
// the 16-byte slot XOR-swizzled by (row>>1)&7 so that both the 8-lane ds_write_b128 groups and the
// 16-lane ds_read_b128 groups are bank-conflict free; MFMA fragments are read as whole 16-byte
// vectors for BOTH dtypes: bf16 feeds one v_mfma_f32_32x32x16_bf16 (16x16x32) per vector pair,
// fp32 feeds four v_mfma_f32_32x32x2_f32 (16x16x4) from the vector's four lanes-worth of k (the k
// order inside a step is permuted identically for A and B, which leaves the dot product unchanged).
#include <stdlib.h>

#include "conv_common.h"

using namespace lt;

namespace {


template <typename T, int BM, int BN, int WM, int WN, int MF>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int VEC = elt<T>::vec;
    constexpr int BK = 8 * VEC;
    constexpr int A_IT = BM / 32;
    constexpr int B_VECS = BN * 8;
    constexpr int B_IT = (B_VECS + 255) / 256;
    constexpr int SM = WM / MF, SN = WN / MF;
    constexpr int WAVES_N = BN / WN;
    constexpr int G = (MF == 32) ? 4 : 2;  // fragment groups per K step
    constexpr int NACC = (MF == 32) ? 16 : 4;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves");
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                          // [2][BM*128]
    unsigned char* sB = smem + 2 * BM * ROW_BYTES;     // [2][BN*128]
    int* s_rowpix = (int*)(sB + 2 * BN * ROW_BYTES);   // [BM] output pixel index or -1
    int4* s_taps = (int4*)(s_rowpix + BM);             // [ntaps]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile_n = blockIdx.x % a.tiles_n;
    const int tile_m = blockIdx.x / a.tiles_n;
    const PhaseArg ph = a.phase[blockIdx.y];
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ x = (const T*)a.x;
    const T* __restrict__ w = (const T*)ph.w;

    for (int i = t; i < ph.ntaps; i += 256) s_taps[i] = ph.taps[i];
    for (int r = t; r < BM; r += 256) {
        int m = m0 + r, pix = -1;
        if (m < a.M) {
            int n, od, oh, ow;
            decode_row(a, m, n, od, oh, ow);
            pix = ((n * a.OD + od * a.osd + ph.ood) * a.OH + oh * a.osh + ph.ooh) * a.OW + ow * a.osw + ph.oow;
        }
        s_rowpix[r] = pix;
    }

    // rows staged by this thread: (t>>3) + 32*i, always 16-byte vector v = t&7 of the 128-byte step
    const int v = t & 7;
    int id0[A_IT], ih0[A_IT], iw0[A_IT], baseC[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + (t >> 3) + 32 * i;
        if (m < a.M) {
            int n, od, oh, ow;
            decode_row(a, m, n, od, oh, ow);
            id0[i] = od * a.sd - a.pd;
            ih0[i] = oh * a.sh - a.ph;
            iw0[i] = ow * a.sw - a.pw;
            baseC[i] = (((n * a.D + id0[i]) * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
        } else {
            id0[i] = -(1 << 24);  // never in range
            ih0[i] = iw0[i] = baseC[i] = 0;
        }
    }
    __syncthreads();

    const int nk = a.k_pad / BK;
    uint4 ra[A_IT], rb[B_IT];

    auto gload = [&](int ks) {
        const int kel = ks * BK + v * VEC;
        const int tap = kel >> a.log2Cin;
        const int c = kel & (a.Cin - 1);
        int4 tp = make_int4(-(1 << 24), 0, 0, 0);
        if (tap < ph.ntaps) tp = s_taps[tap];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int id = id0[i] + tp.x, ih = ih0[i] + tp.y, iw = iw0[i] + tp.z;
            const bool ok = ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            uint4 val = make_uint4(0, 0, 0, 0);
            if (ok) val = *(const uint4*)(x + (baseC[i] + tp.w + c));
            ra[i] = val;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int vid = t + 256 * j;
            if (B_VECS >= 256 * (j + 1) || vid < B_VECS)
                rb[j] = *(const uint4*)(w + (size_t)(n0 + (vid >> 3)) * a.k_pad + ks * BK + v * VEC);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int row = (t >> 3) + 32 * i;
            *(uint4*)(sA + buf * BM * ROW_BYTES + row * ROW_BYTES + ((v ^ ((row >> 1) & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int vid = t + 256 * j;
            if (B_VECS >= 256 * (j + 1) || vid < B_VECS) {
                const int row = vid >> 3;
                *(uint4*)(sB + buf * BN * ROW_BYTES + row * ROW_BYTES + ((v ^ ((row >> 1) & 7)) << 4)) = rb[j];
            }
        }
    };

    // per-lane fragment addresses (swizzle term is constant per lane: tile row bases are multiples of MF)
    const int frow = lane & (MF - 1);
    const int fsw = (frow >> 1) & 7;
    int foff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int vec = (MF == 32) ? (2 * g + (lane >> 5)) : ((lane >> 4) + 4 * g);
        foff[g] = frow * ROW_BYTES + ((vec ^ fsw) << 4);
    }
    const int a_base = (wm * WM) * ROW_BYTES;
    const int b_base = (wn * WN) * ROW_BYTES;

    acc_t acc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int e = 0; e < NACC; ++e) acc[i][j][e] = 0.f;

    gload(0);
    sstore(0);
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) gload(ks + 1);
        const unsigned char* pa = sA + buf * BM * ROW_BYTES + a_base;
        const unsigned char* pb = sB + buf * BN * ROW_BYTES + b_base;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            V16 fa[SM], fb[SN];
#pragma unroll
            for (int i = 0; i < SM; ++i) fa[i].u = *(const uint4*)(pa + i * MF * ROW_BYTES + foff[g]);
#pragma unroll
            for (int j = 0; j < SN; ++j) fb[j].u = *(const uint4*)(pb + j * MF * ROW_BYTES + foff[g]);
#pragma unroll
            for (int i = 0; i < SM; ++i)
#pragma unroll
                for (int j = 0; j < SN; ++j) Mma<T, MF>::run(acc[i][j], fa[i], fb[j]);
        }
        if (ks + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: folded BN / bias, residual, ReLU; lanes of a row segment hold consecutive channels
    const bool relu_pre = a.flags & LT_EPI_RELU_PRE, relu_post = a.flags & LT_EPI_RELU_POST;
    const bool store_f32 = (a.flags & LT_EPI_STORE_F32) != 0, sigm = (a.flags & LT_EPI_SIGMOID) != 0;
    T* __restrict__ y = (T*)a.y;
    const T* __restrict__ res = (const T*)a.res;
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int col = n0 + wn * WN + j * MF + (lane & (MF - 1));
        const float bi = a.bias ? a.bias[col] : 0.f;
        const float sc = a.scale ? a.scale[col] : 1.f;
        const float sf = a.shift ? a.shift[col] : 0.f;
        const bool col_ok = col < a.Cout;
#pragma unroll
        for (int i = 0; i < SM; ++i) {
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                const int r = wm * WM + i * MF + ((MF == 32) ? ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) : ((lane >> 4) * 4 + e));
                const int pix = s_rowpix[r];
                if (pix >= 0 && col_ok) {
                    const size_t off = (size_t)pix * a.ldc + col;
                    float val = (acc[i][j][e] + bi) * sc + sf;
                    if (relu_pre) val = fmaxf(val, 0.f);
                    if (res) val += (sizeof(T) == 2 && (a.flags & LT_EPI_RES_F32)) ? ((const float*)a.res)[off] : elt<T>::ld(res + off);
                    if (relu_post) val = fmaxf(val, 0.f);
                    if (sigm) val = 1.f / (1.f + expf(-val));
                    if (store_f32) ((float*)a.y)[off] = val;
                    else elt<T>::st(y + off, val);
                }
            }
        }
    }
}

// Scalar fp32 cross-check kernel (LT_TILE_DIRECT): one thread per (row, channel).  Debug only.
template <typename T>
__global__ void conv_direct_kernel(const ConvArgs a) {
    const PhaseArg ph = a.phase[blockIdx.y];
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int co = (int)(gid % a.Cout);
    const long long m = gid / a.Cout;
    if (m >= a.M) return;
    int n, od, oh, ow;
    decode_row(a, (int)m, n, od, oh, ow);
    const T* x = (const T*)a.x;
    const T* w = (const T*)ph.w + (size_t)co * a.k_pad;
    float acc = 0.f;
    for (int tp = 0; tp < ph.ntaps; ++tp) {
        const int4 tt = ph.taps[tp];
        const int id = od * a.sd - a.pd + tt.x, ih = oh * a.sh - a.ph + tt.y, iw = ow * a.sw - a.pw + tt.z;
        if ((unsigned)id >= (unsigned)a.D || (unsigned)ih >= (unsigned)a.H || (unsigned)iw >= (unsigned)a.W) continue;
        const T* px = x + ((((size_t)n * a.D + id) * a.H + ih) * a.W + iw) * a.Cin;
        for (int c = 0; c < a.Cin; ++c) acc = fmaf(elt<T>::ld(px + c), elt<T>::ld(w + tp * a.Cin + c), acc);
    }
    const size_t pix = (((size_t)n * a.OD + od * a.osd + ph.ood) * a.OH + oh * a.osh + ph.ooh) * a.OW + ow * a.osw + ph.oow;
    const size_t off = pix * a.ldc + co;
    float val = (acc + (a.bias ? a.bias[co] : 0.f)) * (a.scale ? a.scale[co] : 1.f) + (a.shift ? a.shift[co] : 0.f);
    if (a.flags & LT_EPI_RELU_PRE) val = fmaxf(val, 0.f);
    if (a.res) val += (sizeof(T) == 2 && (a.flags & LT_EPI_RES_F32)) ? ((const float*)a.res)[off] : elt<T>::ld((const T*)a.res + off);
    if (a.flags & LT_EPI_RELU_POST) val = fmaxf(val, 0.f);
    if (a.flags & LT_EPI_SIGMOID) val = 1.f / (1.f + expf(-val));
    if (a.flags & LT_EPI_STORE_F32) ((float*)a.y)[off] = val;
    else elt<T>::st((T*)a.y + off, val);
}

template <typename T, int BM, int BN, int WM, int WN, int MF>
int launch(ConvArgs a, int cout_pad, int nphase, int max_taps, hipStream_t s) {
    LT_REQUIRE(cout_pad % BN == 0, LT_ERR_INVALID, "lt_conv_fwd: cout_pad %d not a multiple of tile N %d", cout_pad, BN);
    a.tiles_n = cout_pad / BN;
    const long long tiles_m = cdiv(a.M, BM);
    const long long nblk = tiles_m * a.tiles_n;
    LT_REQUIRE(nblk < (1ll << 31), LT_ERR_INVALID, "lt_conv_fwd: grid too large");
    const size_t lds = 2 * (BM + BN) * ROW_BYTES + BM * sizeof(int) + (size_t)max_taps * sizeof(int4);
    auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, MF>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, nphase), dim3(256), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd");
    return LT_OK;
}

template <typename T>
int dispatch(const ConvArgs& a, int cout_pad, int nphase, int max_taps, int tile, hipStream_t s) {
    if (tile == LT_TILE_DIRECT) {
        const long long total = (long long)a.M * a.Cout;
        hipLaunchKernelGGL(conv_direct_kernel<T>, dim3((unsigned)cdiv(total, 256), nphase), dim3(256), 0, s, a);
        LT_CHECK_LAUNCH("lt_conv_fwd(direct)");
        return LT_OK;
    }
    static const bool force_v1 = getenv("LT_CONV_V1") != nullptr;   // A/B switches for profiling sessions
    static const bool no_halo = getenv("LT_CONV_NO_HALO") != nullptr;
    static const bool no_respf = getenv("LT_CONV_NO_RESPF") != nullptr;
    if (no_respf) const_cast<ConvArgs&>(a).flags |= LT_EPI_NO_RES_PREFETCH;
    static const bool no_xcd = getenv("LT_CONV_NO_XCD") != nullptr;
    if (no_xcd) const_cast<ConvArgs&>(a).flags |= LT_EPI_NO_XCD_REMAP;
    if ((tile == LT_TILE_AUTO && !force_v1 && !no_halo) || tile == LT_TILE_HALO) {
        const int rc = conv3d_halo_try(sizeof(T) == 4 ? LT_F32 : LT_BF16, a, cout_pad, nphase, tile == LT_TILE_HALO, s);
        if (rc == 1) return LT_OK;
        if (rc < 0) return rc;
        if (tile == LT_TILE_HALO) {
            set_error("lt_conv_fwd: LT_TILE_HALO requested but the problem is not a supported stride-1 3^3/7^3 conv3d");
            return LT_ERR_UNSUPPORTED;
        }
    }
    if (tile == LT_TILE_AUTO && !force_v1) {             // narrow single-tap layers (V2V skip convs, 2x2x2 deconvs): streaming kernel
        const int rc = conv_pw_try(sizeof(T) == 4 ? LT_F32 : LT_BF16, a, cout_pad, nphase, s);
        if (rc < 0) return rc;
        if (rc == 1) return LT_OK;
    }
    if (tile == LT_TILE_AUTO && !force_v1 && sizeof(T) == 2 && a.phase[0].wfrag_t) {   // ResNet layer3's 3x3 256 -> 256 with its weights in layout 2: 2D halo kernel
        const int rc = conv2d_halo_try(LT_BF16, a, cout_pad, nphase, s);
        if (rc < 0) return rc;
        if (rc == 1) return LT_OK;
        // a 2D layer whose weights were packed in layout 2 has NO layout-1 / -3 fragments: the generic tiles would still compute it (from the plain
        // [cout][k] weights) at a fraction of the rate, silently.  The plan builder's gate mirrors conv2d_halo_try's predicate; a mismatch is a bug: say so
        // (ADVICE r5).  (3D layers carry layout-2 fragments for conv3d_halo_wreg_kernel, whose siblings above take them when it declines -- a 3^3 layer on a
        // volume of ONE voxel plane, V2V's deepest level in small test volumes, has D == 1 too: the 2D signature is 256 -> 256 with 9 or 4 taps.)
        if (a.D == 1 && a.Do == 1 && a.OD == 1 && a.Cin == 256 && cout_pad == 256 && (a.phase[0].ntaps == 9 || a.phase[0].ntaps == 4)) {
            set_error("lt_conv_fwd: 2D layer with weight_frag_layout 2 (conv2d_halo_kernel) but the halo kernel does not cover it: Cin %d, Cout %d, ldc %d, "
                      "%d x %d -> %d x %d, stride %d, pad %d, %d phase(s), flags 0x%x%s", a.Cin, a.Cout, a.ldc, a.H, a.W, a.OH, a.OW, a.sh, a.ph, nphase, a.flags,
                      a.res ? ", residual" : "");
            return LT_ERR_UNSUPPORTED;
        }
    }
    static const bool no_v3 = getenv("LT_CONV_NO_V3") != nullptr;   // A/B switch
    if ((tile == LT_TILE_AUTO && !force_v1 && !no_v3) || tile == LT_TILE3_288) {
        const int rc = conv3_try(sizeof(T) == 4 ? LT_F32 : LT_BF16, a, cout_pad, nphase, max_taps, tile == LT_TILE3_288, s);
        if (rc < 0) return rc;
        if (rc == 1) return LT_OK;
        if (tile == LT_TILE3_288) {
            set_error("lt_conv_fwd: LT_TILE3_288 requested but the problem is not supported by the 288-row kernel");
            return LT_ERR_UNSUPPORTED;
        }
    }
    if ((tile == LT_TILE_AUTO && !force_v1) || (tile >= LT_TILE2_128x128 && tile <= LT_TILE2_64x64))
        return conv2_dispatch(sizeof(T) == 4 ? LT_F32 : LT_BF16, a, cout_pad, nphase, max_taps, tile, s);
    if (tile == LT_TILE_AUTO) {
        if (cout_pad <= 16) tile = LT_TILE_256x16;
        else if (cout_pad <= 32) tile = LT_TILE_256x32;
        else if (cout_pad <= 64) tile = LT_TILE_128x64;
        else tile = LT_TILE_128x128;
        if (cout_pad >= 64 && cdiv(a.M, 128) * cdiv(cout_pad, 128) < 192) tile = LT_TILE_64x64;
    }
    switch (tile) {
        case LT_TILE_128x128: return launch<T, 128, 128, 64, 64, 32>(a, cout_pad, nphase, max_taps, s);
        case LT_TILE_128x64: return launch<T, 128, 64, 64, 32, 32>(a, cout_pad, nphase, max_taps, s);
        case LT_TILE_256x32: return launch<T, 256, 32, 64, 32, 32>(a, cout_pad, nphase, max_taps, s);
        case LT_TILE_256x16: return launch<T, 256, 16, 64, 16, 16>(a, cout_pad, nphase, max_taps, s);
        case LT_TILE_64x64: return launch<T, 64, 64, 32, 32, 32>(a, cout_pad, nphase, max_taps, s);
        default: break;
    }
    set_error("lt_conv_fwd: unknown tile id %d", tile);
    return LT_ERR_INVALID;
}

}  // namespace

// Samples per launch of a convolution whose tensors have ``per_sample`` elements per sample (the largest of input, output, second source): the kernels index
// one LAUNCH with 32-bit element offsets, the entry point walks larger batches in sample chunks with 64-bit base pointers (round 6).  All of N when it
// fits; else equal-sized chunks, multiples of 8 samples where possible (the halo / column-walk kernels pin samples to XCDs).  0: one sample alone is too large.
extern "C" int32_t lt_conv_chunk_samples(int32_t N, int64_t per_sample) {
    const long long lim = (1ll << 31) - 1;
    if (N < 1 || per_sample < 1 || per_sample > lim) return 0;
    if ((long long)N * per_sample <= lim) return N;
    long long nmax = lim / per_sample;
    if (nmax >= 8) nmax &= ~7ll;
    const long long chunks = (N + nmax - 1) / nmax;
    long long nc = (N + chunks - 1) / chunks;
    if (nc > 8 && (nc & 7) && ((nc + 7) & ~7ll) <= nmax) nc = (nc + 7) & ~7ll;
    return (int32_t)nc;
}

static int conv_fwd_impl(const lt_conv_desc* d, const void* x, const float* bias, const float* scale, const float* shift,
                         const void* residual, const lt_conv_skip* skip, const lt_conv_cat2* cat2, void* y, void* stream) {
    LT_REQUIRE(d && x && y, LT_ERR_INVALID, "lt_conv_fwd: null argument");
    LT_REQUIRE(d->dtype == LT_F32 || d->dtype == LT_BF16 || d->dtype == LT_FP8, LT_ERR_INVALID, "lt_conv_fwd: bad dtype %d", d->dtype);
    {
        // ---- batches beyond 2^31 elements per tensor (BASELINE config 4 at 32 samples: 32 x 128^3 x 32 channels = 2^31 exactly): sample chunks, each a launch
        // of the same descriptor over fewer samples with every per-sample pointer advanced in 64 bits; results identical (samples are independent units)
        const long long in_s = (long long)d->D * d->H * d->W * d->Cin, out_s = (long long)d->OD * d->OH * d->OW * d->ldc;
        const long long x2_s = cat2 ? (long long)cat2->H * cat2->W * cat2->cin : 0;
        long long big = in_s > out_s ? in_s : out_s;
        if (x2_s > big) big = x2_s;
        if (d->N > 1 && big > 0 && (long long)d->N * big >= (1ll << 31)) {
            const int nc = lt_conv_chunk_samples(d->N, big);
            LT_REQUIRE(nc >= 1, LT_ERR_UNSUPPORTED, "lt_conv_fwd: one sample has %lld elements: beyond 32-bit element offsets", big);
            const size_t ex = d->dtype == LT_F32 ? 4 : d->dtype == LT_BF16 ? 2 : 1;
            const size_t ey = (d->flags & LT_EPI_STORE_F32) ? 4 : (d->dtype == LT_F32 ? 4 : 2);
            const size_t er = (d->flags & LT_EPI_RES_F32) ? 4 : (d->dtype == LT_F32 ? 4 : 2);
            for (int n0 = 0; n0 < d->N; n0 += nc) {
                lt_conv_desc dd = *d;
                dd.N = d->N - n0 < nc ? d->N - n0 : nc;
                lt_conv_skip sk; lt_conv_cat2 c2;
                if (skip) { sk = *skip; sk.x = (const char*)skip->x + (size_t)n0 * d->D * d->H * d->W * skip->cin * 2; }
                if (cat2) { c2 = *cat2; c2.x = (const char*)cat2->x + (size_t)n0 * x2_s * 2; }
                const int rc = conv_fwd_impl(&dd, (const char*)x + (size_t)n0 * in_s * ex, bias, scale, shift,
                                             residual ? (const char*)residual + (size_t)n0 * out_s * er : nullptr, skip ? &sk : nullptr, cat2 ? &c2 : nullptr,
                                             (char*)y + (size_t)n0 * out_s * ey, stream);
                if (rc != LT_OK) return rc;
            }
            return LT_OK;
        }
    }
    const int vec = d->dtype == LT_F32 ? 4 : d->dtype == LT_BF16 ? 8 : 16;
    LT_REQUIRE(d->dtype != LT_FP8 || (!(d->flags & LT_EPI_SIGMOID) &&
                                      (((d->flags & LT_EPI_STORE_F32) && (!residual || (d->flags & LT_EPI_RES_F32))) ||
                                       (!(d->flags & LT_EPI_STORE_F32) && d->Cout % 8 == 0 && d->ldc % 8 == 0))),
               LT_ERR_UNSUPPORTED, "lt_conv_fwd: an fp8 convolution stores fp32 (LT_EPI_STORE_F32, fp32 residual: LT_EPI_RES_F32) or, with Cout %% 8 == 0, bf16 (bf16 residual)");
    LT_REQUIRE(d->dtype != LT_FP8 || d->tile == LT_TILE_AUTO || d->tile == LT_TILE_HALO || (d->tile >= LT_TILE2_128x128 && d->tile <= LT_TILE2_64x64), LT_ERR_UNSUPPORTED,
               "lt_conv_fwd: fp8 convolutions run on the generic implicit-GEMM tiles and the halo kernel only");
    const int l2 = ilog2_exact(d->Cin);
    LT_REQUIRE(l2 >= 0 && d->Cin >= vec, LT_ERR_UNSUPPORTED,
               "lt_conv_fwd: Cin=%d must be a power of two >= %d (pad the channel dimension)", d->Cin, vec);
    LT_REQUIRE(d->nphase >= 1 && d->nphase <= LT_CONV_MAX_PHASES, LT_ERR_INVALID, "lt_conv_fwd: nphase=%d", d->nphase);
    LT_REQUIRE(d->k_pad > 0 && d->k_pad % (8 * vec) == 0, LT_ERR_INVALID, "lt_conv_fwd: k_pad=%d not a multiple of %d", d->k_pad, 8 * vec);
    LT_REQUIRE(d->Cout >= 1 && d->ldc >= d->Cout && d->cout_pad >= d->Cout, LT_ERR_INVALID, "lt_conv_fwd: Cout/ldc/cout_pad");
    LT_REQUIRE(!(d->flags & LT_EPI_RES_F32) || ((d->flags & LT_EPI_STORE_F32) && d->dtype != LT_F32), LT_ERR_INVALID,
               "lt_conv_fwd: LT_EPI_RES_F32 goes with LT_EPI_STORE_F32 on a bf16 / fp8 convolution");
    const long long M = (long long)d->N * d->Do * d->Ho * d->Wo;
    const long long in_elems = (long long)d->N * d->D * d->H * d->W * d->Cin;
    const long long out_pix = (long long)d->N * d->OD * d->OH * d->OW;
    LT_REQUIRE(M > 0 && M < (1ll << 31) && in_elems < (1ll << 31) && out_pix < (1ll << 31), LT_ERR_UNSUPPORTED,
               "lt_conv_fwd: tensor too large for 32-bit indexing (M=%lld, in=%lld)", M, in_elems);
    ConvArgs a;
    a.x = x; a.y = y; a.res = residual; a.bias = bias; a.scale = scale; a.shift = shift;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.log2Cin = l2;
    a.Do = d->Do; a.Ho = d->Ho; a.Wo = d->Wo;
    a.sd = d->stride[0]; a.sh = d->stride[1]; a.sw = d->stride[2];
    a.pd = d->pad[0]; a.ph = d->pad[1]; a.pw = d->pad[2];
    a.OD = d->OD; a.OH = d->OH; a.OW = d->OW;
    a.osd = d->out_stride[0]; a.osh = d->out_stride[1]; a.osw = d->out_stride[2];
    a.Cout = d->Cout; a.ldc = d->ldc; a.k_pad = d->k_pad; a.flags = d->flags; a.M = (int)M; a.tiles_n = 1; a.stages = d->stages;
    a.skip_x = skip ? skip->x : nullptr; a.skip_w = skip ? skip->weight_frag : nullptr;
    a.x2 = cat2 ? cat2->x : nullptr; a.Cin2 = cat2 ? cat2->cin : 0; a.H2 = cat2 ? cat2->H : 0; a.W2 = cat2 ? cat2->W : 0; a.s2 = cat2 ? cat2->stride : 0;
    int max_taps = 0;
    for (int p = 0; p < d->nphase; ++p) {
        const lt_conv_phase& ph = d->phase[p];
        LT_REQUIRE(ph.weight && ph.taps && ph.ntaps >= 1, LT_ERR_INVALID, "lt_conv_fwd: phase %d incomplete", p);
        LT_REQUIRE((long long)ph.ntaps * d->Cin <= d->k_pad, LT_ERR_INVALID, "lt_conv_fwd: phase %d: ntaps*Cin > k_pad", p);
        a.phase[p].w = ph.weight; a.phase[p].wfrag = ph.weight_frag_layout == 1 ? ph.weight_frag : nullptr;
        a.phase[p].wfrag_t = ph.weight_frag_layout == 2 ? ph.weight_frag : nullptr;
        a.phase[p].wfrag32 = ph.weight_frag_layout == 3 ? ph.weight_frag : nullptr; a.phase[p].taps = (const int4*)ph.taps; a.phase[p].ntaps = ph.ntaps;
        a.phase[p].ood = ph.out_off[0]; a.phase[p].ooh = ph.out_off[1]; a.phase[p].oow = ph.out_off[2];
        if (ph.ntaps > max_taps) max_taps = ph.ntaps;
    }
    LT_REQUIRE(max_taps <= 2048, LT_ERR_UNSUPPORTED, "lt_conv_fwd: too many taps (%d)", max_taps);
    hipStream_t s = (hipStream_t)stream;
    if (skip) {   // the computed residual exists in ONE kernel: fail loudly everywhere else
        LT_REQUIRE(skip->x && skip->weight_frag && skip->cin == 16 && d->dtype == LT_BF16 && !residual && d->Cin == 32 && d->Cout == 32 && d->cout_pad == 32,
                   LT_ERR_UNSUPPORTED, "lt_conv_skip_fwd: bf16 3x3x3 32 -> 32 with a 16-channel skip tensor only (cin %d, Cin %d, Cout %d)", skip->cin, d->Cin, d->Cout);
        const int rc = conv3d_halo_try(LT_BF16, a, d->cout_pad, d->nphase, true, s);
        LT_REQUIRE(rc == 1, rc < 0 ? rc : LT_ERR_UNSUPPORTED, "lt_conv_skip_fwd: this shape is not covered by the column-walk halo kernel (N %d, %d x %d x %d)",
                   d->N, d->D, d->H, d->W);
        return LT_OK;
    }
    if (cat2) {   // one kernel (conv_igemm7, pointwise over two sources): fail loudly everywhere else
        const bool pointwise = d->nphase == 1 && d->phase[0].ntaps == 1 && d->D == 1 && d->Do == 1 && d->OD == 1 && d->H == d->Ho && d->W == d->Wo &&
                               d->OH == d->Ho && d->OW == d->Wo && d->stride[1] == 1 && d->stride[2] == 1 && d->pad[1] == 0 && d->pad[2] == 0 &&
                               d->out_stride[1] == 1 && d->out_stride[2] == 1 && !d->phase[0].out_off[1] && !d->phase[0].out_off[2];
        LT_REQUIRE(cat2->x && d->dtype == LT_BF16 && pointwise && d->Cin % 32 == 0 && cat2->cin % 32 == 0 && d->k_pad == d->Cin + cat2->cin && d->k_pad % 64 == 0 &&
                       d->cout_pad % 256 == 0 && d->phase[0].weight_frag_layout == 3 && d->phase[0].weight_frag && (cat2->stride == 1 || cat2->stride == 2) &&
                       cat2->H == d->Ho * cat2->stride && cat2->W == d->Wo * cat2->stride && !(d->flags & (LT_EPI_STORE_F32 | LT_EPI_SIGMOID)) &&
                       (long long)d->N * cat2->H * cat2->W * cat2->cin < (1ll << 31),
                   LT_ERR_UNSUPPORTED, "lt_conv_cat2_fwd: bf16 1x1 convolution, Cin %d + %d (multiples of 32, k_pad %d their sum), cout_pad %d (%% 256), fragment layout 3, "
                   "second map %d x %d = stride %d x the output's", d->Cin, cat2->cin, d->k_pad, d->cout_pad, cat2->H, cat2->W, cat2->stride);
        const int rc = conv7_try(a, d->cout_pad, max_taps, true, s);
        LT_REQUIRE(rc == 1, rc < 0 ? rc : LT_ERR_UNSUPPORTED, "lt_conv_cat2_fwd: not covered by the 288 x 256 pointwise kernel");
        return LT_OK;
    }
    if (d->dtype == LT_F32) return dispatch<float>(a, d->cout_pad, d->nphase, max_taps, d->tile, s);
    if (d->dtype == LT_FP8) {
        // the 3^3 layers of the 64^3 / 32^3 levels: input halo in LDS (conv3d_halo.hip); bf16 stores only
        if ((d->tile == LT_TILE_AUTO || d->tile == LT_TILE_HALO) && !(d->flags & LT_EPI_STORE_F32)) {
            const int rc = conv3d_halo_try(LT_FP8, a, d->cout_pad, d->nphase, d->tile == LT_TILE_HALO, s);
            if (rc == 1) return LT_OK;
            if (rc < 0) return rc;
        }
        LT_REQUIRE(d->tile != LT_TILE_HALO, LT_ERR_UNSUPPORTED, "lt_conv_fwd: LT_TILE_HALO requested but this fp8 problem has no halo kernel");
        return conv2_dispatch(LT_FP8, a, d->cout_pad, d->nphase, max_taps, d->tile, s);
    }
    return dispatch<bf16_t>(a, d->cout_pad, d->nphase, max_taps, d->tile, s);
}

extern "C" int lt_conv_fwd(const lt_conv_desc* d, const void* x, const float* bias, const float* scale, const float* shift,
                           const void* residual, void* y, void* stream) {
    return conv_fwd_impl(d, x, bias, scale, shift, residual, nullptr, nullptr, y, stream);
}

extern "C" int lt_conv_cat2_fwd(const lt_conv_desc* d, const void* x, const lt_conv_cat2* second, const float* bias, const float* scale, const float* shift,
                                const void* residual, void* y, void* stream) {
    LT_REQUIRE(second, LT_ERR_INVALID, "lt_conv_cat2_fwd: null second-source descriptor");
    return conv_fwd_impl(d, x, bias, scale, shift, residual, nullptr, second, y, stream);
}

extern "C" int lt_conv_skip_fwd(const lt_conv_desc* d, const void* x, const float* bias, const float* scale, const float* shift,
                                const lt_conv_skip* skip, void* y, void* stream) {
    LT_REQUIRE(skip, LT_ERR_INVALID, "lt_conv_skip_fwd: null skip descriptor");
    return conv_fwd_impl(d, x, bias, scale, shift, nullptr, skip, nullptr, y, stream);
}

extern "C" int lt_conv_cout_pad(int32_t cout) {
    if (cout <= 16) return 16;
    if (cout <= 32) return 32;
    if (cout <= 64) return 64;
    return (cout + 127) / 128 * 128;
}
