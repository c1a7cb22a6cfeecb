// Implicit-GEMM convolution, second generation (the default path of lt_conv_fwd).
//
// Same GEMM view, tile shapes, swizzled 128-byte LDS rows and MFMA fragment scheme as conv_igemm.hip;
// what changes is how bytes move:
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write pass.  The DMA
//     writes wave-linearly (base + lane*16), so the XOR swizzle is applied to the SOURCE: lane L of a
//     wave-instruction owns physical slot L&7 of row L>>3 and fetches the logical 16-byte K vector
//     (L&7) ^ ((row>>1)&7) of that row.  Out-of-image taps (zero padding) fetch from a 16-byte zero page.
//   * tile k+1 is in flight while tile k feeds the MFMAs (2 LDS stages, one barrier per K step, 2
//     workgroups per CU so one block's DMA wait overlaps the other's MFMAs).
//   * epilogue: accumulators -> per-wave LDS sub-tile -> every lane owns 16 contiguous output bytes of one
//     pixel: folded BN / bias on 4-8 channels at a time, 16-byte residual loads, 16-byte stores (a 64-channel
//     row segment = one contiguous 128/256-byte run).  The v1 kernel stored 2-4 bytes per lane.
//   * fp32 mode flushes the MFMA accumulators into fp64 registers every 2 K steps (64 products), which
//     removes the long fp32 accumulation chain from the parity path (K up to 10976 in the 7^3 layer).
#include "conv_common.h"

using namespace lt;

namespace {

__device__ uint4 g_zero_page[2];  // source of out-of-image taps

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16(const void* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <int VECO> struct OutVec;  // VECO output elements <-> 16 bytes
template <> struct OutVec<4> {      // fp32 out
    static __device__ __forceinline__ void ld_res(const float* p, float (&f)[4]) {
        const float4 v = *(const float4*)p;
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&f)[4]) { *(float4*)p = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct OutVec<8> {      // bf16 out
    static __device__ __forceinline__ void ld_res(const bf16_t* p, float (&f)[8]) {
        const uint4 v = *(const uint4*)p;
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(u[i] << 16); f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&f)[8]) {
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = (unsigned)f32_to_bf16(f[2 * i]) | ((unsigned)f32_to_bf16(f[2 * i + 1]) << 16);
        *(uint4*)p = make_uint4(u[0], u[1], u[2], u[3]);
    }
};

__device__ __forceinline__ float epi_act(float v, bool relu_pre, bool has_res, float r, bool relu_post, bool sigm) {
    if (relu_pre) v = fmaxf(v, 0.f);
    if (has_res) v += r;
    if (relu_post) v = fmaxf(v, 0.f);
    if (sigm) v = 1.f / (1.f + expf(-v));
    return v;
}

template <typename T, int BM, int BN, int WM, int WN, int MF>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(const ConvArgs a) {
    constexpr bool ACC64 = sizeof(T) == 4;  // fp32 = parity mode
    constexpr int VEC = elt<T>::vec;
    constexpr int BK = 8 * VEC;
    constexpr int A_IT = BM / 32;                  // DMA instructions per thread for the A tile
    constexpr int B_VECS = BN * 8;
    constexpr int B_IT = (B_VECS + 255) / 256;
    constexpr int SM = WM / MF, SN = WN / MF;
    constexpr int WAVES_N = BN / WN;
    constexpr int G = (MF == 32) ? 4 : 2;
    constexpr int NACC = (MF == 32) ? 16 : 4;
    constexpr int STAGE = (BM + BN) * ROW_BYTES;   // bytes per pipeline stage
    constexpr int EP_LD = WN + 4;                  // padded fp32 row of the per-wave epilogue tile
    constexpr int EP_WAVE = WM * EP_LD * 4;        // bytes
    constexpr int REGION = (2 * STAGE > 4 * EP_WAVE) ? 2 * STAGE : 4 * EP_WAVE;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves");
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_rowpix = (int*)(smem + REGION);         // [BM]
    int4* s_taps = (int4*)(s_rowpix + BM);         // [ntaps]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
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

    // DMA ownership: thread t fills physical slot t&7 of rows (t>>3) + 32*i; the logical K vector it must fetch is
    // v = (t&7) ^ ((row>>1)&7), the same for all its rows (32*i does not change (row>>1)&7)
    const int v = (t & 7) ^ ((t >> 4) & 7);
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
            id0[i] = -(1 << 24);
            ih0[i] = iw0[i] = baseC[i] = 0;
        }
    }
    const T* wrow[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) wrow[j] = w + (size_t)(n0 + ((t + 256 * j) >> 3)) * a.k_pad + v * VEC;
    __syncthreads();

    const int nk = a.k_pad / BK;
    // wave-uniform LDS destinations: rows 8*wave + 32*i (A) / 8*wave + 32*j (B) of the stage
    auto stage = [&](int ks, int buf) {
        unsigned char* sA = smem + buf * STAGE;
        unsigned char* sB = sA + BM * ROW_BYTES;
        const int kel = ks * BK + v * VEC;
        const int tap = kel >> a.log2Cin;
        const int c = kel & (a.Cin - 1);
        int4 tp = make_int4(-(1 << 24), 0, 0, 0);
        if (tap < ph.ntaps) tp = s_taps[tap];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int id = id0[i] + tp.x, ih = ih0[i] + tp.y, iw = iw0[i] + tp.z;
            const bool ok = ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            const void* src = ok ? (const void*)(x + (baseC[i] + tp.w + c)) : (const void*)g_zero_page;
            dma16(src, sA + (8 * wave + 32 * i) * ROW_BYTES);
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            if (B_VECS >= 256 * (j + 1) || 8 * wave + 32 * j < BN)   // wave-uniform: whole 8-row groups
                dma16(wrow[j] + ks * BK, sB + (8 * wave + 32 * j) * ROW_BYTES);
        }
    };

    const int frow = lane & (MF - 1);
    const int fsw = (frow >> 1) & 7;
    int foff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int vec = (MF == 32) ? (2 * g + (lane >> 5)) : ((lane >> 4) + 4 * g);
        foff[g] = frow * ROW_BYTES + ((vec ^ fsw) << 4);
    }
    const int a_base = (wm * WM) * ROW_BYTES;
    const int b_base = BM * ROW_BYTES + (wn * WN) * ROW_BYTES;

    acc_t acc[SM][SN];
    double dacc[ACC64 ? SM : 1][ACC64 ? SN : 1][ACC64 ? NACC : 1];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                acc[i][j][e] = 0.f;
                if (ACC64) dacc[i][j][e] = 0.0;
            }

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) stage(ks + 1, buf ^ 1);
        const unsigned char* pa = smem + buf * STAGE + a_base;
        const unsigned char* pb = smem + buf * STAGE + b_base;
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
        if (ACC64 && ((ks & 1) == 1 || ks + 1 == nk)) {
#pragma unroll
            for (int i = 0; i < SM; ++i)
#pragma unroll
                for (int j = 0; j < SN; ++j)
#pragma unroll
                    for (int e = 0; e < NACC; ++e) {
                        dacc[i][j][e] += (double)acc[i][j][e];
                        acc[i][j][e] = 0.f;
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: accumulators -> this wave's LDS sub-tile (fp32, padded rows) -> 16-byte vectors ----------------
    float* ep = (float*)(smem + wave * EP_WAVE);
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                const int r = i * MF + ((MF == 32) ? ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) : ((lane >> 4) * 4 + e));
                const int cc = j * MF + (lane & (MF - 1));
                ep[r * EP_LD + cc] = ACC64 ? (float)dacc[i][j][e] : acc[i][j][e];
            }
    __syncthreads();

    const bool relu_pre = a.flags & LT_EPI_RELU_PRE, relu_post = a.flags & LT_EPI_RELU_POST, sigm = a.flags & LT_EPI_SIGMOID;
    const bool store_f32 = (a.flags & LT_EPI_STORE_F32) != 0 || sizeof(T) == 4;
    const bool has_res = a.res != nullptr;
    const int col0 = n0 + wn * WN;                 // first output channel of this wave's sub-tile
    const int veco = store_f32 ? 4 : 8;
    const bool vec_ok = (a.Cout % veco == 0) && (a.ldc % veco == 0);
    if (vec_ok) {
        // lanes_per_row = WN/veco; rows_per_pass = 64/lanes_per_row
        const int lpr = WN / veco;
        const int rpp = 64 / lpr;
        const int cq = (lane % lpr) * veco;        // channel offset inside the sub-tile
        const int col = col0 + cq;
        if (col < a.Cout) {
            float sc[8], sf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = (e < veco && a.scale) ? a.scale[col + e] : 1.f;
                sf[e] = (e < veco && a.shift) ? a.shift[col + e] : 0.f;
            }
            for (int r = lane / lpr; r < WM; r += rpp) {
                const int pix = s_rowpix[wm * WM + r];
                if (pix < 0) continue;
                const size_t off = (size_t)pix * a.ldc + col;
                const float* src = ep + r * EP_LD + cq;
                if (store_f32) {
                    float vv[4], rr[4] = {0.f, 0.f, 0.f, 0.f};
                    const float4 q = *(const float4*)src;
                    vv[0] = q.x; vv[1] = q.y; vv[2] = q.z; vv[3] = q.w;
                    if (has_res) {
                        if (sizeof(T) == 4) OutVec<4>::ld_res((const float*)a.res + off, rr);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) rr[e] = bf16_to_f32(((const bf16_t*)a.res)[off + e]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) vv[e] = epi_act(vv[e] * sc[e] + sf[e], relu_pre, has_res, rr[e], relu_post, sigm);
                    OutVec<4>::st((float*)a.y + off, vv);
                } else {
                    float vv[8], rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const float4 q0 = *(const float4*)src, q1 = *(const float4*)(src + 4);
                    vv[0] = q0.x; vv[1] = q0.y; vv[2] = q0.z; vv[3] = q0.w; vv[4] = q1.x; vv[5] = q1.y; vv[6] = q1.z; vv[7] = q1.w;
                    if (has_res) OutVec<8>::ld_res((const bf16_t*)a.res + off, rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) vv[e] = epi_act(vv[e] * sc[e] + sf[e], relu_pre, has_res, rr[e], relu_post, sigm);
                    OutVec<8>::st((bf16_t*)a.y + off, vv);
                }
            }
        }
    } else {
        // ragged channel counts (Cout = 17, ...): one element per lane, lanes along channels
        for (int idx = lane; idx < WM * WN; idx += 64) {
            const int r = idx / WN, cc = idx - r * WN;
            const int col = col0 + cc;
            const int pix = s_rowpix[wm * WM + r];
            if (pix < 0 || col >= a.Cout) continue;
            const size_t off = (size_t)pix * a.ldc + col;
            float val = ep[r * EP_LD + cc] * (a.scale ? a.scale[col] : 1.f) + (a.shift ? a.shift[col] : 0.f);
            const float rr = has_res ? elt<T>::ld((const T*)a.res + off) : 0.f;
            val = epi_act(val, relu_pre, has_res, rr, relu_post, sigm);
            if (store_f32) ((float*)a.y)[off] = val;
            else elt<T>::st((T*)a.y + off, val);
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int MF>
int launch2(ConvArgs a, int cout_pad, int nphase, int max_taps, hipStream_t s) {
    LT_REQUIRE(cout_pad % BN == 0, LT_ERR_INVALID, "lt_conv_fwd: cout_pad %d not a multiple of tile N %d", cout_pad, BN);
    a.tiles_n = cout_pad / BN;
    const long long nblk = cdiv(a.M, BM) * a.tiles_n;
    LT_REQUIRE(nblk < (1ll << 31), LT_ERR_INVALID, "lt_conv_fwd: grid too large");
    constexpr int STAGE = (BM + BN) * ROW_BYTES;
    constexpr int EP_WAVE = WM * (WN + 4) * 4;
    constexpr int REGION = (2 * STAGE > 4 * EP_WAVE) ? 2 * STAGE : 4 * EP_WAVE;
    const size_t lds = REGION + BM * sizeof(int) + (size_t)max_taps * sizeof(int4);
    LT_REQUIRE(lds <= 160 * 1024, LT_ERR_UNSUPPORTED, "lt_conv_fwd: tile needs %zu B of LDS", lds);
    auto kern = conv_igemm2_kernel<T, BM, BN, WM, WN, MF>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, nphase), dim3(256), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(v2)");
    return LT_OK;
}

template <typename T>
int dispatch2(const ConvArgs& a, int cout_pad, int nphase, int max_taps, int tile, hipStream_t s) {
    if (tile == LT_TILE_AUTO) {
        if (cout_pad <= 16) tile = LT_TILE2_256x16;
        else if (cout_pad <= 32) tile = LT_TILE2_256x32;
        else if (cout_pad <= 64) tile = LT_TILE2_128x64;
        else tile = LT_TILE2_128x128;
        // small problems: more, smaller workgroups
        if (cout_pad >= 64 && cdiv(a.M, 128) * cdiv(cout_pad, 128) < 192) tile = LT_TILE2_64x64;
    }
    switch (tile) {
        case LT_TILE2_128x128: return launch2<T, 128, 128, 64, 64, 32>(a, cout_pad, nphase, max_taps, s);
        case LT_TILE2_128x64: return launch2<T, 128, 64, 64, 32, 32>(a, cout_pad, nphase, max_taps, s);
        case LT_TILE2_256x32: return launch2<T, 256, 32, 64, 32, 32>(a, cout_pad, nphase, max_taps, s);
        case LT_TILE2_256x16: return launch2<T, 256, 16, 64, 16, 16>(a, cout_pad, nphase, max_taps, s);
        case LT_TILE2_64x64: return launch2<T, 64, 64, 32, 32, 32>(a, cout_pad, nphase, max_taps, s);
        default: break;
    }
    set_error("lt_conv_fwd: unknown tile id %d", tile);
    return LT_ERR_INVALID;
}

}  // namespace

namespace lt {
int conv2_dispatch(int dtype, const ConvArgs& a, int cout_pad, int nphase, int max_taps, int tile, hipStream_t s) {
    if (dtype == LT_F32) return dispatch2<float>(a, cout_pad, nphase, max_taps, tile, s);
    return dispatch2<bf16_t>(a, cout_pad, nphase, max_taps, tile, s);
}
}  // namespace lt
