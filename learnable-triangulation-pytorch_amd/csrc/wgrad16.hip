// Weight gradients of the mixed-precision training step on the bf16 MFMA (BASELINE config 5; reference: autograd of the convolutions in
// pose_resnet.py / v2v.py inside train.py:217-243's backward).
//
//   dW[co][tap * Cin + ci] = sum over pixels m of dY[m][co] * X[m @ tap][ci]
//
// The reduction index of this GEMM is the PIXEL, and v_mfma_f32_32x32x16_bf16 wants 8 consecutive K values per lane: with channels-last
// tensors that is a column of 8 pixels -- a transposed operand.  Transposing along a SPATIAL axis would collide with the taps (a tap shifts the
// pixel, so X octets would be misaligned against dY octets and half-empty at the padding).  The axis no tap ever moves is the IMAGE index:
//
//   lt_pack_n8_bf16   [N][P][C] fp32  ->  [ceil(N / 8)][P][C][8] bf16     (element (g, p, c) = the 8 images 8g .. 8g+7 at pixel p, channel c,
//                                                                          16 bytes, zero for images past N)
//
// With that layout one 16-byte load IS an MFMA operand (8 K values = the same pixel of 8 images), a tap shifts whole octets, padding is valid
// or not for a whole octet, strides and transposed layers need nothing special, and every address computation of the fp32 kernels carries
// over with "pixel" read as "pixel octet".  An MFMA covers two octet rows (lane half 0 / 1), i.e. 16 of the K = N * Do * Ho * Wo products.
//
//   conv_wgrad16_kernel<CT, KT>     any layer: operands straight from L2 / L1 (the fp32 conv_wgrad_kernel's structure: one wave owns a
//                                   (32 CT) x (32 KT) block of dW, software pipeline of three register sets, slabs + deterministic reduce)
//   conv3d_wgrad16_brick_kernel     3^3 / stride 1 / pad 1 (V2V): dY brick (2 x 2 x 8 voxels) and X halo brick (4 x 4 x 10) in LDS as octets,
//                                   every wave takes every fourth 32-column block of (tap, ci) -- one ds_read_b128 per MFMA operand
//   conv3d_wgrad16_k7_kernel        the 7^3 front layer (32 -> 16): one kd plane of the filter per workgroup on the 16x16x32 MFMA
#include "conv_common.h"
#include "wgrad_reduce.h"

using namespace lt;

namespace {

__device__ __forceinline__ uint4 zero4() { return make_uint4(0u, 0u, 0u, 0u); }

// ---- image-octet packing ----------------------------------------------------------------------------------------------------------------
// thread = (octet group g = blockIdx.y, pixel p, four channels): eight float4 loads (one per image; a wave reads 1 KB runs of each image's
// rows), four 16-byte stores (64 contiguous bytes per thread)
__global__ __launch_bounds__(256) void pack_n8_vec_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int N, long long P, int C, int ld) {
    const int C4 = C >> 2;
    const long long total = P * C4;
    const int g = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += 256ll * gridDim.x) {
        const long long p = i / C4;
        const int c = (int)(i - p * C4) * 4;
        float4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = 8 * g + e;
            v[e] = n < N ? *(const float4*)(src + ((size_t)n * P + p) * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        uint4* o = dst + ((size_t)g * P + p) * C + c;
        o[0] = make_uint4(pack_bf16x2(v[0].x, v[1].x), pack_bf16x2(v[2].x, v[3].x), pack_bf16x2(v[4].x, v[5].x), pack_bf16x2(v[6].x, v[7].x));
        o[1] = make_uint4(pack_bf16x2(v[0].y, v[1].y), pack_bf16x2(v[2].y, v[3].y), pack_bf16x2(v[4].y, v[5].y), pack_bf16x2(v[6].y, v[7].y));
        o[2] = make_uint4(pack_bf16x2(v[0].z, v[1].z), pack_bf16x2(v[2].z, v[3].z), pack_bf16x2(v[4].z, v[5].z), pack_bf16x2(v[6].z, v[7].z));
        o[3] = make_uint4(pack_bf16x2(v[0].w, v[1].w), pack_bf16x2(v[2].w, v[3].w), pack_bf16x2(v[4].w, v[5].w), pack_bf16x2(v[6].w, v[7].w));
    }
}

// any channel count / row stride (the 17-joint output layer, the 3-channel image)
__global__ __launch_bounds__(256) void pack_n8_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int N, long long P, int C, int ld) {
    const long long total = P * C;
    const int g = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += 256ll * gridDim.x) {
        const long long p = i / C;
        const int c = (int)(i - p * C);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = 8 * g + e;
            v[e] = n < N ? src[((size_t)n * P + p) * ld + c] : 0.f;
        }
        dst[((size_t)g * P + p) * C + c] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
}

// the same from a bf16 [N][P][ld] tensor (the copy the mixed-precision convolutions already read): thread = (g, p, eight channels), eight
// 16-byte loads (one per image), an 8 x 8 transpose of 16-bit values with v_perm_b32, eight 16-byte stores (128 contiguous bytes)
__global__ __launch_bounds__(256) void pack_n8_from_bf16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int N, long long P, int C, int ld) {
    const int C8 = C >> 3, ld8 = ld >> 3;
    const long long total = P * C8;
    const int g = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += 256ll * gridDim.x) {
        const long long p = i / C8;
        const int c8 = (int)(i - p * C8);
        uint4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = 8 * g + e;
            v[e] = n < N ? src[((size_t)n * P + p) * ld8 + c8] : zero4();
        }
        uint4* o = dst + ((size_t)g * P + p) * C + c8 * 8;
#pragma unroll
        for (int d = 0; d < 4; ++d) {          // dword d of every image holds channels 2d (low half) and 2d + 1 (high half)
            const unsigned w0 = d == 0 ? v[0].x : d == 1 ? v[0].y : d == 2 ? v[0].z : v[0].w, w1 = d == 0 ? v[1].x : d == 1 ? v[1].y : d == 2 ? v[1].z : v[1].w;
            const unsigned w2 = d == 0 ? v[2].x : d == 1 ? v[2].y : d == 2 ? v[2].z : v[2].w, w3 = d == 0 ? v[3].x : d == 1 ? v[3].y : d == 2 ? v[3].z : v[3].w;
            const unsigned w4 = d == 0 ? v[4].x : d == 1 ? v[4].y : d == 2 ? v[4].z : v[4].w, w5 = d == 0 ? v[5].x : d == 1 ? v[5].y : d == 2 ? v[5].z : v[5].w;
            const unsigned w6 = d == 0 ? v[6].x : d == 1 ? v[6].y : d == 2 ? v[6].z : v[6].w, w7 = d == 0 ? v[7].x : d == 1 ? v[7].y : d == 2 ? v[7].z : v[7].w;
            // __builtin_amdgcn_perm(hi, lo, sel): bytes 0-3 of the result pick from {lo: 0-3, hi: 4-7}
            o[2 * d] = make_uint4(__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u), __builtin_amdgcn_perm(w5, w4, 0x05040100u),
                                  __builtin_amdgcn_perm(w7, w6, 0x05040100u));
            o[2 * d + 1] = make_uint4(__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u), __builtin_amdgcn_perm(w5, w4, 0x07060302u),
                                      __builtin_amdgcn_perm(w7, w6, 0x07060302u));
        }
    }
}

// ---- generic kernel ----------------------------------------------------------------------------------------------------------------------
struct W16Args {
    const uint4* dy;         // [G * Do*Ho*Wo][ldy] octets: gradient of the convolution output, GEMM row m = (g, od, oh, ow)
    const uint4* x;          // [G][D][H][W][Cin] octets
    const int4* taps;        // [ntaps] = (dd, dh, dw, unused)
    float* out;              // S == 1: dw [cout_pad][k_pad];  S > 1: workspace [S][cout_pad][k_pad] of per-slab partial sums
    int D, H, W, Cin, log2Cin, Do, Ho, Wo, sd, sh, sw, pd, ph, pw;
    int Cout, ldy, k_pad, ntaps, M, accumulate, cout_pad;          // M = G * Do * Ho * Wo octet rows
    int n_k_t, n_tiles, rows_per_slab;
};

// bits t = 0 .. 7 with 0 <= i0 + t < size
__device__ __forceinline__ int range_mask16(int i0, int size) {
    const int lo = max(0, -i0), hi = min(7, size - 1 - i0);          // hi < lo: empty
    return hi >= lo ? ((2 << hi) - 1) & ~((1 << lo) - 1) : 0;
}

template <int CT, int KT, int NS>
__global__ __launch_bounds__(256, 2) void conv_wgrad16_kernel(const W16Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every XCD gets a contiguous range of (slab, tile group) pairs, slab-major: the tiles of one slab (same dY / X rows) meet in one L2
    const int total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int lin2 = (total & 7) ? lin : (lin & 7) * (total >> 3) + (lin >> 3);
    const int slab = lin2 / (int)gridDim.x, tgroup = lin2 - slab * (int)gridDim.x;
    const int t = tgroup * 4 + wave;
    if (t >= a.n_tiles) return;
    const int co0 = (t / a.n_k_t) * (32 * CT), k0 = (t % a.n_k_t) * (32 * KT);
    const int col = lane & 31, half = lane >> 5;          // A: co = co0 + 32 c + col, octet row m + half;  B: k = k0 + 32 j + col, octet row m + half
    int tsel[KT], toff[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int k = k0 + 32 * j + col;
        const int tap = k >> a.log2Cin;
        if (tap < a.ntaps) {
            const int4 tp = a.taps[tap];
            tsel[j] = (1 << tp.x) | (1 << (8 + tp.y)) | (1 << (16 + tp.z));
            toff[j] = ((tp.x * a.H + tp.y) * a.W + tp.z) * a.Cin + (k & (a.Cin - 1));
        } else {
            tsel[j] = (int)0x80000000; toff[j] = 0;
        }
    }
    bool co_ok[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) co_ok[c] = co0 + 32 * c + col < a.Cout;
    f32x16 acc[CT][KT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][j][e] = 0.f;
    const int m_begin = slab * a.rows_per_slab;
    const int m_end = min(a.M, m_begin + a.rows_per_slab);
    int m = m_begin + half;
    const int hw = a.Ho * a.Wo, dhw = a.Do * hw;
    int g = m / dhw, r = m - g * dhw;
    int od = r / hw; r -= od * hw;
    int oh = r / a.Wo, ow = r - oh * a.Wo;

    // branch-free loads, validity bits applied in mma() (behind the loads of the next pipeline stages) -- see conv_wgrad_kernel in train.hip
    auto load = [&](uint4 (&av)[CT], uint4 (&bv)[KT], unsigned& okbits) {
        const bool m_ok = m < m_end;
        unsigned bits = 0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const bool ok = m_ok & co_ok[c];
            av[c] = a.dy[ok ? (size_t)m * a.ldy + co0 + 32 * c + col : (size_t)0];
            bits |= ok ? 1u << (KT + c) : 0u;
        }
        const int id0 = od * a.sd - a.pd, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
        const int pix = (((g * a.D + id0) * a.H + ih0) * a.W + iw0) * a.Cin;          // may point in front of the tensor (padding): only in-bounds taps are read
        const int rmask = m_ok ? (range_mask16(id0, a.D) | (range_mask16(ih0, a.H) << 8) | (range_mask16(iw0, a.W) << 16)) : 0;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const bool ok = (rmask & tsel[j]) == tsel[j];
            bv[j] = a.x[ok ? (unsigned)(pix + toff[j]) : 0u];
            bits |= ok ? 1u << j : 0u;
        }
        okbits = bits;
        m += 2; ow += 2;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) { const int w = ow >= a.Wo; ow -= w ? a.Wo : 0; oh += w; }
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) { const int w = oh >= a.Ho; oh -= w ? a.Ho : 0; od += w; }
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) { const int w = od >= a.Do; od -= w ? a.Do : 0; g += w; }
    };
    auto mma = [&](const uint4 (&av)[CT], const uint4 (&bv)[KT], unsigned bits) {
        V16 af[CT], bf[KT];
#pragma unroll
        for (int c = 0; c < CT; ++c) af[c].u = (bits >> (KT + c)) & 1u ? av[c] : zero4();
#pragma unroll
        for (int j = 0; j < KT; ++j) bf[j].u = (bits >> j) & 1u ? bv[j] : zero4();
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int j = 0; j < KT; ++j) acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c].h, bf[j].h, acc[c][j], 0, 0, 0);
    };
    // two octet-row pairs in flight ahead of the one in the MFMAs (rows past m_end load nothing: the extra MFMAs add zeros)
    // (NS = 2 for the 32 x 256 wave tile: nine operands per set, a third set would spill)
    uint4 av0[CT], bv0[KT], av1[CT], bv1[KT];
    unsigned ok0, ok1;
    const int nit = (m_end - m_begin + 1) >> 1;
    if constexpr (NS == 3) {
        uint4 av2[CT], bv2[KT];
        unsigned ok2;
        load(av0, bv0, ok0); load(av1, bv1, ok1);
        for (int it = 0; it < nit; it += 3) {
            load(av2, bv2, ok2); mma(av0, bv0, ok0);
            load(av0, bv0, ok0); mma(av1, bv1, ok1);
            load(av1, bv1, ok1); mma(av2, bv2, ok2);
        }
    } else {
        load(av0, bv0, ok0);
        for (int it = 0; it < nit; it += 2) {
            load(av1, bv1, ok1); mma(av0, bv0, ok0);
            load(av0, bv0, ok0); mma(av1, bv1, ok1);
        }
    }
    float* out = a.out + (size_t)slab * a.cout_pad * a.k_pad;
    const bool direct_acc = a.accumulate && gridDim.y == 1;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = co0 + 32 * c + 8 * (e >> 2) + 4 * half + (e & 3);       // C layout of the 32x32 MFMA: row = co, column = k
                const int k = k0 + 32 * j + col;
                if (k < a.k_pad && row < a.cout_pad) {
                    float* dst = out + (size_t)row * a.k_pad + k;
                    *dst = direct_acc ? *dst + acc[c][j][e] : acc[c][j][e];
                }
            }
}

struct Plan16 { int variant, n_co_t, n_k_t, S, rows_per_slab; };

// tile shape by layer shape as in the fp32 kernel.  Slab count: an octet row carries eight times the work of a pixel row and the MFMA is 16x faster,
// so the partial sums (written once, read once by the reduce) cost as much as the GEMM itself: S minimises a two-term model -- the kernel
// (rounds of 512 resident workgroups x octet-row pairs per slab x 8 MFMAs at ~1.5x their issue time) plus the reduce (S x n floats at ~3 TB/s)
Plan16 plan16(long long M, int cout_pad, int k_pad) {
    Plan16 p;
    p.variant = k_pad <= 64 ? 0 : cout_pad <= 32 ? 2 : 1;
    const int ct = p.variant == 0 ? 4 : p.variant == 1 ? 2 : 1, kt = 8 / ct;
    p.n_co_t = (int)cdiv(cout_pad, 32 * ct); p.n_k_t = (int)cdiv(k_pad, 32 * kt);
    const long long wgs = cdiv((long long)p.n_co_t * p.n_k_t, 4);
    const double n_bytes = (double)cout_pad * k_pad * 4.0;
    const long long cap = (long long)(((size_t)48 << 20) / (size_t)n_bytes);
    long long best = 1;
    double best_t = 1e30;
    for (long long S = 1; S <= 512; S = S < 8 ? S + 1 : S + 8) {
        if (S > 1 && (S > cap || S > M / 16)) break;
        const double rounds = (double)cdiv(wgs * S, 512);
        const double kernel_us = rounds * (double)cdiv(M, 2 * S) * 0.226;
        const double reduce_us = S > 1 ? 3.0 + S * n_bytes / 3.0e6 : 0.0;
        if (kernel_us + reduce_us < best_t) { best_t = kernel_us + reduce_us; best = S; }
    }
    long long S = best;                 // >= 8: a multiple of 8 (the kernel gives each XCD a contiguous range of slabs)
    long long rps = cdiv(M, S);
    rps += rps & 1;
    p.rows_per_slab = (int)rps;
    p.S = (int)S;                   // trailing slabs may be empty (they write zeros)
    return p;
}

// ---- the same GEMM straight from the channels-last bf16 tensors (no octet pack) ------------------------------------------------------------------
// The octet pack is a transpose of (image, channel) per pixel: eight 16-byte loads (one per image of the group, the same pixel, eight channels),
// v_perm_b32 on the halves, eight 16-byte stores -- and the generic kernel then reads each octet back exactly once per tile column / row.  Here the
// loads and the v_perm stay, the stores and the second read go: a lane of the 16x16x32 MFMA takes ACH (BCH) CONSECUTIVE CHANNELS of dY (X at its tap)
// from the eight images of its pixel, and the transpose hands it ACH (BCH) operands at once -- operand a of lane i is channel ACH i + a, so the rows
// of an accumulator tile are the channels {ACH i + a : i}: a permutation of the output rows that only the epilogue has to know.  K of one MFMA = 4
// pixels (lane / 16) x 8 images.  Per step and wave: 8 loads of 2 ACH bytes + 8 of 2 BCH bytes (runs of 32 ACH bytes per pixel and image), ACH x BCH
// MFMAs; the same bytes per MAC as the packed kernel's 64 x 128 wave tile.  Measured on the 16-bit-activation step at 8 samples: the packs were
// 7.9 ms per step on one stream (12.5 ms beside the main stream's kernels) of 22 ms of weight-gradient work.
struct W16UArgs {
    const unsigned short* dy;   // [N][Do*Ho*Wo][ldy] bf16
    const unsigned short* x;    // [N][D][H][W][ldx] bf16
    const int4* taps;
    float* out;
    int N, D, H, W, Cin, log2Cin, ldx, Do, Ho, Wo, sd, sh, sw, pd, ph, pw;
    int Cout, ldy, k_pad, ntaps, M, accumulate, cout_pad;
    int n_k_t, n_tiles, rows_per_slab;
    unsigned img_a, img_b;      // elements between two images of dY / of X
};

template <int CH> struct RawCh;
template <> struct RawCh<8> {
    unsigned d[4];
    __device__ __forceinline__ void load(const unsigned short* p) { const uint4 v = *(const uint4*)p; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
};
template <> struct RawCh<4> {
    unsigned d[2];
    __device__ __forceinline__ void load(const unsigned short* p) { const uint2 v = *(const uint2*)p; d[0] = v.x; d[1] = v.y; }
};

// The four waves of a workgroup share ONE tile and split the slab's rows four ways (their sums meet in LDS, added in wave order: deterministic): a
// quarter of the slabs for the same number of waves, i.e. a quarter of the partial sums written and read back -- which at 8 samples per step cost as
// much as the GEMM (layer3, 1024 x 256 outputs over 2304 octet rows: 32 slabs x 1 MB before, 8 now).
template <int ACH, int BCH, int NS>
__global__ __launch_bounds__(256) void conv_wgrad16u_kernel(const W16UArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;          // XCD-contiguous (slab, tile) ranges, slab-major
    const int lin2 = (total & 7) ? lin : (lin & 7) * (total >> 3) + (lin >> 3);
    const int slab = lin2 / (int)gridDim.x, t = lin2 - slab * (int)gridDim.x;
    const int co0 = (t / a.n_k_t) * (16 * ACH), k0 = (t % a.n_k_t) * (16 * BCH);
    const int i = lane & 15, q = lane >> 4;
    const int co = co0 + ACH * i;
    const bool co_ok = co < a.Cout;          // Cout is a multiple of ACH: all of the lane's channels or none
    const int kc = k0 + BCH * i, tap = kc >> a.log2Cin;          // Cin is a multiple of BCH: the lane's columns share the tap
    int tsel, toff;
    if (tap < a.ntaps) {
        const int4 tp = a.taps[tap];
        tsel = (1 << tp.x) | (1 << (8 + tp.y)) | (1 << (16 + tp.z));
        toff = ((tp.x * a.H + tp.y) * a.W + tp.z) * a.ldx + (kc & (a.Cin - 1));
    } else {
        tsel = (int)0x80000000; toff = 0;
    }
    f32x4 acc[ACH][BCH];
#pragma unroll
    for (int c = 0; c < ACH; ++c)
#pragma unroll
        for (int j = 0; j < BCH; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][j][e] = 0.f;
    const int rq = a.rows_per_slab >> 2;          // (a multiple of 4: whole steps)
    const int m_begin = slab * a.rows_per_slab + wave * rq;
    const int m_end = min(a.M, m_begin + rq);
    int m = m_begin + q;
    const int hw = a.Ho * a.Wo, dhw = a.Do * hw;
    int g = m / dhw, r = m - g * dhw;
    int od = r / hw; r -= od * hw;
    int oh = r / a.Wo, ow = r - oh * a.Wo;

    // bits 0-7 of okbits: image e of the lane's dY row is there (row < m_end, channel < Cout, image < N); bit 8: the lane's X tap is inside the volume.
    // A zero dY operand is enough for rows / images that do not exist (X is then read at a clamped, valid address), a zero X operand for padding.
    auto load = [&](RawCh<ACH> (&ra)[8], RawCh<BCH> (&rb)[8], unsigned& okbits) {
        const bool m_ok = m < m_end;
        const int nv = m_ok ? min(8, a.N - 8 * g) : 0;
        const bool aok = m_ok & co_ok;
        const unsigned offa = aok ? (unsigned)(8 * g) * a.img_a + (unsigned)(((od * a.Ho + oh) * a.Wo + ow) * a.ldy + co) : 0u;
        unsigned bits = aok ? (1u << nv) - 1u : 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[e].load(a.dy + offa + (e < nv ? (unsigned)e * a.img_a : 0u));
        const int id0 = od * a.sd - a.pd, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
        const int pix = ((id0 * a.H + ih0) * a.W + iw0) * a.ldx;          // may point in front of the image (padding): only in-bounds taps are read
        const int rmask = m_ok ? (range_mask16(id0, a.D) | (range_mask16(ih0, a.H) << 8) | (range_mask16(iw0, a.W) << 16)) : 0;
        const bool bok = (rmask & tsel) == tsel;
        const unsigned offb = bok ? (unsigned)(8 * g) * a.img_b + (unsigned)(pix + toff) : 0u;
        bits |= bok ? 256u : 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) rb[e].load(a.x + offb + (bok && e < nv ? (unsigned)e * a.img_b : 0u));
        okbits = bits;
        m += 4; ow += 4;
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) { const int w = ow >= a.Wo; ow -= w ? a.Wo : 0; oh += w; }
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) { const int w = oh >= a.Ho; oh -= w ? a.Ho : 0; od += w; }
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) { const int w = od >= a.Do; od -= w ? a.Do : 0; g += w; }
    };
    // operand c of the lane = its channel c of the eight images: dword d = (image 2d, image 2d + 1), the image-octet layout of lt_pack_n8_bf16
    auto mma = [&](const RawCh<ACH> (&ra)[8], const RawCh<BCH> (&rb)[8], unsigned bits) {
        V16 af[ACH], bf[BCH];
        const bool bok = bits & 256u;
#pragma unroll
        for (int c = 0; c < ACH; ++c) {
            unsigned w[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned lo = (bits >> (2 * d)) & 1u ? ra[2 * d].d[c >> 1] : 0u, hi = (bits >> (2 * d + 1)) & 1u ? ra[2 * d + 1].d[c >> 1] : 0u;
                w[d] = __builtin_amdgcn_perm(hi, lo, (c & 1) ? 0x07060302u : 0x05040100u);
            }
            af[c].u = make_uint4(w[0], w[1], w[2], w[3]);
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            unsigned w[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] = bok ? __builtin_amdgcn_perm(rb[2 * d + 1].d[j >> 1], rb[2 * d].d[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u) : 0u;
            bf[j].u = make_uint4(w[0], w[1], w[2], w[3]);
        }
#pragma unroll
        for (int c = 0; c < ACH; ++c)
#pragma unroll
            for (int j = 0; j < BCH; ++j) acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c].h, bf[j].h, acc[c][j], 0, 0, 0);
    };
    RawCh<ACH> ra0[8], ra1[8];
    RawCh<BCH> rb0[8], rb1[8];
    unsigned ok0, ok1;
    const int nit = max(0, (m_end - m_begin + 3) >> 2);
    if constexpr (NS == 3) {          // two steps in flight ahead of the one in the MFMAs (one wave per SIMD: nothing else hides the L2 latency)
        RawCh<ACH> ra2[8];
        RawCh<BCH> rb2[8];
        unsigned ok2;
        load(ra0, rb0, ok0); load(ra1, rb1, ok1);
        for (int it = 0; it < nit; it += 3) {          // (rows past m_end load nothing new: a zero dY operand)
            load(ra2, rb2, ok2); mma(ra0, rb0, ok0);
            load(ra0, rb0, ok0); mma(ra1, rb1, ok1);
            load(ra1, rb1, ok1); mma(ra2, rb2, ok2);
        }
    } else {
        load(ra0, rb0, ok0);
        for (int it = 0; it < nit; it += 2) {
            load(ra1, rb1, ok1); mma(ra0, rb0, ok0);
            load(ra0, rb0, ok0); mma(ra1, rb1, ok1);
        }
    }
    // waves 1-3 hand their sums to wave 0 through LDS ([wave - 1][register][lane]: conflict-free both ways)
    float* red = (float*)smem16;
    if (wave > 0) {
#pragma unroll
        for (int c = 0; c < ACH; ++c)
#pragma unroll
            for (int j = 0; j < BCH; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[(((wave - 1) * (ACH * BCH * 4)) + (c * BCH + j) * 4 + e) * 64 + lane] = acc[c][j][e];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll 1
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int c = 0; c < ACH; ++c)
#pragma unroll
            for (int j = 0; j < BCH; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c][j][e] += red[((w * (ACH * BCH * 4)) + (c * BCH + j) * 4 + e) * 64 + lane];
    float* out = a.out + (size_t)slab * a.cout_pad * a.k_pad;
    const bool direct_acc = a.accumulate && gridDim.y == 1;
    // C of the 16x16 MFMA: row = 4 (lane / 16) + e <-> the A lane of that index, column = lane % 16 <-> the B lane: four (BCH = 4) or eight consecutive
    // columns of dW per lane and row -- 16-byte stores
#pragma unroll
    for (int c = 0; c < ACH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = co0 + ACH * (4 * q + e) + c;
            if (row >= a.cout_pad) continue;
#pragma unroll
            for (int j4 = 0; j4 < BCH; j4 += 4) {
                const int k = kc + j4;
                if (k >= a.k_pad) continue;          // (k_pad is a multiple of 4)
                float4* dst = (float4*)(out + (size_t)row * a.k_pad + k);
                float4 v = make_float4(acc[c][j4][e], acc[c][j4 + 1][e], acc[c][j4 + 2][e], acc[c][j4 + 3][e]);
                if (direct_acc) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *dst = v;
            }
        }
}

// wave tile (16 ACH) x (16 BCH): 128 x 64 where there are 128 output channels to fill it, 64 x 128 below
Plan16 plan16u(long long M, int cout_pad, int k_pad) {
    Plan16 p;
    p.variant = cout_pad > 64 ? 0 : 1;
    const int tco = p.variant == 0 ? 128 : 64, tk = p.variant == 0 ? 64 : 128;
    p.n_co_t = (int)cdiv(cout_pad, tco); p.n_k_t = (int)cdiv(k_pad, tk);
    const long long wgs = (long long)p.n_co_t * p.n_k_t;          // one tile per workgroup, its four waves split the slab's rows
    const double n_bytes = (double)cout_pad * k_pad * 4.0;
    const long long cap = (long long)(((size_t)48 << 20) / (size_t)n_bytes);
    long long best = 1;
    double best_t = 1e30;
    for (long long S = 1; S <= 512; S = S < 8 ? S + 1 : S + 8) {          // the packed kernel's two-term model: a step is four octet rows and 32 MFMAs of 16 cycles
        if (S > 1 && (S > cap || S > M / 64)) break;
        const double rounds = (double)cdiv(wgs * S, 256);
        const double kernel_us = rounds * ((double)cdiv(M, 16 * S) * 0.8 + 2.0);          // (measured: ~0.8 us per step -- the operands come from L2 at ~9 TB/s)
        const double reduce_us = S > 1 ? 3.0 + S * n_bytes / 3.0e6 : 0.0;
        if (kernel_us + reduce_us < best_t) { best_t = kernel_us + reduce_us; best = S; }
    }
    long long rps = cdiv(M, best);
    rps = (rps + 15) & ~15ll;
    p.rows_per_slab = (int)rps;
    p.S = (int)best;
    return p;
}

// ---- 3^3 / stride 1 / pad 1 from LDS bricks of octets ---------------------------------------------------------------------------------------
constexpr int B16_D = 2, B16_H = 2, B16_W = 8, B16_VOX = B16_D * B16_H * B16_W;                                   // 32 voxels = 16 MFMA K steps
constexpr int B16_HD = B16_D + 2, B16_HH = B16_H + 2, B16_HW = B16_W + 2, B16_HVOX = B16_HD * B16_HH * B16_HW;    // 160 halo voxels

struct Brick16Args {
    const uint4* dy;         // [G * D*H*W][ldy] octets
    const uint4* x;          // [G][D][H][W][Cin] octets
    const int4* taps;        // 27 x (dd, dh, dw, -)
    float* out;              // [S][cout_pad][k_pad]
    int D, H, W, Cin, ldy, cout_pad, k_pad;
    int cib;                 // ci per workgroup: min(Cin, 32) (16 or 32)
    int nblk;                // 32-column blocks of the (tap, ci within the workgroup's cib) space: ceil(27 * cib / 32)
    int nbd, nbh, nbw;       // bricks per dimension
    int nbricks, bricks_per_slab;          // over (g, bd, bh, bw)
};

// workgroup = (32 co, CIB ci, slab of bricks); wave w takes the column blocks w, w + 4, ... (at most 7: 27 taps x 32 ci = 27 blocks).
// LDS: X halo brick [160][CIB] octets (80 KB at CIB = 32) + dY brick [32][32] octets (16 KB); lanes of a half-wave read 512 contiguous bytes.
// The NEXT brick's operands are loaded into registers (up to 24 x 16 bytes per thread) before the MFMA loop of the current one and written to
// LDS behind it: the global-memory latency hides behind the MFMAs (one workgroup per CU, so nothing else would).
template <int CIB>
__global__ __launch_bounds__(256) void conv3d_wgrad16_brick_kernel(const Brick16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    constexpr int L2C = CIB == 32 ? 5 : 4;
    constexpr int NX = B16_HVOX * CIB / 256;              // X octets per thread (20 / 10)
    constexpr int NY = B16_VOX * 32 / 256;                // dY octets per thread (4)
    uint4* xs = (uint4*)smem16;                           // [B16_HVOX][CIB]
    uint4* ds = xs + B16_HVOX * CIB;                      // [B16_VOX][32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * CIB;
    int toff[7], kcol[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int b = wave + 4 * i;
        const int c = 32 * b + col;                       // column of the (tap, ci) space of this workgroup
        const int tap = c >> L2C, cil = c & (CIB - 1);
        if (b < a.nblk && tap < 27) {
            const int4 tp = a.taps[tap];
            toff[i] = ((tp.x * B16_HH + tp.y) * B16_HW + tp.z) * CIB + cil;
            kcol[i] = tap * a.Cin + ci0 + cil;
        } else {
            toff[i] = 0; kcol[i] = -1;
        }
    }
    // staging slots of this thread: X octet k = halo voxel (tid >> L2C) + (256 >> L2C) k, channel tid & (CIB - 1); coordinates packed once
    const int xq = threadIdx.x & (CIB - 1);
    int xco[NX];                                          // hd | hh << 8 | hw << 16 of the halo voxel
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int v = (threadIdx.x >> L2C) + (256 >> L2C) * k;
        const int hw_ = v % B16_HW, t2 = v / B16_HW;
        xco[k] = (t2 / B16_HH) | ((t2 % B16_HH) << 8) | (hw_ << 16);
    }
    const int yq = threadIdx.x & 31;
    f32x16 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int b_begin = blockIdx.z * a.bricks_per_slab, b_end = min(a.nbricks, b_begin + a.bricks_per_slab);
    uint4 rx[NX], ry[NY];
    unsigned okm = 0;
    auto issue = [&](int b) {
        int r = b;
        const int bw = r % a.nbw; r /= a.nbw;
        const int bh = r % a.nbh; r /= a.nbh;
        const int bd = r % a.nbd;
        const int g = r / a.nbd;
        const int d0 = bd * B16_D, h0 = bh * B16_H, w0 = bw * B16_W;
        const uint4* xg = a.x + (size_t)g * a.D * a.H * a.W * a.Cin + ci0 + xq;
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int d = d0 + (xco[k] & 255) - 1, h = h0 + ((xco[k] >> 8) & 255) - 1, w = w0 + (xco[k] >> 16) - 1;
            const bool ok = (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            rx[k] = xg[ok ? (unsigned)(((d * a.H + h) * a.W + w) * a.Cin) : 0u];          // (validity applied when the registers go to LDS)
            m |= ok ? 1u << k : 0u;
        }
        okm = m;
        const uint4* yg = a.dy + co0 + yq;
#pragma unroll
        for (int k = 0; k < NY; ++k) {
            const int v = (threadIdx.x >> 5) + 8 * k;
            const int w = v % B16_W, t2 = v / B16_W;
            const int h = t2 % B16_H, d = t2 / B16_H;
            const size_t row = (((size_t)g * a.D + d0 + d) * a.H + h0 + h) * a.W + w0 + w;
            ry[k] = yg[row * a.ldy];
        }
    };
    if (b_begin < b_end) issue(b_begin);
    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();          // the previous brick's reads are done
#pragma unroll
        for (int k = 0; k < NX; ++k) xs[threadIdx.x + 256 * k] = (okm >> k) & 1u ? rx[k] : zero4();
#pragma unroll
        for (int k = 0; k < NY; ++k) ds[threadIdx.x + 256 * k] = ry[k];
        __syncthreads();
        if (b + 1 < b_end) issue(b + 1);                  // in flight during the MFMAs below
#pragma unroll 2
        for (int p = 0; p < B16_VOX / 2; ++p) {
            const int v = 2 * p + half;                       // brick-linear voxel (w fastest): a pair never straddles a row
            const int w = v % B16_W, t2 = v / B16_W;
            const int h = t2 % B16_H, d = t2 / B16_H;
            V16 av; av.u = ds[v * 32 + col];
            const int hb = ((d * B16_HH + h) * B16_HW + w) * CIB;          // halo voxel of tap (0, 0, 0)
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                V16 bv; bv.u = xs[hb + toff[i]];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.h, bv.h, acc[i], 0, 0, 0);
            }
        }
    }
    float* out = a.out + (size_t)blockIdx.z * a.cout_pad * a.k_pad;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (kcol[i] < 0) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = co0 + 8 * (e >> 2) + 4 * half + (e & 3);
            out[(size_t)row * a.k_pad + kcol[i]] = acc[i][e];
        }
    }
}

// The same brick kernel fed from the channels-last bf16 tensors (Cin a multiple of 32): a staging UNIT is (voxel, eight channels) -- eight 16-byte loads,
// one per image of the octet group, transposed with v_perm_b32 into the eight octets of those channels on their way into LDS.  A thread's unit is a
// 128-byte chunk of the LDS image, so the eight octets are written ROTATED inside the chunk (slot = (channel + 4 (voxel & 1) + chunk) & 7: eight
// consecutive lanes = two voxels x four chunks hit eight different slots; unrotated the ds_write_b128 of a lane group would be an 8-way bank conflict);
// the readers undo it with one XOR per MFMA (X: the voxel's parity flips bit 2 of the slot) or not at all (dY: the parity is the lane half).
struct Brick16UArgs {
    const unsigned short* dy;   // [N][D*H*W][ldy] bf16
    const unsigned short* x;    // [N][D][H][W][Cin] bf16
    const int4* taps;
    float* out;
    int N, D, H, W, Cin, ldy, cout_pad, k_pad;
    int nbd, nbh, nbw, nbricks, bricks_per_slab;
};

__global__ __launch_bounds__(256) void conv3d_wgrad16_brick_u_kernel(const Brick16UArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    constexpr int CIB = 32;
    constexpr int XU = B16_HVOX * 4, NXU = (XU + 255) / 256;          // 640 units of X: two per thread and a third for the first 128
    uint4* xs = (uint4*)smem16;                           // [B16_HVOX][CIB], rotated inside every 8-octet chunk
    uint4* ds = xs + B16_HVOX * CIB;                      // [B16_VOX][32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * CIB;
    int toff[7], kcol[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int b = wave + 4 * i;
        const int c = 32 * b + col;                       // column of the (tap, ci) space of this workgroup
        const int tap = c >> 5, cil = c & 31;
        if (tap < 27) {
            const int4 tp = a.taps[tap];
            const int tv = (tp.x * B16_HH + tp.y) * B16_HW + tp.z;
            toff[i] = tv * CIB + (cil & ~7) + ((((cil & 7) + (cil >> 3)) & 7) ^ (4 * (tv & 1)));
            kcol[i] = tap * a.Cin + ci0 + cil;
        } else {
            toff[i] = 0; kcol[i] = -1;
        }
    }
    const int acol = (col & ~7) + (((col & 7) + (col >> 3) + 4 * half) & 7);          // dY: voxel parity = lane half
    // staging units of this thread: X unit k = (halo voxel, chunk) (tid + 256 k) / 4, % 4; dY unit (tid < 128) = (voxel tid / 4, chunk tid % 4)
    int xco[NXU];
#pragma unroll
    for (int k = 0; k < NXU; ++k) {
        const int v = (threadIdx.x + 256 * k) >> 2;
        const int hw_ = v % B16_HW, t2 = v / B16_HW;
        xco[k] = (t2 / B16_HH) | ((t2 % B16_HH) << 8) | (hw_ << 16);
    }
    const int c8 = threadIdx.x & 3;
    const bool x2 = threadIdx.x < XU - 512, yu = threadIdx.x < 128;
    f32x16 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int b_begin = blockIdx.z * a.bricks_per_slab, b_end = min(a.nbricks, b_begin + a.bricks_per_slab);
    const unsigned img_x = (unsigned)(a.D * a.H * a.W) * (unsigned)a.Cin, img_y = (unsigned)(a.D * a.H * a.W) * (unsigned)a.ldy;
    uint4 rx[NXU][8], ry[8];
    unsigned okm = 0;          // bits 0-2: the X units' voxels are inside the volume; bits 8-15: image e of the group exists
    auto issue = [&](int b) {
        int r = b;
        const int bw = r % a.nbw; r /= a.nbw;
        const int bh = r % a.nbh; r /= a.nbh;
        const int bd = r % a.nbd;
        const int g = r / a.nbd;
        const int d0 = bd * B16_D, h0 = bh * B16_H, w0 = bw * B16_W;
        const int nv = min(8, a.N - 8 * g);
        const unsigned short* xg = a.x + (size_t)(8 * g) * img_x + ci0 + 8 * c8;
        unsigned m = ((1u << nv) - 1u) << 8;
#pragma unroll
        for (int k = 0; k < NXU; ++k) {
            if (k == 2 && !x2) break;
            const int d = d0 + (xco[k] & 255) - 1, h = h0 + ((xco[k] >> 8) & 255) - 1, w = w0 + (xco[k] >> 16) - 1;
            const bool ok = (unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((d * a.H + h) * a.W + w) * a.Cin) : 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) rx[k][e] = *(const uint4*)(xg + off + (e < nv ? (unsigned)e * img_x : 0u));          // (validity applied on the way to LDS)
            m |= ok ? 1u << k : 0u;
        }
        okm = m;
        if (yu) {
            const int v = threadIdx.x >> 2;
            const int w = v % B16_W, t2 = v / B16_W;
            const int h = t2 % B16_H, d = t2 / B16_H;
            const unsigned row = (unsigned)(((d0 + d) * a.H + h0 + h) * a.W + w0 + w);
            const unsigned short* yg = a.dy + (size_t)(8 * g) * img_y + (size_t)row * a.ldy + co0 + 8 * c8;
#pragma unroll
            for (int e = 0; e < 8; ++e) ry[e] = *(const uint4*)(yg + (e < nv ? (unsigned)e * img_y : 0u));
        }
    };
    // the eight octets of a unit (channel c of the eight images: dword d = images 2d, 2d + 1) into the chunk at dst, rotated by rot
    auto put = [&](uint4* dst, const uint4 (&raw)[8], unsigned imgs, bool ok, int rot) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned w[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint4 &r0 = raw[2 * d], &r1 = raw[2 * d + 1];
                const unsigned lo_ = (c >> 1) == 0 ? r0.x : (c >> 1) == 1 ? r0.y : (c >> 1) == 2 ? r0.z : r0.w;
                const unsigned hi_ = (c >> 1) == 0 ? r1.x : (c >> 1) == 1 ? r1.y : (c >> 1) == 2 ? r1.z : r1.w;
                const unsigned lo = ok && ((imgs >> (2 * d)) & 1u) ? lo_ : 0u, hi = ok && ((imgs >> (2 * d + 1)) & 1u) ? hi_ : 0u;
                w[d] = __builtin_amdgcn_perm(hi, lo, (c & 1) ? 0x07060302u : 0x05040100u);
            }
            dst[(c + rot) & 7] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    };
    if (b_begin < b_end) issue(b_begin);
    for (int b = b_begin; b < b_end; ++b) {
        __syncthreads();          // the previous brick's reads are done
        const unsigned imgs = okm >> 8;
#pragma unroll
        for (int k = 0; k < NXU; ++k) {
            if (k == 2 && !x2) break;
            const int u = threadIdx.x + 256 * k;          // = 4 voxel + chunk: rot = 4 (voxel & 1) + chunk = u & 7
            put(xs + 8 * u, rx[k], imgs, (okm >> k) & 1u, u & 7);
        }
        if (yu) put(ds + 8 * threadIdx.x, ry, imgs, true, threadIdx.x & 7);
        __syncthreads();
        if (b + 1 < b_end) issue(b + 1);                  // in flight during the MFMAs below
#pragma unroll 2
        for (int p = 0; p < B16_VOX / 2; ++p) {
            const int v = 2 * p + half;                       // brick-linear voxel (w fastest): a pair never straddles a row
            const int w = v % B16_W, t2 = v / B16_W;
            const int h = t2 % B16_H, d = t2 / B16_H;
            V16 av; av.u = ds[v * 32 + acol];
            const int hv = (d * B16_HH + h) * B16_HW + w;          // halo voxel of tap (0, 0, 0)
            const int hb = hv * CIB, x4 = 4 * (hv & 1);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                V16 bv; bv.u = xs[hb + (toff[i] ^ x4)];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.h, bv.h, acc[i], 0, 0, 0);
            }
        }
    }
    float* out = a.out + (size_t)blockIdx.z * a.cout_pad * a.k_pad;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (kcol[i] < 0) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = co0 + 8 * (e >> 2) + 4 * half + (e & 3);
            out[(size_t)row * a.k_pad + kcol[i]] = acc[i][e];
        }
    }
}

// ---- the 7^3 front layer (32 -> 16) ---------------------------------------------------------------------------------------------------------
// One workgroup owns ONE kd plane of the filter: 49 taps x 16 co x 32 ci = 98 blocks of the 16x16x32 MFMA (K = 4 voxels x 8 images), 24-25 blocks
// of 4 registers per wave.  Bricks of 1 x 2 x 8 voxels: dY brick (16 voxels x 16 co octets, 4 KB) and the X plane the kd needs with a 3-voxel
// (h, w) halo (8 x 14 voxels x 32 ci octets, 56 KB) in LDS: two workgroups per CU.  Taps must be in (kd, kh, kw) order.
constexpr int K16_H = 2, K16_W = 8, K16_VOX = K16_H * K16_W, K16_HH = K16_H + 6, K16_HW = K16_W + 6, K16_HVOX = K16_HH * K16_HW;

__global__ __launch_bounds__(256, 2) void conv3d_wgrad16_k7_kernel(const Brick16Args a) {
    __shared__ __attribute__((aligned(16))) uint4 xs[K16_HVOX * 32];          // 56 KB
    __shared__ __attribute__((aligned(16))) uint4 ds[K16_VOX * 16];           // 4 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;            // MFMA 16x16x32: A row / B column = lane % 16, K group (voxel of the quad) = lane / 16
    const int kd = blockIdx.x;
    int boff[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        const int b = wave + 4 * i;
        const int t = b < 98 ? b >> 1 : 0;
        const int4 tp = a.taps[kd * 49 + t];
        if (tp.x != kd) __builtin_trap();
        boff[i] = (tp.y * K16_HW + tp.z) * 32 + (b & 1) * 16;
    }
    f32x4 acc[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int b_begin = blockIdx.z * a.bricks_per_slab, b_end = min(a.nbricks, b_begin + a.bricks_per_slab);
    for (int br = b_begin; br < b_end; ++br) {
        int r = br;
        const int bw = r % a.nbw; r /= a.nbw;
        const int bh = r % a.nbh; r /= a.nbh;
        const int d = r % a.D;                            // bricks are one plane deep: nbd = D
        const int g = r / a.D;
        const int h0 = bh * K16_H, w0 = bw * K16_W;
        const int dx_ = d + kd - 3;                       // the X plane this kd reads for output plane d
        __syncthreads();
        const bool plane_ok = (unsigned)dx_ < (unsigned)a.D;
        for (int i = threadIdx.x; i < K16_HVOX * 32; i += 256) {
            const int v = i >> 5, q = i & 31;
            const int hw_ = v % K16_HW, hh_ = v / K16_HW;
            const int h = h0 + hh_ - 3, w = w0 + hw_ - 3;
            uint4 val = zero4();
            if (plane_ok && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                val = a.x[((((size_t)g * a.D + dx_) * a.H + h) * a.W + w) * 32 + q];
            xs[i] = val;
        }
        for (int i = threadIdx.x; i < K16_VOX * 16; i += 256) {
            const int v = i >> 4, q = i & 15;
            const int w = v % K16_W, h = v / K16_W;
            const size_t row = (((size_t)g * a.D + d) * a.H + h0 + h) * a.W + w0 + w;
            ds[i] = a.dy[row * a.ldy + q];
        }
        __syncthreads();
        if (plane_ok) {
#pragma unroll 2
            for (int s4 = 0; s4 < K16_VOX / 4; ++s4) {
                const int v = 4 * s4 + kq;                        // four voxels along w per MFMA (8 wide: a quad never straddles a row)
                const int w = v % K16_W, h = v / K16_W;
                V16 av; av.u = ds[v * 16 + col];
                const uint4* xb = xs + (h * K16_HW + w) * 32 + col;
#pragma unroll
                for (int i = 0; i < 25; ++i) {
                    V16 bv; bv.u = xb[boff[i]];
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av.h, bv.h, acc[i], 0, 0, 0);
                }
            }
        }
    }
    float* out = a.out + (size_t)blockIdx.z * a.cout_pad * a.k_pad;
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        const int b = wave + 4 * i;
        if (b >= 98) break;
        const int k = (kd * 49 + (b >> 1)) * 32 + (b & 1) * 16 + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(size_t)(4 * kq + e) * a.k_pad + k] = acc[i][e];       // C: row = 4 (lane / 16) + e = co, column = lane % 16 = ci
    }
}

bool brick16_ok(int D, int H, int W, int Cin, int Do, int Ho, int Wo, const int32_t* stride, const int32_t* pad, int Cout, int cout_pad, int k_pad, int ntaps) {
    if (ntaps != 27 || D != Do || H != Ho || W != Wo || stride[0] != 1 || stride[1] != 1 || stride[2] != 1 || pad[0] != 1 || pad[1] != 1 || pad[2] != 1) return false;
    if ((Cin != 16 && Cin % 32) || Cout % 32 || cout_pad != Cout || k_pad != 27 * Cin) return false;
    return D % B16_D == 0 && H % B16_H == 0 && W % B16_W == 0;
}

void reduce16(const void* workspace, float* dw, long long n, int S, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid(n)), dim3(256), 0, st, (const float*)workspace, dw, n, S, accumulate);
}

}  // namespace

extern "C" size_t lt_pack_n8_bf16_bytes(int32_t N, int64_t P, int32_t C) {
    if (N < 1 || P < 1 || C < 1) return 0;
    return (size_t)cdiv(N, 8) * (size_t)P * (size_t)C * 16;
}

extern "C" int lt_pack_n8_bf16(const float* src, void* dst, int32_t N, int64_t P, int32_t C, int32_t ld, void* stream) {
    LT_REQUIRE(src && dst && N >= 1 && P >= 1 && C >= 1 && ld >= C, LT_ERR_INVALID, "lt_pack_n8_bf16: bad argument");
    LT_REQUIRE((size_t)dst % 16 == 0, LT_ERR_INVALID, "lt_pack_n8_bf16: dst must be 16-byte aligned");
    const int G = (int)cdiv(N, 8);
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && ld % 4 == 0 && (size_t)src % 16 == 0) {
        const long long blocks = cdiv(P * (C / 4), 256);
        hipLaunchKernelGGL(pack_n8_vec_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192), (unsigned)G), dim3(256), 0, st, src, (uint4*)dst, N, (long long)P, C, ld);
    } else {
        const long long blocks = cdiv(P * C, 256);
        hipLaunchKernelGGL(pack_n8_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192), (unsigned)G), dim3(256), 0, st, src, (uint4*)dst, N, (long long)P, C, ld);
    }
    LT_CHECK_LAUNCH("lt_pack_n8_bf16");
    return LT_OK;
}

extern "C" int lt_pack_n8_from_bf16(const void* src, void* dst, int32_t N, int64_t P, int32_t C, int32_t ld, void* stream) {
    LT_REQUIRE(src && dst && N >= 1 && P >= 1 && C >= 8 && ld >= C && C % 8 == 0 && ld % 8 == 0, LT_ERR_INVALID, "lt_pack_n8_from_bf16: bad argument (C, ld multiples of 8)");
    LT_REQUIRE((size_t)dst % 16 == 0 && (size_t)src % 16 == 0, LT_ERR_INVALID, "lt_pack_n8_from_bf16: 16-byte aligned pointers");
    const int G = (int)cdiv(N, 8);
    const long long blocks = cdiv(P * (C / 8), 256);
    hipLaunchKernelGGL(pack_n8_from_bf16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192), (unsigned)G), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
                       (uint4*)dst, N, (long long)P, C, ld);
    LT_CHECK_LAUNCH("lt_pack_n8_from_bf16");
    return LT_OK;
}

// partial sums of any kernel behind lt_conv_wgrad_bf16: at most 16 MiB, or one slab set of 256 workgroups for the brick kernels
extern "C" size_t lt_conv_wgrad_bf16_workspace(int64_t octet_rows, int32_t cout_pad, int32_t k_pad) {
    if (octet_rows < 1 || cout_pad < 1 || k_pad < 1) return 0;
    const size_t n = (size_t)cout_pad * k_pad * sizeof(float);
    const Plan16 p = plan16(octet_rows, cout_pad, k_pad);
    size_t need = p.S > 1 ? (size_t)p.S * n : 0;
    const size_t brick = (size_t)512 * n < ((size_t)64 << 20) ? (size_t)512 * n : ((size_t)64 << 20);
    if (brick > need) need = brick;
    return need;
}

extern "C" int lt_conv_wgrad_bf16(const void* dy16, const void* x16, const int32_t* taps, float* dw, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                                  int32_t Do, int32_t Ho, int32_t Wo, const int32_t stride[3], const int32_t pad[3], int32_t Cout, int32_t ldy,
                                  int32_t cout_pad, int32_t k_pad, int32_t ntaps, int32_t accumulate, void* workspace, void* stream) {
    LT_REQUIRE(dy16 && x16 && taps && dw && stride && pad, LT_ERR_INVALID, "lt_conv_wgrad_bf16: null argument");
    const int l2 = ilog2_exact(Cin);
    LT_REQUIRE(l2 >= 0, LT_ERR_UNSUPPORTED, "lt_conv_wgrad_bf16: Cin=%d must be a power of two", Cin);
    LT_REQUIRE(Cout >= 1 && ldy >= Cout && cout_pad >= Cout && k_pad >= ntaps * Cin && ntaps >= 1 && N >= 1, LT_ERR_INVALID, "lt_conv_wgrad_bf16: bad sizes");
    const int G = (int)cdiv(N, 8);
    const long long M = (long long)G * Do * Ho * Wo;
    LT_REQUIRE(M >= 1 && M < (1ll << 31) && (long long)G * D * H * W * Cin < (1ll << 31) && M * ldy < (1ll << 40), LT_ERR_UNSUPPORTED,
               "lt_conv_wgrad_bf16: too many rows / elements for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)cout_pad * k_pad;
    const size_t ws_bytes = lt_conv_wgrad_bf16_workspace(M, cout_pad, k_pad);
    const bool unit = stride[0] == 1 && stride[1] == 1 && stride[2] == 1 && D == Do && H == Ho && W == Wo;
    if (brick16_ok(D, H, W, Cin, Do, Ho, Wo, stride, pad, Cout, cout_pad, k_pad, ntaps)) {
        LT_REQUIRE(workspace, LT_ERR_INVALID, "lt_conv_wgrad_bf16: this shape needs a workspace of lt_conv_wgrad_bf16_workspace() bytes");
        Brick16Args b;
        b.dy = (const uint4*)dy16; b.x = (const uint4*)x16; b.taps = (const int4*)taps; b.out = (float*)workspace;
        b.D = D; b.H = H; b.W = W; b.Cin = Cin; b.ldy = ldy; b.cout_pad = cout_pad; b.k_pad = k_pad;
        b.cib = Cin < 32 ? Cin : 32; b.nblk = (int)cdiv(27 * b.cib, 32);
        b.nbd = D / B16_D; b.nbh = H / B16_H; b.nbw = W / B16_W; b.nbricks = G * b.nbd * b.nbh * b.nbw;
        const long long blocks = (long long)(Cout / 32) * (Cin / b.cib);
        long long S = cdiv(256, blocks);                  // one workgroup per CU (96 KB of LDS each): exactly one round
        const long long cap = (long long)(ws_bytes / ((size_t)n * 4));
        S = S > cap ? cap : S;
        S = S > b.nbricks ? b.nbricks : S;
        S = S < 1 ? 1 : S;
        b.bricks_per_slab = (int)cdiv(b.nbricks, S);
        S = cdiv(b.nbricks, b.bricks_per_slab);
        const size_t lds = (size_t)(B16_HVOX * b.cib + B16_VOX * 32) * 16;
        if (b.cib == 32) hipLaunchKernelGGL(conv3d_wgrad16_brick_kernel<32>, dim3(Cout / 32, Cin / 32, (unsigned)S), dim3(256), lds, st, b);
        else hipLaunchKernelGGL(conv3d_wgrad16_brick_kernel<16>, dim3(Cout / 32, 1, (unsigned)S), dim3(256), lds, st, b);
        LT_CHECK_LAUNCH("lt_conv_wgrad_bf16(brick)");
        reduce16(workspace, dw, n, (int)S, accumulate, st);
        LT_CHECK_LAUNCH("lt_conv_wgrad_bf16(reduce)");
        return LT_OK;
    }
    if (unit && ntaps == 343 && pad[0] == 3 && pad[1] == 3 && pad[2] == 3 && Cin == 32 && Cout == 16 && cout_pad == 16 && k_pad == 343 * 32 && ldy >= 16 &&
        H % K16_H == 0 && W % K16_W == 0 && workspace) {
        Brick16Args b;
        b.dy = (const uint4*)dy16; b.x = (const uint4*)x16; b.taps = (const int4*)taps; b.out = (float*)workspace;
        b.D = D; b.H = H; b.W = W; b.Cin = Cin; b.ldy = ldy; b.cout_pad = cout_pad; b.k_pad = k_pad; b.cib = 32; b.nblk = 0;
        b.nbd = D; b.nbh = H / K16_H; b.nbw = W / K16_W; b.nbricks = G * b.nbd * b.nbh * b.nbw;
        long long S = 73;                                  // 7 kd planes x 73 slabs = 511 workgroups, two per CU
        const long long cap = (long long)(ws_bytes / ((size_t)n * 4));
        S = S > cap ? cap : S;
        S = S > b.nbricks ? b.nbricks : S;
        S = S < 1 ? 1 : S;
        b.bricks_per_slab = (int)cdiv(b.nbricks, S);
        S = cdiv(b.nbricks, b.bricks_per_slab);
        hipLaunchKernelGGL(conv3d_wgrad16_k7_kernel, dim3(7, 1, (unsigned)S), dim3(256), 0, st, b);
        LT_CHECK_LAUNCH("lt_conv_wgrad_bf16(7^3)");
        reduce16(workspace, dw, n, (int)S, accumulate, st);
        LT_CHECK_LAUNCH("lt_conv_wgrad_bf16(reduce)");
        return LT_OK;
    }
    const Plan16 p = plan16(M, cout_pad, k_pad);
    LT_REQUIRE(p.S == 1 || workspace, LT_ERR_INVALID, "lt_conv_wgrad_bf16: this shape needs a workspace of lt_conv_wgrad_bf16_workspace() bytes");
    W16Args a;
    a.dy = (const uint4*)dy16; a.x = (const uint4*)x16; a.taps = (const int4*)taps; a.out = p.S > 1 ? (float*)workspace : dw;
    a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.log2Cin = l2; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.sd = stride[0]; a.sh = stride[1]; a.sw = stride[2]; a.pd = pad[0]; a.ph = pad[1]; a.pw = pad[2];
    a.Cout = Cout; a.ldy = ldy; a.k_pad = k_pad; a.ntaps = ntaps; a.M = (int)M; a.accumulate = accumulate; a.cout_pad = cout_pad;
    a.n_k_t = p.n_k_t; a.n_tiles = p.n_co_t * p.n_k_t; a.rows_per_slab = p.rows_per_slab;
    const dim3 grid((unsigned)cdiv(a.n_tiles, 4), (unsigned)p.S);
    if (p.variant == 0) hipLaunchKernelGGL((conv_wgrad16_kernel<4, 2, 3>), grid, dim3(256), 0, st, a);
    else if (p.variant == 1) hipLaunchKernelGGL((conv_wgrad16_kernel<2, 4, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad16_kernel<1, 8, 2>), grid, dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_conv_wgrad_bf16");
    if (p.S > 1) {
        reduce16(workspace, dw, n, p.S, accumulate, st);
        LT_CHECK_LAUNCH("lt_conv_wgrad_bf16(reduce)");
    }
    return LT_OK;
}

extern "C" int lt_conv_wgrad_bf16_nhwc_ok(int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Do, int32_t Ho, int32_t Wo, const int32_t stride[3],
                                          const int32_t pad[3], int32_t Cout, int32_t ldy, int32_t cout_pad, int32_t k_pad, int32_t ntaps) {
    if (!stride || !pad || N < 1 || ntaps < 1 || Cin < 1) return 0;
    if (getenv("LT_WGRAD16_PACKED")) return 0;
    const int G = (int)cdiv(N, 8);
    const long long M = (long long)G * Do * Ho * Wo;
    const bool unit = stride[0] == 1 && stride[1] == 1 && stride[2] == 1 && D == Do && H == Ho && W == Wo;
    // the V2V 3^3 layers: the LDS-brick kernel, staged from the channels-last tensors when a workgroup's 32 input channels are whole 8-channel chunks
    if (brick16_ok(D, H, W, Cin, Do, Ho, Wo, stride, pad, Cout, cout_pad, k_pad, ntaps))
        return Cin % 32 == 0 && ldx == Cin && ldy % 8 == 0 && !getenv("LT_WGRAD16_BRICK_PACKED") && (long long)N * D * H * W * Cin < (1ll << 31) &&
               (long long)N * D * H * W * ldy < (1ll << 31);
    if (unit && ntaps == 343 && Cin == 32 && Cout == 16) return 0;
    if (ilog2_exact(Cin) < 0 || k_pad % 4 || k_pad < ntaps * Cin || cout_pad < Cout || ldy < Cout || ldx < Cin) return 0;
    const int ach = cout_pad > 64 ? 8 : 4, bch = cout_pad > 64 ? 4 : 8;
    if (Cout % ach || ldy % ach || Cin % bch || ldx % bch) return 0;
    if (M < 1 || M >= (1ll << 31) || (long long)N * D * H * W * ldx >= (1ll << 31) || (long long)N * Do * Ho * Wo * ldy >= (1ll << 31)) return 0;
    for (int t = 0; t < 3; ++t)
        if (pad[t] < 0 || stride[t] < 1) return 0;
    return 1;
}

extern "C" int lt_conv_wgrad_bf16_nhwc(const void* dy16, const void* x16, const int32_t* taps, float* dw, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                                       int32_t ldx, int32_t Do, int32_t Ho, int32_t Wo, const int32_t stride[3], const int32_t pad[3], int32_t Cout, int32_t ldy,
                                       int32_t cout_pad, int32_t k_pad, int32_t ntaps, int32_t accumulate, void* workspace, void* stream) {
    LT_REQUIRE(dy16 && x16 && taps && dw && stride && pad, LT_ERR_INVALID, "lt_conv_wgrad_bf16_nhwc: null argument");
    LT_REQUIRE(lt_conv_wgrad_bf16_nhwc_ok(N, D, H, W, Cin, ldx, Do, Ho, Wo, stride, pad, Cout, ldy, cout_pad, k_pad, ntaps), LT_ERR_UNSUPPORTED,
               "lt_conv_wgrad_bf16_nhwc: shape not covered (ask lt_conv_wgrad_bf16_nhwc_ok; pack with lt_pack_n8_from_bf16 and call lt_conv_wgrad_bf16)");
    LT_REQUIRE((size_t)dy16 % 16 == 0 && (size_t)x16 % 16 == 0, LT_ERR_INVALID, "lt_conv_wgrad_bf16_nhwc: 16-byte aligned tensors");
    hipStream_t st = (hipStream_t)stream;
    const int G = (int)cdiv(N, 8);
    const long long M = (long long)G * Do * Ho * Wo, n = (long long)cout_pad * k_pad;
    if (brick16_ok(D, H, W, Cin, Do, Ho, Wo, stride, pad, Cout, cout_pad, k_pad, ntaps)) {
        LT_REQUIRE(workspace, LT_ERR_INVALID, "lt_conv_wgrad_bf16_nhwc: this shape needs a workspace of lt_conv_wgrad_bf16_workspace() bytes");
        const size_t ws_bytes = lt_conv_wgrad_bf16_workspace(M, cout_pad, k_pad);
        Brick16UArgs b;
        b.dy = (const unsigned short*)dy16; b.x = (const unsigned short*)x16; b.taps = (const int4*)taps; b.out = (float*)workspace;
        b.N = N; b.D = D; b.H = H; b.W = W; b.Cin = Cin; b.ldy = ldy; b.cout_pad = cout_pad; b.k_pad = k_pad;
        b.nbd = D / B16_D; b.nbh = H / B16_H; b.nbw = W / B16_W; b.nbricks = G * b.nbd * b.nbh * b.nbw;
        const long long blocks = (long long)(Cout / 32) * (Cin / 32);
        long long S = cdiv(256, blocks);                  // one workgroup per CU (96 KB of LDS each): exactly one round
        const long long cap = (long long)(ws_bytes / ((size_t)n * 4));
        S = S > cap ? cap : S;
        S = S > b.nbricks ? b.nbricks : S;
        S = S < 1 ? 1 : S;
        b.bricks_per_slab = (int)cdiv(b.nbricks, S);
        S = cdiv(b.nbricks, b.bricks_per_slab);
        const size_t lds = (size_t)(B16_HVOX * 32 + B16_VOX * 32) * 16;
        hipLaunchKernelGGL(conv3d_wgrad16_brick_u_kernel, dim3(Cout / 32, Cin / 32, (unsigned)S), dim3(256), lds, st, b);
        LT_CHECK_LAUNCH("lt_conv_wgrad_bf16_nhwc(brick)");
        reduce16(workspace, dw, n, (int)S, accumulate, st);
        LT_CHECK_LAUNCH("lt_conv_wgrad_bf16_nhwc(reduce)");
        return LT_OK;
    }
    const Plan16 p = plan16u(M, cout_pad, k_pad);
    LT_REQUIRE(p.S == 1 || workspace, LT_ERR_INVALID, "lt_conv_wgrad_bf16_nhwc: this shape needs a workspace of lt_conv_wgrad_bf16_workspace() bytes");
    W16UArgs a;
    a.dy = (const unsigned short*)dy16; a.x = (const unsigned short*)x16; a.taps = (const int4*)taps; a.out = p.S > 1 ? (float*)workspace : dw;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.log2Cin = ilog2_exact(Cin); a.ldx = ldx; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.sd = stride[0]; a.sh = stride[1]; a.sw = stride[2]; a.pd = pad[0]; a.ph = pad[1]; a.pw = pad[2];
    a.Cout = Cout; a.ldy = ldy; a.k_pad = k_pad; a.ntaps = ntaps; a.M = (int)M; a.accumulate = accumulate; a.cout_pad = cout_pad;
    a.n_k_t = p.n_k_t; a.n_tiles = p.n_co_t * p.n_k_t; a.rows_per_slab = p.rows_per_slab;
    a.img_a = (unsigned)((long long)Do * Ho * Wo * ldy); a.img_b = (unsigned)((long long)D * H * W * ldx);
    const dim3 grid((unsigned)a.n_tiles, (unsigned)p.S);
    const size_t lds = (size_t)3 * 128 * 64 * sizeof(float);          // the three other waves' 128 accumulator registers
    // Register sets in flight: three are 3 % faster on one stream (17.5 vs 18.1 ms of weight gradients per step) and slower in the step: the kernel runs
    // BESIDE the main stream's BatchNorm passes, one wave per SIMD -- at 448 allocated registers a 96-register wave of the BatchNorm-backward reduction
    // does not fit next to it (it ran 10.1 instead of 4.2 ms per step), at 400 it does: 140.1 -> 144.2 samples/s, same session.  LT_WGRAD16U_NS=3: the deep one.
    static const int ns = [] { const char* e = getenv("LT_WGRAD16U_NS"); return e ? atoi(e) : 2; }();
    if (ns == 2) {
        if (p.variant == 0) hipLaunchKernelGGL((conv_wgrad16u_kernel<8, 4, 2>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((conv_wgrad16u_kernel<4, 8, 2>), grid, dim3(256), lds, st, a);
    } else {
        if (p.variant == 0) hipLaunchKernelGGL((conv_wgrad16u_kernel<8, 4, 3>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((conv_wgrad16u_kernel<4, 8, 3>), grid, dim3(256), lds, st, a);
    }
    LT_CHECK_LAUNCH("lt_conv_wgrad_bf16_nhwc");
    if (p.S > 1) {
        reduce16(workspace, dw, n, p.S, accumulate, st);
        LT_CHECK_LAUNCH("lt_conv_wgrad_bf16_nhwc(reduce)");
    }
    return LT_OK;
}
