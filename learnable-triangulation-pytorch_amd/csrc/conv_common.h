// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip = register-staged v1,
// conv_igemm2.hip = LDS-DMA staged v2 with the LDS-transposed vector epilogue).
#pragma once
#include "lt_common.h"

namespace lt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct PhaseArg {
    const void* w;
    const void* wfrag;   // the same weights in MFMA B-fragment order (lt_conv_pack_weights, layout 1), or null
    const void* wfrag_t; // ... in the order of the transposed product (lt_conv_pack_weights_t32, layout 2), or null
    const void* wfrag32; // ... in B-fragment order of the 32x32x16 MFMA (lt_conv_pack_weights32, layout 3), or null
    const int4* taps;
    int ntaps;
    int ood, ooh, oow;
};

struct ConvArgs {
    const void* x;
    void* y;
    const void* res;
    const float* bias;
    const float* scale;
    const float* shift;
    int N, D, H, W, Cin, log2Cin;
    int Do, Ho, Wo;
    int sd, sh, sw, pd, ph, pw;
    int OD, OH, OW, osd, osh, osw;
    int Cout, ldc, k_pad, flags;
    int M;        // N*Do*Ho*Wo
    int tiles_n;  // cout_pad / BN
    int stages;   // requested LDS-DMA ring depth (0 = auto)
    PhaseArg phase[LT_CONV_MAX_PHASES];
    const void* skip_x;   // lt_conv_skip_fwd: the residual is computed as W_skip . skip_x[voxel] (conv3d_halo_col_kernel only), else null
    const void* skip_w;
    // lt_conv_cat2_fwd (conv_igemm7 MODE 3): a 1x1 convolution over the channel concatenation of TWO tensors -- K steps below Cin / 32 read x, the rest read
    // x2 at pixel (n, oh * s2, ow * s2) of its H2 x W2 map (Cin2 channels per pixel); null = off
    const void* x2;
    int Cin2, H2, W2, s2;
};

union V16 {
    uint4 u;
    f32x4 f;
    bf16x8 h;
};

template <typename T, int MF> struct Mma;
template <> struct Mma<float, 32> {
    typedef f32x16 acc_t;
    static __device__ __forceinline__ void run(acc_t& c, const V16& a, const V16& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[e], b.f[e], c, 0, 0, 0);
    }
};
template <> struct Mma<float, 16> {
    typedef f32x4 acc_t;
    static __device__ __forceinline__ void run(acc_t& c, const V16& a, const V16& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[e], b.f[e], c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t, 32> {
    typedef f32x16 acc_t;
    static __device__ __forceinline__ void run(acc_t& c, const V16& a, const V16& b) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t, 16> {
    typedef f32x4 acc_t;
    static __device__ __forceinline__ void run(acc_t& c, const V16& a, const V16& b) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    }
};

// fp8 (e4m3): a 16-byte fragment vector holds 16 K elements = two 8-byte operands of v_mfma_f32_*_fp8_fp8 (K = 16 / 32 per instruction; the K
// order inside a fragment group differs from the bf16 kernels', identically for both operands, which a sum over K does not see)
union V16Q {
    uint4 u;
    long q[2];
};
template <> struct Mma<fp8_t, 32> {
    typedef f32x16 acc_t;
    static __device__ __forceinline__ void run(acc_t& c, const V16& a, const V16& b) {
        V16Q qa, qb;
        qa.u = a.u; qb.u = b.u;
        c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(qa.q[0], qb.q[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(qa.q[1], qb.q[1], c, 0, 0, 0);
    }
};
template <> struct Mma<fp8_t, 16> {
    typedef f32x4 acc_t;
    static __device__ __forceinline__ void run(acc_t& c, const V16& a, const V16& b) {
        V16Q qa, qb;
        qa.u = a.u; qb.u = b.u;
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(qa.q[0], qb.q[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(qa.q[1], qb.q[1], c, 0, 0, 0);
    }
};

// decode a GEMM row into (sample, od, oh, ow)
__device__ __forceinline__ void decode_row(const ConvArgs& a, int m, int& n, int& od, int& oh, int& ow) {
    int hw = a.Ho * a.Wo;
    int dhw = a.Do * hw;
    n = m / dhw;
    int r = m - n * dhw;
    od = r / hw;
    r -= od * hw;
    oh = r / a.Wo;
    ow = r - oh * a.Wo;
}

constexpr int ROW_BYTES = 128;  // K bytes per tile row per step
constexpr int LT_EPI_NO_RES_PREFETCH = 1 << 16;   // internal A/B switch (env LT_CONV_NO_RESPF), not part of the ABI
constexpr int LT_EPI_NO_XCD_REMAP = 1 << 17;      // internal A/B switch (env LT_CONV_NO_XCD)

// launchers implemented in conv_igemm2.hip, used by the dispatcher in conv_igemm.hip
int conv2_dispatch(int dtype, const ConvArgs& a, int cout_pad, int nphase, int max_taps, int tile, hipStream_t s);
// conv_igemm3.hip (288-row tile, 8 waves): 1 = launched, 0 = not applicable (fall back), < 0 = error
int conv3_try(int dtype, const ConvArgs& a, int cout_pad, int nphase, int max_taps, bool forced, hipStream_t s);
// conv_igemm7.hip (288 x 256 tile on 32x32x16 MFMAs, weights in layout 3): 1 / 0 / < 0 as above
int conv7_try(const ConvArgs& a, int cout_pad, int max_taps, bool pw, hipStream_t s);
// conv_pw.hip (streaming kernel for single-tap phases: 1x1x1 convs, 2x2x2 stride-2 deconvs): 1 / 0 / < 0 as above
int conv_pw_try(int dtype, const ConvArgs& a, int cout_pad, int nphase, hipStream_t s);
// conv2d_halo.hip (3x3 256 -> 256 on 24-wide maps, input halo in LDS, weights in layout 2): 1 / 0 / < 0 as above
int conv2d_halo_try(int dtype, const ConvArgs& a, int cout_pad, int nphase, hipStream_t s);
// conv3d_halo.hip: 1 = launched, 0 = not applicable (fall back), < 0 = error
int conv3d_halo_try(int dtype, const ConvArgs& a, int cout_pad, int nphase, bool forced, hipStream_t s);

}  // namespace lt
