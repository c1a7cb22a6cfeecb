// Channels-last max pooling and the two layout conversions at the API boundary.
// All three are HBM-bound streaming kernels: 16-byte (or widest possible) accesses along C.
#include "lt_common.h"

using namespace lt;

namespace {

// One thread per (output pixel, 16-byte channel vector).  Padding never wins (F.max_pool semantics).
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int D, int H, int W, int C, int Do,
                               int Ho, int Wo, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw) {
    constexpr int VEC = elt<T>::vec;
    const int cv = C / VEC;
    const long long total = (long long)N * Do * Ho * Wo * cv;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(g % cv) * VEC;
        long long r = g / cv;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int od = (int)(r % Do);
        const int n = (int)(r / Do);
        float m[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
        for (int a = 0; a < kd; ++a) {
            const int id = od * sd - pd + a;
            if ((unsigned)id >= (unsigned)D) continue;
            for (int b = 0; b < kh; ++b) {
                const int ih = oh * sh - ph + b;
                if ((unsigned)ih >= (unsigned)H) continue;
                for (int cc = 0; cc < kw; ++cc) {
                    const int iw = ow * sw - pw + cc;
                    if ((unsigned)iw >= (unsigned)W) continue;
                    const T* p = x + ((((long long)n * D + id) * H + ih) * W + iw) * C + c;
                    uint4 raw = *(const uint4*)p;
                    const T* e8 = (const T*)&raw;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], elt<T>::ld(e8 + e));
                }
            }
        }
        uint4 out;
        T* o8 = (T*)&out;
#pragma unroll
        for (int e = 0; e < VEC; ++e) elt<T>::st(o8 + e, m[e]);
        *(uint4*)(y + ((((long long)n * Do + od) * Ho + oh) * Wo + ow) * C + c) = out;
    }
}

// N,C,HW fp32 -> N,HW,c_pad T.  Reads are coalesced along HW per channel, the pixel's padded channel
// vector is written as one contiguous run (c_pad*sizeof(T) bytes per thread, 16 B for the 3->4/8 stem case).
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int HW, int c_pad) {
    const long long total = (long long)N * HW;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(g / HW);
        const int p = (int)(g - (long long)n * HW);
        T* o = y + g * c_pad;
        for (int c = 0; c < c_pad; ++c) {
            const float v = c < C ? x[((long long)n * C + c) * HW + p] : 0.f;
            elt<T>::st(o + c, v);
        }
    }
}

// N,HW,(ld) T -> N,C,HW fp32 through a 64-pixel x C LDS tile so that both sides are coalesced.
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int C, int HW, int ld) {
    extern __shared__ float tile[];  // [64][C+1]
    const int tiles = (HW + 63) / 64;
    const int n = blockIdx.x / tiles;
    const int p0 = (blockIdx.x % tiles) * 64;
    const int np = min(64, HW - p0);
    for (int i = threadIdx.x; i < np * C; i += blockDim.x) {
        const int p = i / C, c = i - p * C;
        tile[p * (C + 1) + c] = elt<T>::ld(x + ((long long)n * HW + p0 + p) * ld + c);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 64; i += blockDim.x) {
        const int c = i >> 6, p = i & 63;
        if (p < np) y[((long long)n * C + c) * HW + p0 + p] = tile[p * (C + 1) + c];
    }
}

// mean over HW per (sample, channel): one workgroup per (n, 64-channel slab); lanes along C so every row
// read is contiguous, 4 row-groups reduced through LDS.
template <typename T>
__global__ __launch_bounds__(256) void global_avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int HW, int C) {
    __shared__ float red[4][64];
    const int slabs = (C + 63) / 64;
    const int n = blockIdx.x / slabs, c = (blockIdx.x % slabs) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int p = g; p < HW; p += 4) s += elt<T>::ld(x + ((long long)n * HW + p) * C + c);
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < C) {
        const int l = threadIdx.x & 63;
        elt<T>::st(y + (long long)n * C + c, (red[0][l] + red[1][l] + red[2][l] + red[3][l]) / (float)HW);
    }
}

inline unsigned grid_for(long long total, int block) {
    long long b = cdiv(total, block);
    const long long cap = 256 * 16;  // 256 CUs x a few waves; grid-stride the rest
    return (unsigned)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace

extern "C" int lt_maxpool_fwd(int32_t dtype, const void* x, void* y, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C,
                              const int32_t k[3], const int32_t s[3], const int32_t p[3], void* stream) {
    LT_REQUIRE(x && y && k && s && p, LT_ERR_INVALID, "lt_maxpool_fwd: null argument");
    LT_REQUIRE(dtype == LT_F32 || dtype == LT_BF16, LT_ERR_INVALID, "lt_maxpool_fwd: bad dtype");
    const int vec = dtype == LT_F32 ? 4 : 8;
    LT_REQUIRE(C % vec == 0, LT_ERR_UNSUPPORTED, "lt_maxpool_fwd: C=%d must be a multiple of %d", C, vec);
    for (int i = 0; i < 3; ++i) LT_REQUIRE(k[i] >= 1 && s[i] >= 1 && p[i] >= 0 && 2 * p[i] <= k[i], LT_ERR_INVALID, "lt_maxpool_fwd: bad window");
    const int Do = (D + 2 * p[0] - k[0]) / s[0] + 1, Ho = (H + 2 * p[1] - k[1]) / s[1] + 1, Wo = (W + 2 * p[2] - k[2]) / s[2] + 1;
    LT_REQUIRE(Do >= 1 && Ho >= 1 && Wo >= 1, LT_ERR_INVALID, "lt_maxpool_fwd: output size is too small");
    const long long total = (long long)N * Do * Ho * Wo * (C / vec);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LT_F32)
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const float*)x, (float*)y, N, D, H, W,
                           C, Do, Ho, Wo, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2]);
    else
        hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, D,
                           H, W, C, Do, Ho, Wo, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2]);
    LT_CHECK_LAUNCH("lt_maxpool_fwd");
    return LT_OK;
}

extern "C" int lt_nchw_to_nhwc(int32_t dtype, const float* x, void* y, int32_t N, int32_t C, int32_t HW, int32_t c_pad, void* stream) {
    LT_REQUIRE(x && y && c_pad >= C && C >= 1, LT_ERR_INVALID, "lt_nchw_to_nhwc: bad argument");
    const long long total = (long long)N * HW;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LT_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, x, (float*)y, N, C, HW, c_pad);
    else if (dtype == LT_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, x, (bf16_t*)y, N, C, HW, c_pad);
    else LT_REQUIRE(false, LT_ERR_INVALID, "lt_nchw_to_nhwc: bad dtype");
    LT_CHECK_LAUNCH("lt_nchw_to_nhwc");
    return LT_OK;
}

extern "C" int lt_nhwc_to_nchw_f32(int32_t dtype, const void* x, float* y, int32_t N, int32_t C, int32_t HW, int32_t ld, void* stream) {
    LT_REQUIRE(x && y && ld >= C && C >= 1 && C <= 2048, LT_ERR_INVALID, "lt_nhwc_to_nchw_f32: bad argument");
    const long long blocks = (long long)N * ((HW + 63) / 64);
    LT_REQUIRE(blocks < (1ll << 31), LT_ERR_UNSUPPORTED, "lt_nhwc_to_nchw_f32: too large");
    const size_t lds = (size_t)64 * (C + 1) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LT_F32) {
        LT_OPT_IN_LDS(nhwc_to_nchw_kernel<float>, 160 * 1024);
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, st, (const float*)x, y, N, C, HW, ld);
    } else if (dtype == LT_BF16) {
        LT_OPT_IN_LDS(nhwc_to_nchw_kernel<bf16_t>, 160 * 1024);
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), lds, st, (const bf16_t*)x, y, N, C, HW, ld);
    } else LT_REQUIRE(false, LT_ERR_INVALID, "lt_nhwc_to_nchw_f32: bad dtype");
    LT_CHECK_LAUNCH("lt_nhwc_to_nchw_f32");
    return LT_OK;
}

extern "C" int lt_global_avgpool(int32_t dtype, const void* x, void* y, int32_t N, int32_t HW, int32_t C, void* stream) {
    LT_REQUIRE(x && y && N >= 1 && HW >= 1 && C >= 1, LT_ERR_INVALID, "lt_global_avgpool: bad argument");
    const unsigned grid = (unsigned)(N * ((C + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LT_F32) hipLaunchKernelGGL(global_avgpool_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, HW, C);
    else if (dtype == LT_BF16) hipLaunchKernelGGL(global_avgpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, HW, C);
    else LT_REQUIRE(false, LT_ERR_INVALID, "lt_global_avgpool: bad dtype");
    LT_CHECK_LAUNCH("lt_global_avgpool");
    return LT_OK;
}
