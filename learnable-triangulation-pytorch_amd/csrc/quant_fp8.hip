// fp8 (OCP e4m3) operands for lt_conv_fwd(dtype = LT_FP8): BASELINE config 5 names "fp8 MFMA for the V2V 3D convolutions" of the training step
// (reference loop train.py:154-243, V2V mvn/models/v2v.py:141-169; the reference itself trains in fp32).  Per-tensor amax scaling: scale = amax / 448,
// q = rne(x / scale) with v_cvt_pk_fp8_f32 (gfx950 converts to the OCP format, the one torch.float8_e4m3fn describes).  Everything stays on the
// device: the amax, the scale and the per-layer scale product are read by the next kernel from device memory (no host synchronisation, the
// training step is recorded once and replayed).
#include "lt_common.h"

using namespace lt;

namespace {

constexpr float FP8_MAX = 448.f;

// max |x| of non-negative floats == max of their bit patterns as unsigned integers: an integer atomic, exact and independent of the order
__global__ __launch_bounds__(256) void amax_kernel(const void* __restrict__ x, int x16, long long n, unsigned* __restrict__ amax) {
    float m = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = ld4_f32_or_bf16(x, (size_t)i * 4, x16);
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(ld1_f32_or_bf16(x, (size_t)i, x16)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // ONE atomic per workgroup (round 4: one per wave from up to 4096 workgroups serialised 16 K atomics on one address -- 64 us per call, 10 ms per step)
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(amax, __float_as_uint(m));   // NaN never compares greater: a NaN input leaves the maximum of the rest
    }
}

__device__ __forceinline__ float inv_scale_of(float amax, float& scale) {
    scale = amax > 0.f ? amax / FP8_MAX : 1.f;
    return 1.f / scale;
}

// four floats -> four e4m3 bytes (v_cvt_pk_fp8_f32 x 2)
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

__global__ __launch_bounds__(256) void quant_fp8_kernel(const void* __restrict__ x, int x16, unsigned char* __restrict__ q, long long n, const float* __restrict__ amax,
                                                        float* __restrict__ scale_out) {
    float scale;
    const float inv = inv_scale_of(*amax, scale);
    if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) *scale_out = scale;
    const long long n16 = n >> 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        const float4 v0 = ld4_f32_or_bf16(x, (size_t)i * 16, x16), v1 = ld4_f32_or_bf16(x, (size_t)i * 16 + 4, x16), v2 = ld4_f32_or_bf16(x, (size_t)i * 16 + 8, x16),
                     v3 = ld4_f32_or_bf16(x, (size_t)i * 16 + 12, x16);
        ((uint4*)q)[i] = make_uint4(pack4_fp8(v0.x * inv, v0.y * inv, v0.z * inv, v0.w * inv), pack4_fp8(v1.x * inv, v1.y * inv, v1.z * inv, v1.w * inv),
                                    pack4_fp8(v2.x * inv, v2.y * inv, v2.z * inv, v2.w * inv), pack4_fp8(v3.x * inv, v3.y * inv, v3.z * inv, v3.w * inv));
    }
    for (long long i = (n16 << 4) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        q[i] = (unsigned char)(pack4_fp8(ld1_f32_or_bf16(x, (size_t)i, x16) * inv, 0.f, 0.f, 0.f) & 0xff);
}

__global__ __launch_bounds__(256) void gather_fp8_kernel(const float* __restrict__ src, const int* __restrict__ idx, unsigned char* __restrict__ q, long long n,
                                                         const float* __restrict__ amax, float* __restrict__ scale_out) {
    float scale;
    const float inv = inv_scale_of(*amax, scale);
    if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) *scale_out = scale;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const int4 j = ((const int4*)idx)[i];
        const float a = j.x >= 0 ? src[j.x] * inv : 0.f, b = j.y >= 0 ? src[j.y] * inv : 0.f, c = j.z >= 0 ? src[j.z] * inv : 0.f, d = j.w >= 0 ? src[j.w] * inv : 0.f;
        ((unsigned*)q)[i] = pack4_fp8(a, b, c, d);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        q[i] = (unsigned char)(pack4_fp8(idx[i] >= 0 ? src[idx[i]] * inv : 0.f, 0.f, 0.f, 0.f) & 0xff);
}

__global__ void scale_product_kernel(float* __restrict__ dst, int n, const float* __restrict__ a, const float* __restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = *a * *b;
}

unsigned grid_for(long long work_items) {
    const long long b = cdiv(work_items, 256);
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
unsigned amax_grid(long long work_items) {          // a few workgroups per CU, each with a long grid-stride loop and ONE atomic at its end
    const long long b = cdiv(work_items, 256 * 8);
    return (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" int lt_amax_f32(const float* x, int64_t n, float* amax, void* stream) { return lt_amax_dt(LT_F32, x, n, amax, stream); }

extern "C" int lt_amax_dt(int32_t dtype, const void* x, int64_t n, float* amax, void* stream) {
    LT_REQUIRE(x && amax && n >= 1 && ((size_t)x % 16 == 0) && (dtype == LT_F32 || dtype == LT_BF16), LT_ERR_INVALID, "lt_amax: bad argument (16-byte aligned x, fp32 or bf16)");
    hipLaunchKernelGGL(amax_kernel, dim3(amax_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, dtype == LT_BF16 ? 1 : 0, (long long)n, (unsigned*)amax);
    LT_CHECK_LAUNCH("lt_amax");
    return LT_OK;
}

extern "C" int lt_quant_fp8(const float* x, void* q, int64_t n, const float* amax, float* scale_out, void* stream) {
    return lt_quant_fp8_dt(LT_F32, x, q, n, amax, scale_out, stream);
}

extern "C" int lt_quant_fp8_dt(int32_t dtype, const void* x, void* q, int64_t n, const float* amax, float* scale_out, void* stream) {
    LT_REQUIRE(x && q && amax && n >= 1 && ((size_t)x % 16 == 0) && ((size_t)q % 16 == 0) && (dtype == LT_F32 || dtype == LT_BF16), LT_ERR_INVALID,
               "lt_quant_fp8: bad argument (16-byte aligned pointers, fp32 or bf16 source)");
    hipLaunchKernelGGL(quant_fp8_kernel, dim3(grid_for(n / 16 + 1)), dim3(256), 0, (hipStream_t)stream, x, dtype == LT_BF16 ? 1 : 0, (unsigned char*)q, (long long)n, amax,
                       scale_out);
    LT_CHECK_LAUNCH("lt_quant_fp8");
    return LT_OK;
}

extern "C" int lt_gather_f32_fp8(const float* src, const int32_t* idx, void* q, int64_t n, const float* amax, float* scale_out, void* stream) {
    LT_REQUIRE(src && idx && q && amax && n >= 1 && ((size_t)idx % 16 == 0) && ((size_t)q % 4 == 0), LT_ERR_INVALID, "lt_gather_f32_fp8: bad argument");
    hipLaunchKernelGGL(gather_fp8_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src, idx, (unsigned char*)q, (long long)n, amax, scale_out);
    LT_CHECK_LAUNCH("lt_gather_f32_fp8");
    return LT_OK;
}

extern "C" int lt_scale_product(float* dst, int32_t n, const float* a, const float* b, void* stream) {
    LT_REQUIRE(dst && a && b && n >= 1, LT_ERR_INVALID, "lt_scale_product: bad argument");
    hipLaunchKernelGGL(scale_product_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dst, n, a, b);
    LT_CHECK_LAUNCH("lt_scale_product");
    return LT_OK;
}
