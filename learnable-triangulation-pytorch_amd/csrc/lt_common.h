// Shared host/device helpers for liblt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lt_hip.h"

namespace lt {

void set_error(const char* fmt, ...);

#define LT_REQUIRE(cond, code, ...)        \
    do {                                   \
        if (!(cond)) {                     \
            lt::set_error(__VA_ARGS__);    \
            return (code);                 \
        }                                  \
    } while (0)

// Checks the launch that was just enqueued (no synchronisation).
#define LT_CHECK_LAUNCH(what)                                                            \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            lt::set_error("%s: launch failed: %s", (what), hipGetErrorString(e_));       \
            return LT_ERR_LAUNCH;                                                        \
        }                                                                                \
    } while (0)

// Large dynamic LDS is an opt-in per kernel function AND per device: remembered per call site in a bitmask over the device ids
// (a process-wide bool left every device but the first without the attribute: launches needing > 64 KB failed there), the return
// code checked.  Cheap enough for the recording / eager paths (hipGetDevice); graph replays do not come through here.
int current_device();
#define LT_OPT_IN_LDS(kern, bytes)                                                                                  \
    do {                                                                                                            \
        static std::atomic<unsigned long long> done_{0};                                                            \
        const unsigned long long bit_ = 1ull << (lt::current_device() & 63);                                        \
        if (!(done_.load(std::memory_order_acquire) & bit_)) {                                                      \
            hipError_t e_ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
            if (e_ != hipSuccess) {                                                                                 \
                lt::set_error("hipFuncSetAttribute(max dynamic LDS %d): %s", (int)(bytes), hipGetErrorString(e_));  \
                return LT_ERR_LAUNCH;                                                                               \
            }                                                                                                       \
            done_.fetch_or(bit_, std::memory_order_release);                                                        \
        }                                                                                                           \
    } while (0)
// compute units of the current device, rounded down to a multiple of 8 (XCDs); cached per device id
int device_cu_count8();

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even, NaN preserved (same rounding as torch's float -> bfloat16)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16 pair (lo in bits 0-15) with one v_cvt_pk_bf16_f32: round-to-nearest-even like torch; a NaN comes
// out as a quiet NaN (payload not preserved).  The software f32_to_bf16 above costs ~12 VALU instructions per value: in the
// vector epilogues that was most of ~2400 instructions per tile (shader-clock accounting of the persistent 3^3 kernel).
typedef __bf16 lt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    lt_f32x2 f = {lo, hi};
    lt_bf16x2 b = __builtin_convertvector(f, lt_bf16x2);
    return __builtin_bit_cast(unsigned, b);
}

// The activation part of the conv epilogues with the ReLU flags turned into floors: v = max(max(v, pre) + r, post) with
// floor = 0 when that ReLU is on and -inf when it is off (max(v, -inf) == v; one v_max instead of v_max + v_cndmask), and
// r = -0.0f when there is no residual (v + -0.0 == v for every v, signed zeros included).  A NaN v does not survive a
// floor (maxNum semantics), exactly as it did not survive an active ReLU before.
struct EpiFloors {
    float pre, post;
};
__device__ __forceinline__ EpiFloors epi_floors(int flags) {
    EpiFloors f;
    f.pre = (flags & LT_EPI_RELU_PRE) ? 0.f : -__builtin_inff();
    f.post = (flags & LT_EPI_RELU_POST) ? 0.f : -__builtin_inff();
    return f;
}
__device__ __forceinline__ float epi_apply(float v, const EpiFloors& f, float r) { return fmaxf(fmaxf(v, f.pre) + r, f.post); }

template <typename T> struct elt;
template <> struct elt<float> {
    static constexpr int bytes = 4;
    static constexpr int vec = 4;  // elements per 16 B
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct elt<bf16_t> {
    static constexpr int bytes = 2;
    static constexpr int vec = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// OCP e4m3 operand bytes of lt_conv_fwd(dtype = LT_FP8); never an output type (the fp8 convolutions store fp32), so ld / st only exist for
// the generic code paths that are instantiated but not reachable
struct fp8_t { unsigned char v; };
template <> struct elt<fp8_t> {
    static constexpr int bytes = 1;
    static constexpr int vec = 16;
    __device__ static __forceinline__ float ld(const fp8_t*) { return 0.f; }
    __device__ static __forceinline__ void st(fp8_t*, float) {}
};

// a BatchNorm input that is either fp32 or bf16 (LT_BN_Y_BF16: the mixed-precision training step stores its convolution outputs in bf16)
__device__ __forceinline__ float4 ld4_f32_or_bf16(const void* p, size_t off, int is_bf16) {
    if (is_bf16) {
        const uint2 u = *(const uint2*)((const bf16_t*)p + off);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    return *(const float4*)((const float*)p + off);
}
__device__ __forceinline__ float ld1_f32_or_bf16(const void* p, size_t off, int is_bf16) {
    return is_bf16 ? __uint_as_float((unsigned)((const unsigned short*)p)[off] << 16) : ((const float*)p)[off];
}

__device__ __forceinline__ float4 bf16x4_to_f32(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

// stores of the same kind (LT_ACT_BF16: the activations and activation gradients of the 16-bit training step are bf16 tensors)
__device__ __forceinline__ void st4_f32_or_bf16(void* p, size_t off, int is_bf16, float a, float b, float c, float d) {
    if (is_bf16) *(uint2*)((bf16_t*)p + off) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
    else *(float4*)((float*)p + off) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void st1_f32_or_bf16(void* p, size_t off, int is_bf16, float v) {
    if (is_bf16) ((bf16_t*)p)[off] = (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
    else ((float*)p)[off] = v;
}

static inline int ilog2_exact(int v) {  // -1 when v is not a power of two
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace lt
