// Pieces shared by the weight-gradient kernels (train.hip: exact-fp32 MFMA; wgrad16.hip: bf16 MFMA over image-octet operands): the
// deterministic reduction of the per-slab partial sums and the slab-count helpers.
#pragma once
#include "lt_common.h"

namespace {

// dw = (accumulate ? dw : 0) + sum over the S slab partials, slabs added in ascending order (fixed order: deterministic).  A thread owns four
// consecutive elements and keeps four slabs' loads in flight: with one scalar load per slab in a dependent chain this pass was latency-bound
// (14 ms of a training step for ~200 launches whose traffic is worth 3 ms).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long n, int S, int accumulate) {
    if ((n & 3) == 0) {
        const long long n4 = n >> 2;
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += 256ll * gridDim.x) {
            float4 s = accumulate ? ((const float4*)dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* src = (const float4*)ws + i;
            int z = 0;
            for (; z + 4 <= S; z += 4) {
                const float4 a = src[(size_t)z * n4], b = src[(size_t)(z + 1) * n4], c = src[(size_t)(z + 2) * n4], d = src[(size_t)(z + 3) * n4];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
                s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
                s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
            }
            for (; z < S; ++z) {
                const float4 a = src[(size_t)z * n4];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            ((float4*)dw)[i] = s;
        }
        return;
    }
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        float s = accumulate ? dw[i] : 0.f;
        for (int z = 0; z < S; ++z) s += ws[(size_t)z * n + i];
        dw[i] = s;
    }
}

inline unsigned reduce_grid(long long n) {
    const long long b = lt::cdiv((n & 3) == 0 ? n >> 2 : n, 256);
    return (unsigned)(b < 1 ? 1 : b < 8192 ? b : 8192);
}


// a whole number of workgroups per CU where possible (320 workgroups on 256 CUs take as long as 512)
inline long long balance_slabs(long long blocks, long long S) {
    const long long wgs = blocks * S;
    if (wgs > 256 && blocks <= 256) {
        const long long r = (wgs / 256) * 256 / blocks;
        if (r >= 1) return r;
    }
    return S;
}


}  // namespace
