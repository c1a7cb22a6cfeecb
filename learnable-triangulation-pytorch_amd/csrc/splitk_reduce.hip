// lt_splitk_reduce: the second half of a convolution whose reduction (taps x Cin) was cut into S slices.
//
// V2V's 3^3 128 -> 128 convolutions at the 8^3 / 4^3 / 2^3 levels of the hourglass (mvn/models/v2v.py:78-90: encoder_res3-5, mid_res,
// decoder_res3-5 and their skip blocks, 18 launches) have K = 27 x 128 = 3456 and only B x 512 / 64 / 8 output rows: one or a handful of
// workgroups walk 54 K steps of a latency-bound ring, ~30 us per launch whatever the batch (15 % of the single-sample latency).  The
// host (lt_engine.PlanBuilder.conv) cuts the TAPS into S <= 8 groups and hands them to lt_conv_fwd as S *phases* -- the mechanism that
// runs the parity phases of a transposed convolution as grid.y -- each with its own tap table and weight slice, an identity epilogue
// and fp32 stores into slice p of a depth-stacked partial tensor [N][S * Do][Ho][Wo][Cout]; S x as many workgroups each walk 1 / S of
// the K steps.  This kernel then adds the S partial sums in a fixed order (p = 0 .. S-1, fp32) and applies the convolution's real
// epilogue (bias, folded BatchNorm, ReLU before / after the residual add) exactly as the conv kernels do.  No atomics, no tickets:
// two launches.  (Round 2 tried the cut inside one launch -- per-tile ticket counters, the last arriver adds -- and with L2 float
// atomics; both lost to the single workgroup that then has to pull every partial tile through one CU, DESIGN.md.)
#include "conv_common.h"

using namespace lt;

namespace {

struct SkrArgs {
    const float* part;       // [N][S][R][C] fp32 (R = Do * Ho * Wo rows per sample)
    const float* bias; const float* scale; const float* shift;   // [>= C] or null
    const void* res;         // [N][R][C] T or null
    void* y;                 // [N][R][C] T
    long long R, total4;     // rows per sample; N * R * C / 4
    int S, C, flags;
};

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SkrArgs a) {
    const EpiFloors fl = epi_floors(a.flags);
    const int c4n = a.C >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total4; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % c4n);
        const long long row = i / c4n;                     // n * R + r
        const long long n = row / a.R, r = row - n * a.R;
        const float4* p = (const float4*)(a.part + ((n * a.S) * a.R + r) * a.C) + cv;
        float4 s = p[0];
        for (int k = 1; k < a.S; ++k) {
            const float4 q = p[(long long)k * a.R * c4n];
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        const int c = cv * 4;
        float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bi = a.bias ? a.bias[c + e] : 0.f, sc = a.scale ? a.scale[c + e] : 1.f, sh = a.shift ? a.shift[c + e] : 0.f;
            const float rr = a.res ? elt<T>::ld((const T*)a.res + row * a.C + c + e) : -0.0f;
            v[e] = epi_apply((v[e] + bi) * sc + sh, fl, rr);
        }
        T* dst = (T*)a.y + row * a.C + c;
        if (sizeof(T) == 2) {
            *(uint2*)dst = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        } else {
            *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

}  // namespace

extern "C" int lt_splitk_reduce(int32_t dtype, const float* partial, int32_t S, int64_t N, int64_t rows_per_sample, int32_t C, const float* bias,
                                const float* scale, const float* shift, const void* residual, void* y, int32_t flags, void* stream) {
    LT_REQUIRE(partial && y && S >= 1 && N >= 1 && rows_per_sample >= 1 && C >= 4 && C % 4 == 0, LT_ERR_INVALID, "lt_splitk_reduce: bad argument (C %% 4 == 0)");
    LT_REQUIRE(dtype == LT_F32 || dtype == LT_BF16, LT_ERR_INVALID, "lt_splitk_reduce: bad dtype %d", dtype);
    LT_REQUIRE((flags & ~(LT_EPI_RELU_PRE | LT_EPI_RELU_POST)) == 0, LT_ERR_UNSUPPORTED, "lt_splitk_reduce: flags 0x%x (ReLU flags only)", flags);
    SkrArgs a;
    a.part = partial; a.bias = bias; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.R = rows_per_sample; a.total4 = N * rows_per_sample * (C / 4); a.S = S; a.C = C; a.flags = flags;
    const long long blocks = cdiv(a.total4, 256);
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192));
    if (dtype == LT_F32) hipLaunchKernelGGL(splitk_reduce_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    LT_CHECK_LAUNCH("lt_splitk_reduce");
    return LT_OK;
}
