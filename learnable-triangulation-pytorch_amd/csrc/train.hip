// Training-step kernels around the convolutions (BASELINE config 5), fp32, channels-last:
//   lt_bn_act_fwd      z = act(gamma * (y - mean) * invstd + beta, residual)          train-mode BatchNorm + ReLU (+ residual), reference
//                      pose_resnet.py:75-95 / v2v.py:7-66 with the modules in training mode (batch statistics from lt_bn_stats_fwd)
//   lt_bn_act_bwd      autograd of the above: g = dz * relu-mask, dbeta = sum g, dgamma = sum g x^, dy = gamma invstd (g - dbeta/n - x^ dgamma/n),
//                      and the gradient of the residual input (accumulated into the buffer the producer of the residual will read)
//   lt_act_bwd         layers without BatchNorm (process_features, output_layer): dz -> dy through the ReLU flags / residual
//   lt_channel_sum     bias gradient: sum over rows of a rows x C tensor (fp64 accumulation)
//   lt_maxpool_bwd     scatter of dy to the FIRST maximal element of each window (torch's max_pool backward), accumulating
//   lt_conv_wgrad      dW[co][tap * Cin + ci] = sum_pixels dY[pix][co] * X[pix * stride + tap - pad][ci] on the exact-fp32 MFMA
//                      (v_mfma_f32_32x32x2_f32: its A/B operands are ONE value per lane with the lanes along the 32 output rows / columns,
//                      so channels-last dY / X rows feed it with fully coalesced 128-byte loads and no transposition -- the 16-bit MFMAs
//                      want 8 consecutive K = 8 consecutive PIXELS per lane, i.e. a transposed operand).  K here is the pixel index.
//                      Kernels behind it, chosen by layer shape: conv_wgrad_kernel<CT, KT> (generic: operands straight from L2, branch-free
//                      4-stage software pipeline, pixel slabs + deterministic reduce), conv3d_wgrad_brick_kernel (3^3 / stride 1: dY brick
//                      and X halo brick in LDS, taps split over the waves), conv3d_wgrad_k7_kernel (7^3 32 -> 16: one filter plane per
//                      workgroup on the 16x16x4 MFMA), conv2d_wgrad_brick_kernel (2D 3x3: 8 x 8 pixel bricks), wgrad_pw_kernel (1x1: LDS GEMM)
//   lt_gather_f32(_multi)  dst[i] = src[idx[i]]: live Parameters -> GEMM layouts, weight-gradient blocks -> Parameter layout
//   lt_cast_f32_bf16   bf16 copies of the convolution operands of the mixed-precision step (lt_bn_act_fwd / _bwd can write them on the way)
//   lt_adam_step(_multi)  torch.optim.Adam's update (train.py:430-437): one launch per tensor / one launch per parameter group
// The convolution dgrad needs no kernel of its own: it is lt_conv_fwd over dY with the weights transposed / flipped (stride 1), as a
// parity-phase transposed convolution (stride-2 layers) or as a strided convolution (the transposed layers) -- see lt_train.py.
#include <type_traits>

#include "colsum.h"
#include "conv_common.h"
#include "wgrad_reduce.h"

using namespace lt;

namespace {

__device__ __forceinline__ float relu_mask_post(float v) { return v > 0.f ? 1.f : 0.f; }

struct BnActArgs {
    const void* y;         // rows x C: convolution output (bias included), fp32 or (y16) bf16
    const float* mean; const float* var; const float* gamma; const float* beta;
    const void* res;       // rows x C or null (a16: bf16)
    void* z;               // fp32, or (a16) bf16
    bf16_t* z16;           // optional bf16 copy of z (mixed-precision training: the next convolution's operand), or null
    float eps;
    int flags, C, y16, a16;
    long long rows;
};

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const BnActArgs a) {
    const int c4n = a.C >> 2;
    const long long total = a.rows * c4n;
    const EpiFloors fl = epi_floors(a.flags);
    // the grid's stride is a multiple of C / 4 (launch): a thread keeps its four channels, whose constants -- the sqrt and the division of
    // invstd above all, which cost more than the rest of an element's work -- are computed once
    const long long g0 = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    const int c = (int)(g0 % c4n) * 4;
    float invstd[4], mean[4], gam[4], bet[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        invstd[e] = 1.0f / sqrtf(a.var[c + e] + a.eps);          // ATen's batch_norm (training): invstd = 1 / sqrt(var + eps) in fp32
        mean[e] = a.mean[c + e]; gam[e] = a.gamma[c + e]; bet[e] = a.beta[c + e];
    }
    for (long long g = g0; g < total; g += stride) {
        const size_t off = (size_t)g * 4;          // row * C + c
        const float4 y4 = ld4_f32_or_bf16(a.y, off, a.y16);
        const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
        float rr[4] = {-0.0f, -0.0f, -0.0f, -0.0f};
        if (a.res) { const float4 r4 = ld4_f32_or_bf16(a.res, off, a.a16); rr[0] = r4.x; rr[1] = r4.y; rr[2] = r4.z; rr[3] = r4.w; }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (yy[e] - mean[e]) * invstd[e] * gam[e] + bet[e];          // (x - mean) * invstd * weight + bias
            o[e] = epi_apply(v, fl, rr[e]);
        }
        st4_f32_or_bf16(a.z, off, a.a16, o[0], o[1], o[2], o[3]);
        if (a.z16) *(uint2*)(a.z16 + off) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
}

struct BnBwdArgs {
    const void* dz; const void* y; const void* res;        // rows x C (y: fp32 or, with y16, bf16; dz / res / dy / dres: fp32 or, with a16, bf16)
    const float* mean; const float* var; const float* gamma; const float* beta;
    double* part;            // [nslab][C][2]: sum g, sum g x^
    float* dgamma; float* dbeta;
    void* dy;                // rows x C
    bf16_t* dy16;            // optional bf16 copy of dy (mixed-precision training: the input-gradient convolution's operand), or null
    void* dres;              // rows x C or null: gradient of the residual input (accumulated when accumulate_res)
    float eps;
    int flags, C, nslab, accumulate_res, y16, a16;
    long long rows;
};

// g = dz * mask: RELU_POST: mask = [v + res > 0]; RELU_PRE: mask = [v > 0] (the residual is added after the ReLU); none: 1
__device__ __forceinline__ float bn_g(const BnBwdArgs& a, float dz, float y, float r, int c, float& xhat) {
    const float invstd = 1.0f / sqrtf(a.var[c] + a.eps);
    xhat = (y - a.mean[c]) * invstd;
    const float v = xhat * a.gamma[c] + a.beta[c];
    float m = 1.f;
    if (a.flags & LT_EPI_RELU_POST) m = (v + r) > 0.f ? 1.f : 0.f;
    else if (a.flags & LT_EPI_RELU_PRE) m = v > 0.f ? 1.f : 0.f;
    return dz * m;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdArgs a) {
    __shared__ double ss[256], sq[256];
    const int slab = blockIdx.x;
    const long long r0 = a.rows * slab / a.nslab, r1 = a.rows * (slab + 1) / a.nslab;
    const int Cw = a.C < 256 ? a.C : 256, RL = 256 / Cw;
    const int rl = threadIdx.x / Cw, cl = threadIdx.x - rl * Cw;
    for (int c0 = 0; c0 < a.C; c0 += Cw) {
        const int c = c0 + cl;
        double s = 0.0, q = 0.0;
        if (rl < RL && c < a.C)
            for (long long r = r0 + rl; r < r1; r += RL) {
                const size_t off = (size_t)r * a.C + c;
                float xh;
                const float g = bn_g(a, ld1_f32_or_bf16(a.dz, off, a.a16), ld1_f32_or_bf16(a.y, off, a.y16), a.res ? ld1_f32_or_bf16(a.res, off, a.a16) : 0.f, c, xh);
                s += (double)g; q += (double)g * (double)xh;
            }
        ss[threadIdx.x] = s; sq[threadIdx.x] = q;
        __syncthreads();
        if (rl == 0 && c < a.C) {
            for (int k = 1; k < RL; ++k) { s += ss[k * Cw + cl]; q += sq[k * Cw + cl]; }
            a.part[((long long)slab * a.C + c) * 2] = s;
            a.part[((long long)slab * a.C + c) * 2 + 1] = q;
        }
        __syncthreads();
    }
}

__global__ void bn_bwd_finalize_kernel(const BnBwdArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < a.nslab; ++k) { s += a.part[((long long)k * a.C + c) * 2]; q += a.part[((long long)k * a.C + c) * 2 + 1]; }
    a.dbeta[c] = (float)s;
    a.dgamma[c] = (float)q;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdArgs a) {
    const long long total = a.rows * a.C;
    const float inv_n = (a.flags & LT_BN_FROZEN) ? 0.f : 1.0f / (float)a.rows;       // frozen statistics: no batch-statistics terms
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % a.C);
        float xh;
        const float r = a.res ? ld1_f32_or_bf16(a.res, (size_t)i, a.a16) : 0.f;
        const float dzv = ld1_f32_or_bf16(a.dz, (size_t)i, a.a16);
        const float g = bn_g(a, dzv, ld1_f32_or_bf16(a.y, (size_t)i, a.y16), r, c, xh);
        const float invstd = 1.0f / sqrtf(a.var[c] + a.eps);
        st1_f32_or_bf16(a.dy, (size_t)i, a.a16, a.gamma[c] * invstd * (g - a.dbeta[c] * inv_n - xh * a.dgamma[c] * inv_n));
        if (a.dres) {
            const float dr = (a.flags & LT_EPI_RELU_POST) ? g : dzv;     // RELU_PRE / none: the residual is added after the activation
            st1_f32_or_bf16(a.dres, (size_t)i, a.a16, a.accumulate_res ? ld1_f32_or_bf16(a.dres, (size_t)i, a.a16) + dr : dr);
        }
    }
}

// the same three stages on float4 lanes (colsum.h); every BatchNorm layer of these networks takes this path
// ALL16: dz, y and the residual are bf16 tensors (the 16-bit-activation step: 8 bytes per tensor and row in flight); else the element types are the
// runtime flags' (fp32 step, and round 3's mixed step with its optional bf16 y): four rows in flight of 16-byte slots
template <bool ALL16, int NROWS>
struct BnBwdLoad {
    struct Raw16 { uint2 dz, y, r; };
    struct Raw32 { uint4 dz, y, r; };
    typedef typename std::conditional<ALL16, Raw16, Raw32>::type Raw;
    static constexpr int ROWS = NROWS;
    BnBwdArgs a;
    float invstd[4], mean[4], gam[4], bet[4];          // of the thread's four channels (prepare): bn_g's expressions with the constants hoisted
    __device__ __forceinline__ void prepare(int c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            invstd[e] = 1.0f / sqrtf(a.var[c + e] + a.eps);
            mean[e] = a.mean[c + e]; gam[e] = a.gamma[c + e]; bet[e] = a.beta[c + e];
        }
    }
    __device__ __forceinline__ Raw fetch(long long row, int c) const {
        const size_t off = (size_t)row * a.C + c;
        Raw w;
        if constexpr (ALL16) {
            w.dz = *(const uint2*)((const bf16_t*)a.dz + off);
            w.y = *(const uint2*)((const bf16_t*)a.y + off);
            w.r = a.res ? *(const uint2*)((const bf16_t*)a.res + off) : make_uint2(0u, 0u);
        } else {
            w.dz = w.y = w.r = make_uint4(0u, 0u, 0u, 0u);
            if (a.a16) { const uint2 t = *(const uint2*)((const bf16_t*)a.dz + off); w.dz.x = t.x; w.dz.y = t.y; }
            else w.dz = *(const uint4*)((const float*)a.dz + off);
            if (a.y16) { const uint2 t = *(const uint2*)((const bf16_t*)a.y + off); w.y.x = t.x; w.y.y = t.y; }
            else w.y = *(const uint4*)((const float*)a.y + off);
            if (a.res) {
                if (a.a16) { const uint2 t = *(const uint2*)((const bf16_t*)a.res + off); w.r.x = t.x; w.r.y = t.y; }
                else w.r = *(const uint4*)((const float*)a.res + off);
            }
        }
        return w;
    }
    __device__ __forceinline__ void eval(const Raw& w, int, float (&q)[2][4]) const {
        float4 dz4, y4, r4;
        if constexpr (ALL16) { dz4 = bf16x4_to_f32(w.dz); y4 = bf16x4_to_f32(w.y); r4 = bf16x4_to_f32(w.r); }
        else {
            dz4 = a.a16 ? bf16x4_to_f32(make_uint2(w.dz.x, w.dz.y)) : make_float4(__uint_as_float(w.dz.x), __uint_as_float(w.dz.y), __uint_as_float(w.dz.z), __uint_as_float(w.dz.w));
            y4 = a.y16 ? bf16x4_to_f32(make_uint2(w.y.x, w.y.y)) : make_float4(__uint_as_float(w.y.x), __uint_as_float(w.y.y), __uint_as_float(w.y.z), __uint_as_float(w.y.w));
            r4 = a.a16 ? bf16x4_to_f32(make_uint2(w.r.x, w.r.y)) : make_float4(__uint_as_float(w.r.x), __uint_as_float(w.r.y), __uint_as_float(w.r.z), __uint_as_float(w.r.w));
        }
        const float dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
        const bool post = a.flags & LT_EPI_RELU_POST, pre = a.flags & LT_EPI_RELU_PRE;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yv[e] - mean[e]) * invstd[e];          // the same expressions as bn_g / the apply kernel: the masks must agree
            const float v = xh * gam[e] + bet[e];
            float mk = 1.f;
            if (post) mk = (v + rv[e]) > 0.f ? 1.f : 0.f;
            else if (pre) mk = v > 0.f ? 1.f : 0.f;
            const float g = dzv[e] * mk;
            q[0][e] = g; q[1][e] = g * xh;
        }
    }
};

template <bool ALL16, int NROWS>
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const BnBwdArgs a, int cw4, int rl) {
    BnBwdLoad<ALL16, NROWS> ld;
    ld.a = a;
    colsum_partial<2>(a.rows, a.C, a.nslab, cw4, rl, a.part, ld);
}

struct BnBwdFin {
    float* dgamma; float* dbeta;
    __device__ __forceinline__ void operator()(int c, const double (&t)[2]) const { dbeta[c] = (float)t[0]; dgamma[c] = (float)t[1]; }
};

__global__ __launch_bounds__(256) void bn_bwd_finalize_vec_kernel(const BnBwdArgs a) { colsum_finalize<2>(a.part, a.C, a.nslab, BnBwdFin{a.dgamma, a.dbeta}); }

// dy (and the residual's gradient): a thread keeps ONE float4 of channels -- its per-channel constants live in registers -- and walks rows
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const BnBwdArgs a, int nslab, int cw4, int rl_n) {
    const long long r0 = a.rows * blockIdx.x / nslab, r1 = a.rows * (blockIdx.x + 1) / nslab;
    const int rl = threadIdx.x / cw4, cv = threadIdx.x - rl * cw4;
    const int c = (blockIdx.y * 256 + cv) * 4;
    if (c >= a.C) return;
    const float inv_n = (a.flags & LT_BN_FROZEN) ? 0.f : 1.0f / (float)a.rows;       // frozen statistics: no batch-statistics terms
    float invstd[4], mean[4], gam[4], bet[4], k1[4], kb[4], kg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        invstd[e] = 1.0f / sqrtf(a.var[c + e] + a.eps);
        mean[e] = a.mean[c + e]; gam[e] = a.gamma[c + e]; bet[e] = a.beta[c + e];
        k1[e] = gam[e] * invstd[e]; kb[e] = a.dbeta[c + e] * inv_n; kg[e] = a.dgamma[c + e] * inv_n;
    }
    const bool post = a.flags & LT_EPI_RELU_POST, pre = a.flags & LT_EPI_RELU_PRE;
    for (long long r = r0 + rl; r < r1; r += rl_n) {
        const size_t off = (size_t)r * a.C + c;
        const float4 dz4 = ld4_f32_or_bf16(a.dz, off, a.a16), y4 = ld4_f32_or_bf16(a.y, off, a.y16);
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.res) r4 = ld4_f32_or_bf16(a.res, off, a.a16);
        const float dzv[4] = {dz4.x, dz4.y, dz4.z, dz4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
        float o[4], dr[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yv[e] - mean[e]) * invstd[e];          // the same expressions as bn_g: the mask must match the reduction's
            const float v = xh * gam[e] + bet[e];
            float mk = 1.f;
            if (post) mk = (v + rv[e]) > 0.f ? 1.f : 0.f;
            else if (pre) mk = v > 0.f ? 1.f : 0.f;
            const float g = dzv[e] * mk;
            o[e] = k1[e] * (g - kb[e] - xh * kg[e]);
            dr[e] = post ? g : dzv[e];     // RELU_PRE / none: the residual is added after the activation
        }
        st4_f32_or_bf16(a.dy, off, a.a16, o[0], o[1], o[2], o[3]);
        if (a.dy16) *(uint2*)(a.dy16 + off) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        if (a.dres) {
            if (a.accumulate_res) {
                const float4 d0 = ld4_f32_or_bf16(a.dres, off, a.a16);
                dr[0] += d0.x; dr[1] += d0.y; dr[2] += d0.z; dr[3] += d0.w;
            }
            st4_f32_or_bf16(a.dres, off, a.a16, dr[0], dr[1], dr[2], dr[3]);
        }
    }
}

struct ChanSumLoad {
    typedef float4 Raw;
    static constexpr int ROWS = 8;
    const void* x; int C, x16;
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ Raw fetch(long long row, int c) const { return ld4_f32_or_bf16(x, (size_t)row * C + c, x16); }
    __device__ __forceinline__ void eval(const Raw& v, int, float (&q)[1][4]) const { q[0][0] = v.x; q[0][1] = v.y; q[0][2] = v.z; q[0][3] = v.w; }
};
__global__ __launch_bounds__(256) void channel_sum_vec_kernel(const void* __restrict__ x, int x16, long long rows, int C, int nslab, int cw4, int rl, double* __restrict__ part) {
    colsum_partial<1>(rows, C, nslab, cw4, rl, part, ChanSumLoad{x, C, x16});
}
struct ChanSumFin {
    float* out; int accumulate;
    __device__ __forceinline__ void operator()(int c, const double (&t)[1]) const { out[c] = accumulate ? out[c] + (float)t[0] : (float)t[0]; }
};
__global__ __launch_bounds__(256) void channel_sum_finalize_vec_kernel(const double* __restrict__ part, int C, int nslab, ChanSumFin fin) {
    colsum_finalize<1>(part, C, nslab, fin);
}

// layers without BatchNorm: z = act(y, res) with y = conv + bias: dy = dz * mask, dres likewise
__global__ __launch_bounds__(256) void act_bwd_kernel(const void* __restrict__ dz, const void* __restrict__ z, const void* __restrict__ res,
                                                      void* __restrict__ dy, void* __restrict__ dres, int flags, int accumulate_res, long long total) {
    const int a16 = (flags & LT_ACT_BF16) ? 1 : 0, z16 = a16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float m = 1.f;
        const float zi = (flags & (LT_EPI_SIGMOID | LT_EPI_RELU_POST | LT_EPI_RELU_PRE)) ? ld1_f32_or_bf16(z, (size_t)i, z16) : 0.f;
        if (flags & LT_EPI_SIGMOID) m = zi * (1.f - zi);                                   // z = sigmoid(v): the confidence heads' last layer
        else if (flags & LT_EPI_RELU_POST) m = zi > 0.f ? 1.f : 0.f;                       // z = relu(v + res)
        else if (flags & LT_EPI_RELU_PRE) m = zi > 0.f ? 1.f : 0.f;                        // z = relu(v); with a residual behind the ReLU the sign of v is
                                                                                               // not recoverable from z (lt_act_bwd refuses that combination)
        const float dzi = ld1_f32_or_bf16(dz, (size_t)i, a16);
        const float g = dzi * m;
        st1_f32_or_bf16(dy, (size_t)i, a16, g);
        if (dres) {
            const float dr = (flags & LT_EPI_RELU_POST) ? g : dzi;
            st1_f32_or_bf16(dres, (size_t)i, a16, accumulate_res ? ld1_f32_or_bf16(dres, (size_t)i, a16) + dr : dr);
        }
    }
}

__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const void* __restrict__ x, int x16, long long rows, int C, int nslab, double* __restrict__ part) {
    __shared__ double ss[256];
    const int slab = blockIdx.x;
    const long long r0 = rows * slab / nslab, r1 = rows * (slab + 1) / nslab;
    const int Cw = C < 256 ? C : 256, RL = 256 / Cw;
    const int rl = threadIdx.x / Cw, cl = threadIdx.x - rl * Cw;
    for (int c0 = 0; c0 < C; c0 += Cw) {
        const int c = c0 + cl;
        double s = 0.0;
        if (rl < RL && c < C)
            for (long long r = r0 + rl; r < r1; r += RL) s += (double)ld1_f32_or_bf16(x, (size_t)r * C + c, x16);
        ss[threadIdx.x] = s;
        __syncthreads();
        if (rl == 0 && c < C) {
            for (int k = 1; k < RL; ++k) s += ss[k * Cw + cl];
            part[(long long)slab * C + c] = s;
        }
        __syncthreads();
    }
}
__global__ void channel_sum_finalize_kernel(const double* __restrict__ part, int C, int nslab, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < nslab; ++k) s += part[(long long)k * C + c];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// max pool backward: one thread per (OUTPUT pixel, channel): finds the first maximal input of the window (scan order d, h, w like
// ATen's CPU max_pool) and adds dy there.  Windows overlap for k > s, so the outputs are walked in ceil(k / s)^3 CLASSES (output index
// modulo ceil(k / s) per dimension), one launch each: two windows of one class never share an input, every launch adds with plain
// read-add-write, and the order in which an input collects its (up to 4 for the stem's 3x3 / stride 2 pool) contributions is the class
// order -- bitwise repeatable (round 2: float atomics in arbitrary order)
struct PoolClass { int cd, ch, cw, md, mh, mw, nd, nh, nw; };      // class offset, class stride, outputs of the class per dimension

__global__ void maxpool_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy, void* __restrict__ dx, int a16, int N, int D, int H, int W, int C,
                                   int Do, int Ho, int Wo, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, PoolClass pc) {
    const long long total = (long long)N * pc.nd * pc.nh * pc.nw * C;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(g % C);
        long long r = g / C;
        const int ow = (int)(r % pc.nw) * pc.mw + pc.cw; r /= pc.nw;
        const int oh = (int)(r % pc.nh) * pc.mh + pc.ch; r /= pc.nh;
        const int od = (int)(r % pc.nd) * pc.md + pc.cd;
        const int n = (int)(r / pc.nd);
        float best = -INFINITY;
        long long bi = -1;
        for (int a = 0; a < kd; ++a) {
            const int id = od * sd - pd + a;
            if ((unsigned)id >= (unsigned)D) continue;
            for (int b = 0; b < kh; ++b) {
                const int ih = oh * sh - ph + b;
                if ((unsigned)ih >= (unsigned)H) continue;
                for (int cc = 0; cc < kw; ++cc) {
                    const int iw = ow * sw - pw + cc;
                    if ((unsigned)iw >= (unsigned)W) continue;
                    const long long idx = ((((long long)n * D + id) * H + ih) * W + iw) * C + c;
                    const float v = ld1_f32_or_bf16(x, (size_t)idx, a16);
                    if (v > best || bi < 0) { best = v; bi = idx; }
                }
            }
        }
        if (bi >= 0)
            st1_f32_or_bf16(dx, (size_t)bi, a16, ld1_f32_or_bf16(dx, (size_t)bi, a16) + ld1_f32_or_bf16(dy, (size_t)(((((long long)n * Do + od) * Ho + oh) * Wo + ow) * C + c), a16));
    }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------------------
// dW[co][k], k = tap * Cin + ci (the [cout][k_pad] layout of lt_conv_fwd's weights, fp32).  A workgroup owns a 32 (co) x 128 (k) block
// of dW and walks ALL GEMM rows m (output pixels of the forward conv) with its four waves taking every fourth pair of rows; per
// pair one A load (dY[m][co0 + lane & 31], two rows per wave-instruction) and four B loads (X[m @ tap][ci], lanes along k) feed four
// 32x32x2 MFMAs; the four waves' accumulators are summed through LDS at the end.  No atomics: the result does not depend on the run.
struct WgradArgs {
    const float* dy;         // [M][ldy]: gradient of the convolution output, GEMM row m = (n, od, oh, ow)
    const float* x;          // channels-last input [N][D][H][W][Cin]
    const int4* taps;        // [ntaps] = (dd, dh, dw, unused)
    float* out;              // S == 1: dw [cout_pad][k_pad];  S > 1: workspace [S][cout_pad][k_pad] of per-slab partial sums
    int N, D, H, W, Cin, log2Cin, Do, Ho, Wo, sd, sh, sw, pd, ph, pw;
    int Cout, ldy, k_pad, ntaps, M, accumulate, cout_pad;
    int n_k_t, n_tiles, rows_per_slab;
};

// dW = dY^T x im2col(X): a GEMM whose reduction runs over the M = N*Do*Ho*Wo output pixels (1e4 .. 1e6) while the result is small
// (Cout x taps*Cin).  One WAVE owns a (32 CT) x (32 KT) block of dW in 128 accumulator registers (exact-fp32 32x32x2 MFMA: one
// value per lane and operand, so the operands go from global memory straight into the MFMA -- each lane's loads are 128-byte
// coalesced rows of dY / of one tap's channels -- with no LDS stage and no barrier); the four waves of a workgroup take neighbouring
// blocks (same dY columns: their dY loads hit L1), and the pixel range is cut into S slabs across blockIdx.y so that the launch
// fills the chip whatever the layer's shape; slab partial sums are reduced by wgrad_reduce_kernel in a fixed order (deterministic).
// Operands of the next pixel pair are loaded before the MFMAs of the current one (software pipeline, two register sets).
// bits t = 0 .. 7 with 0 <= i0 + t < size
__device__ __forceinline__ int range_mask(int i0, int size) {
    const int lo = max(0, -i0), hi = min(7, size - 1 - i0);          // hi < lo: empty
    return hi >= lo ? ((2 << hi) - 1) & ~((1 << lo) - 1) : 0;
}

template <int CT, int KT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroups are dealt round-robin to the 8 XCDs: give every XCD a CONTIGUOUS range of (slab, tile group) pairs, slab-major, so that
    // the tiles of one slab (same dY / X rows) and neighbouring slabs (shared halo rows / planes) meet in the same L2
    const int total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int lin2 = (total & 7) ? lin : (lin & 7) * (total >> 3) + (lin >> 3);
    const int slab = lin2 / (int)gridDim.x, tgroup = lin2 - slab * (int)gridDim.x;
    const int t = tgroup * 4 + wave;
    if (t >= a.n_tiles) return;
    const int co0 = (t / a.n_k_t) * (32 * CT), k0 = (t % a.n_k_t) * (32 * KT);
    const int col = lane & 31, half = lane >> 5;          // A: co = co0 + 32 c + col, row m + half;  B: k = k0 + 32 j + col, row m + half
    // per GEMM column: element offset of its tap and channel, and a selector with one bit per dimension (bit dd, bit 8 + dh, bit 16 + dw;
    // bit 31 alone for a column past K): a pixel's taps are valid where its per-dimension range masks contain the selector
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
    // this lane's row walks m_begin + half, + 2, + 4, ...: (n, od, oh, ow) advance incrementally
    int m = m_begin + half;
    const int hw = a.Ho * a.Wo, dhw = a.Do * hw;
    int n = m / dhw, r = m - n * dhw;
    int od = r / hw; r -= od * hw;
    int oh = r / a.Wo, ow = r - oh * a.Wo;
    const float* dyp = a.dy + co0 + col;

    // BRANCH-FREE loads with NOTHING behind them: an absent element (padding, a row past the slab, a column past K) is read from
    // offset 0 and its validity bit is cleared; the bits are applied in mma(), i.e. in program order BEHIND the loads of the next
    // pipeline stages.  (Loads inside divergent branches cannot be counted by the compiler, and a select directly behind a load makes it
    // wait there: either way every MFMA group would sit behind vmcnt(0) and the software pipeline below would be serialised.)
    auto load = [&](float (&av)[CT], float (&bv)[KT], unsigned& okbits) {
        const bool m_ok = m < m_end;
        unsigned bits = 0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const bool ok = m_ok & co_ok[c];
            av[c] = dyp[ok ? (long long)m * a.ldy + 32 * c : 0ll];
            bits |= ok ? 1u << (KT + c) : 0u;
        }
        const int id0 = od * a.sd - a.pd, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
        const int pix = (((n * a.D + id0) * a.H + ih0) * a.W + iw0) * a.Cin;          // may point in front of the tensor (padding): only in-bounds taps are read
        // per dimension the tap offsets t with 0 <= i0 + t < size, as a bit range [max(0, -i0), min(7, size - 1 - i0)] (taps are 0 .. 7)
        const int rmask = m_ok ? (range_mask(id0, a.D) | (range_mask(ih0, a.H) << 8) | (range_mask(iw0, a.W) << 16)) : 0;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const bool ok = (rmask & tsel[j]) == tsel[j];
            bv[j] = *(const float*)((const char*)a.x + ((size_t)(ok ? (unsigned)(pix + toff[j]) : 0u) << 2));
            bits |= ok ? 1u << j : 0u;
        }
        okbits = bits;
        // next pixel pair, branch-free: a coordinate moves by at most 2, so two conditional wraps per level always suffice
        m += 2; ow += 2;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) { const int w = ow >= a.Wo; ow -= w ? a.Wo : 0; oh += w; }
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) { const int w = oh >= a.Ho; oh -= w ? a.Ho : 0; od += w; }
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) { const int w = od >= a.Do; od -= w ? a.Do : 0; n += w; }
    };
    auto mma = [&](const float (&av)[CT], const float (&bv)[KT], unsigned bits) {
        float af[CT], bf[KT];
#pragma unroll
        for (int c = 0; c < CT; ++c) af[c] = (bits >> (KT + c)) & 1u ? av[c] : 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j) bf[j] = (bits >> j) & 1u ? bv[j] : 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int j = 0; j < KT; ++j) acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c], bf[j], acc[c][j], 0, 0, 0);
    };
    // three pixel pairs in flight ahead of the one in the MFMAs (rows past m_end load nothing: the extra MFMAs add zeros)
    float av0[CT], bv0[KT], av1[CT], bv1[KT], av2[CT], bv2[KT], av3[CT], bv3[KT];
    unsigned ok0, ok1, ok2, ok3;
    const int nit = (m_end - m_begin + 1) >> 1;
    load(av0, bv0, ok0); load(av1, bv1, ok1); load(av2, bv2, ok2);
    for (int it = 0; it < nit; it += 4) {
        load(av3, bv3, ok3); mma(av0, bv0, ok0);
        load(av0, bv0, ok0); mma(av1, bv1, ok1);
        load(av1, bv1, ok1); mma(av2, bv2, ok2);
        load(av2, bv2, ok2); mma(av3, bv3, ok3);
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

// ---- weight gradient of the 3x3x3 / stride 1 / pad 1 convolutions (V2V) from LDS bricks ---------------------------------------------
// The generic kernel above streams its operands from L2, and the im2col view of a 3^3 convolution re-reads every voxel 27 times with
// reuse distances (a row, a plane) that no cache level holds: 2.3 ms for 58 GFLOP.  Here a workgroup owns a (32 co) x (27 taps x 32 ci)
// block of dW and walks BRICKS of 2 x 4 x 16 voxels: the brick's dY rows (16 KB) and its X halo (4 x 6 x 18 voxels x 32 channels,
// 54 KB) are staged in LDS once, then every wave takes every fourth tap (7 accumulator blocks of 16 registers) and feeds the exact-fp32
// 32x32x2 MFMA with ds_read_b32 operands -- one A read and seven B reads per seven MFMAs.  Two workgroups per CU: one loads while
// the other multiplies.  Slabs of bricks across blockIdx.z, partial sums reduced by wgrad_reduce_kernel (deterministic).
constexpr int BRK_D = 2, BRK_H = 4, BRK_W = 16, BRK_VOX = BRK_D * BRK_H * BRK_W;
constexpr int BRK_HD = BRK_D + 2, BRK_HH = BRK_H + 2, BRK_HW = BRK_W + 2, BRK_HVOX = BRK_HD * BRK_HH * BRK_HW;

struct BrickArgs {
    const float* dy;         // [N*D*H*W][ldy]
    const float* x;          // [N][D][H][W][Cin]
    const int4* taps;        // 27 x (dd, dh, dw, -)
    float* out;              // [S][cout_pad][k_pad]
    int N, D, H, W, Cin, ldy, cout_pad, k_pad;
    int nbd, nbh, nbw;       // bricks per dimension
    int nbricks, bricks_per_slab;
};

__global__ __launch_bounds__(256, 2) void conv3d_wgrad_brick_kernel(const BrickArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[BRK_HVOX * 32];
    __shared__ __attribute__((aligned(16))) float ds[BRK_VOX * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
    // this wave's taps: wave, wave + 4, ...; LDS offset (in voxels of the halo brick) of each
    int toff[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int t = wave + 4 * i;
        const int4 tp = a.taps[t < 27 ? t : 0];
        toff[i] = ((tp.x * BRK_HH + tp.y) * BRK_HW + tp.z) * 32;
    }
    f32x16 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int b_begin = blockIdx.z * a.bricks_per_slab, b_end = min(a.nbricks, b_begin + a.bricks_per_slab);
    for (int b = b_begin; b < b_end; ++b) {
        int r = b;
        const int bw = r % a.nbw; r /= a.nbw;
        const int bh = r % a.nbh; r /= a.nbh;
        const int bd = r % a.nbd;
        const int n = r / a.nbd;
        const int d0 = bd * BRK_D, h0 = bh * BRK_H, w0 = bw * BRK_W;
        __syncthreads();          // the previous brick's reads are done
        // X halo brick: 432 voxels x 8 float4
        for (int i = threadIdx.x; i < BRK_HVOX * 8; i += 256) {
            const int v = i >> 3, q = i & 7;
            const int hw_ = v % BRK_HW, t2 = v / BRK_HW;
            const int hh_ = t2 % BRK_HH, hd_ = t2 / BRK_HH;
            const int d = d0 + hd_ - 1, h = h0 + hh_ - 1, w = w0 + hw_ - 1;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                val = *(const float4*)(a.x + ((((size_t)n * a.D + d) * a.H + h) * a.W + w) * a.Cin + ci0 + q * 4);
            *(float4*)(xs + v * 32 + q * 4) = val;
        }
        // dY brick: 128 voxels x 8 float4
        for (int i = threadIdx.x; i < BRK_VOX * 8; i += 256) {
            const int v = i >> 3, q = i & 7;
            const int w = v % BRK_W, t2 = v / BRK_W;
            const int h = t2 % BRK_H, d = t2 / BRK_H;
            const size_t row = (((size_t)n * a.D + d0 + d) * a.H + h0 + h) * a.W + w0 + w;
            *(float4*)(ds + v * 32 + q * 4) = *(const float4*)(a.dy + row * a.ldy + co0 + q * 4);
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < BRK_VOX / 2; ++p) {
            const int v = 2 * p + half;                       // brick-linear voxel (w fastest): a pair never straddles a row
            const int w = v % BRK_W, t2 = v / BRK_W;
            const int h = t2 % BRK_H, d = t2 / BRK_H;
            const float av = ds[v * 32 + col];
            const int hb = ((d * BRK_HH + h) * BRK_HW + w) * 32 + col;          // halo voxel of tap (0, 0, 0)
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const float bv = xs[hb + toff[i]];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
            }
        }
    }
    float* out = a.out + (size_t)blockIdx.z * a.cout_pad * a.k_pad;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int t = wave + 4 * i;
        if (t >= 27) break;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = co0 + 8 * (e >> 2) + 4 * half + (e & 3);
            out[(size_t)row * a.k_pad + t * a.Cin + ci0 + col] = acc[i][e];
        }
    }
}

// ---- weight gradient of V2V's 7x7x7 front convolution (32 -> 16 channels, stride 1, pad 3) ----------------------------------------------
// 343 taps x (16 x 32) = 686 KB of accumulators: one workgroup owns ONE kd plane of the filter (49 taps x 16 co x 32 ci = 98 blocks of
// the 16x16x4 fp32 MFMA -- no padding of the 16 output channels to a 32-row tile -- 24-25 blocks of 4 registers per wave).  Bricks of
// 2 x 4 x 16 voxels: dY brick (8 KB) and the two X planes the kd needs with a 3-voxel (h, w) halo (2 x 10 x 22 voxels x 32 ch = 55 KB) in
// LDS.  Taps must be in (kd, kh, kw) order (the kernel traps otherwise).  13.5 ms -> see DESIGN.md.
constexpr int K7_HD = 2, K7_HH = BRK_H + 6, K7_HW = BRK_W + 6, K7_HVOX = K7_HD * K7_HH * K7_HW;

__global__ __launch_bounds__(256, 2) void conv3d_wgrad_k7_kernel(const BrickArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[K7_HVOX * 32];
    __shared__ __attribute__((aligned(16))) float ds[BRK_VOX * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;            // MFMA 16x16x4: A row / B column = lane % 16, k (voxel of the quad) = lane / 16
    const int kd = blockIdx.x;
    // this wave's blocks: b = wave, wave + 4, ... < 98; block b = (tap of the plane, ci half)
    int boff[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        const int b = wave + 4 * i;
        const int t = b < 98 ? b >> 1 : 0;
        const int4 tp = a.taps[kd * 49 + t];
        if (tp.x != kd) __builtin_trap();
        boff[i] = (tp.y * K7_HW + tp.z) * 32 + (b & 1) * 16;
    }
    f32x4 acc[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int b_begin = blockIdx.z * a.bricks_per_slab, b_end = min(a.nbricks, b_begin + a.bricks_per_slab);
    for (int br = b_begin; br < b_end; ++br) {
        int r = br;
        const int bw = r % a.nbw; r /= a.nbw;
        const int bh = r % a.nbh; r /= a.nbh;
        const int bd = r % a.nbd;
        const int n = r / a.nbd;
        const int d0 = bd * BRK_D, h0 = bh * BRK_H, w0 = bw * BRK_W;
        __syncthreads();
        for (int i = threadIdx.x; i < K7_HVOX * 8; i += 256) {
            const int v = i >> 3, q = i & 7;
            const int hw_ = v % K7_HW, t2 = v / K7_HW;
            const int hh_ = t2 % K7_HH, hd_ = t2 / K7_HH;
            const int d = d0 + hd_ + kd - 3, h = h0 + hh_ - 3, w = w0 + hw_ - 3;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)d < (unsigned)a.D && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                val = *(const float4*)(a.x + ((((size_t)n * a.D + d) * a.H + h) * a.W + w) * 32 + q * 4);
            *(float4*)(xs + v * 32 + q * 4) = val;
        }
        for (int i = threadIdx.x; i < BRK_VOX * 4; i += 256) {          // 128 voxels x 4 float4 (16 co)
            const int v = i >> 2, q = i & 3;
            const int w = v % BRK_W, t2 = v / BRK_W;
            const int h = t2 % BRK_H, d = t2 / BRK_H;
            const size_t row = (((size_t)n * a.D + d0 + d) * a.H + h0 + h) * a.W + w0 + w;
            *(float4*)(ds + v * 16 + q * 4) = *(const float4*)(a.dy + row * a.ldy + q * 4);
        }
        __syncthreads();
#pragma unroll 2
        for (int s4 = 0; s4 < BRK_VOX / 4; ++s4) {
            const int v = 4 * s4 + kq;                        // four voxels along w per MFMA (16 wide: a quad never straddles a row)
            const int w = v % BRK_W, t2 = v / BRK_W;
            const int h = t2 % BRK_H, d = t2 / BRK_H;
            const float av = ds[v * 16 + col];
            const float* xb = xs + ((d * K7_HH + h) * K7_HW + w) * 32 + col;
#pragma unroll
            for (int i = 0; i < 25; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xb[boff[i]], acc[i], 0, 0, 0);
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

// ---- weight gradient of the 1x1 / stride 1 convolutions: dW[co][ci] = sum_m dY[m][co] X[m][ci], an LDS-tiled GEMM ----------------------
// Workgroup tile 128 (co) x 128 (ci), 2 x 2 waves of 64 x 64 (four 32x32 accumulator blocks each); the reduction index m (pixels) is
// walked in chunks of 32 rows staged in LDS (row stride 160 floats: the two half-waves of a ds_read_b32 land in different bank halves);
// slabs of chunks across blockIdx.z.  Replaces the L2-streamed generic kernel for the bottleneck 1x1 layers (120 -> ~45 us on 1024<->256).
constexpr int PW_R = 32, PW_LD = 160;

struct PwArgs {
    const float* dy; const float* x; float* out;
    int M, Cout, Cin, ldy, cout_pad, k_pad, chunks, chunks_per_slab;
};

__global__ __launch_bounds__(256, 2) void wgrad_pw_kernel(const PwArgs a) {
    __shared__ __attribute__((aligned(16))) float ys[PW_R * PW_LD];
    __shared__ __attribute__((aligned(16))) float xs[PW_R * PW_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int co0 = blockIdx.x * 128, ci0 = blockIdx.y * 128;
    const int wc = (wave >> 1) * 64, wk = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][j][e] = 0.f;
    const int c_begin = blockIdx.z * a.chunks_per_slab, c_end = min(a.chunks, c_begin + a.chunks_per_slab);
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int m0 = ch * PW_R;
        __syncthreads();
        for (int i = threadIdx.x; i < PW_R * 32; i += 256) {          // 32 rows x 32 float4 of each operand
            const int r = i >> 5, q = (i & 31) * 4;
            const int m = m0 + r;
            float4 yv = make_float4(0.f, 0.f, 0.f, 0.f), xv = yv;
            if (m < a.M) {
                if (co0 + q < a.Cout) yv = *(const float4*)(a.dy + (size_t)m * a.ldy + co0 + q);      // Cout % 4 == 0 (checked by the launcher)
                if (ci0 + q < a.Cin) xv = *(const float4*)(a.x + (size_t)m * a.Cin + ci0 + q);
            }
            *(float4*)(ys + r * PW_LD + q) = yv;
            *(float4*)(xs + r * PW_LD + q) = xv;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < PW_R / 2; ++p) {
            const int r = 2 * p + half;
            const float a0 = ys[r * PW_LD + wc + col], a1 = ys[r * PW_LD + wc + 32 + col];
            const float b0 = xs[r * PW_LD + wk + col], b1 = xs[r * PW_LD + wk + 32 + col];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float* out = a.out + (size_t)blockIdx.z * a.cout_pad * a.k_pad;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = co0 + wc + 32 * c + 8 * (e >> 2) + 4 * half + (e & 3);
                const int k = ci0 + wk + 32 * j + col;
                if (row < a.cout_pad && k < a.k_pad) out[(size_t)row * a.k_pad + k] = acc[c][j][e];
            }
}

// ---- weight gradient of the 2D 3x3 / stride 1 / pad 1 convolutions (ResNet) from LDS bricks of 8 x 8 pixels ------------------------------
// Workgroup: 32 (co) x (9 taps x 128 ci); the four waves take one 32-channel ci block each and all nine taps (144 accumulator registers);
// LDS: the dY brick (64 pixels x 32 co) and the X halo brick (10 x 10 pixels x 128 ci, pixel stride 128 floats + the same half-wave bank
// argument: a pixel pair is 128 floats apart = bank 0 again on 64 banks, so odd pixels are stored 32 floats further: stride 160).
constexpr int B2_H = 8, B2_W = 8, B2_PIX = 64, B2_HH = 10, B2_HW = 10, B2_HPIX = 100, B2_LD = 160;

struct Brick2Args {
    const float* dy; const float* x; float* out;
    int N, H, W, Cin, Cout, ldy, cout_pad, k_pad, nbh, nbw, nbricks, bricks_per_slab;
};

__global__ __launch_bounds__(256, 2) void conv2d_wgrad_brick_kernel(const Brick2Args a) {
    __shared__ __attribute__((aligned(16))) float xs[B2_HPIX * B2_LD];          // 64 KB
    __shared__ __attribute__((aligned(16))) float ds[B2_PIX * 32];              // 8 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 128;
    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int b_begin = blockIdx.z * a.bricks_per_slab, b_end = min(a.nbricks, b_begin + a.bricks_per_slab);
    for (int b = b_begin; b < b_end; ++b) {
        int r = b;
        const int bw = r % a.nbw; r /= a.nbw;
        const int bh = r % a.nbh;
        const int n = r / a.nbh;
        const int h0 = bh * B2_H, w0 = bw * B2_W;
        __syncthreads();
        for (int i = threadIdx.x; i < B2_HPIX * 32; i += 256) {          // 100 halo pixels x 32 float4 (128 channels)
            const int v = i >> 5, q = (i & 31) * 4;
            const int hh = v / B2_HW, hw = v - hh * B2_HW;
            const int h = h0 + hh - 1, w = w0 + hw - 1;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && ci0 + q < a.Cin)
                val = *(const float4*)(a.x + (((size_t)n * a.H + h) * a.W + w) * a.Cin + ci0 + q);
            *(float4*)(xs + v * B2_LD + q) = val;
        }
        for (int i = threadIdx.x; i < B2_PIX * 8; i += 256) {            // 64 pixels x 8 float4 (32 co)
            const int v = i >> 3, q = (i & 7) * 4;
            const int h = v >> 3, w = v & 7;
            const size_t row = ((size_t)n * a.H + h0 + h) * a.W + w0 + w;
            *(float4*)(ds + v * 32 + q) = *(const float4*)(a.dy + row * a.ldy + co0 + q);
        }
        __syncthreads();
#pragma unroll 2
        for (int p = 0; p < B2_PIX / 2; ++p) {
            const int v = 2 * p + half;                      // pixel pair along w (8 wide: never straddles a row)
            const int h = v >> 3, w = v & 7;
            const float av = ds[v * 32 + col];
            const float* xb = xs + (h * B2_HW + w) * B2_LD + wave * 32 + col;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float bv = xb[((t / 3) * B2_HW + (t % 3)) * B2_LD];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
    float* out = a.out + (size_t)blockIdx.z * a.cout_pad * a.k_pad;
    const int ci = ci0 + wave * 32 + col;
    if (ci < a.Cin)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = co0 + 8 * (e >> 2) + 4 * half + (e & 3);
                out[(size_t)row * a.k_pad + t * a.Cin + ci] = acc[t][e];
            }
}

__global__ void gather_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        const int j = idx[i];
        dst[i] = j >= 0 ? src[j] : 0.f;
    }
}

// fp32 -> bf16 (round to nearest even), eight elements per thread: the operands of the mixed-precision training convolutions
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
    const long long n8 = n >> 3;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += 256ll * gridDim.x) {
        const float4 a = ((const float4*)src)[2 * i], b = ((const float4*)src)[2 * i + 1];
        ((uint4*)dst)[i] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    }
    if (blockIdx.x == 0)
        for (long long i = (n8 << 3) + threadIdx.x; i < n; i += 256) dst[i] = f32_to_bf16(src[i]);
}

// y += x (an input gradient that cannot ride in a bf16 kernel's epilogue), float4 lanes
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ y, const float* __restrict__ x, long long n) {
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += 256ll * gridDim.x) {
        float4 a = ((float4*)y)[i];
        const float4 b = ((const float4*)x)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        ((float4*)y)[i] = a;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) y[i] += x[i];
}

// dst[r][c] = c < C ? src[r][c] : 0 -- dY of the 17-joint output layer widened to the 32 channels lt_conv_fwd wants on its input
__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int C, int Cpad) {
    const long long total = rows * Cpad;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
        const long long r = i / Cpad;
        const int c = (int)(i - r * Cpad);
        dst[i] = c < C ? src[r * C + c] : 0.f;
    }
}

struct GatherJob { const float* src; const int* idx; float* dst; long long n; int first_block; int out_bf16; };          // out_bf16: dst is a bf16 array

// many gathers in ONE launch (a layer's gather is a few-microsecond kernel: 750 of them per training step were launch-bound);
// a workgroup serves 1024 consecutive elements of the job its index falls into
__global__ __launch_bounds__(256) void gather_multi_kernel(const GatherJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const GatherJob j = jobs[lo];
    const long long base = (long long)(blockIdx.x - j.first_block) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i >= j.n) break;
        const int k = j.idx[i];
        const float v = k >= 0 ? j.src[k] : 0.f;
        if (j.out_bf16) ((bf16_t*)j.dst)[i] = f32_to_bf16(v);
        else j.dst[i] = v;
    }
}

// the bf16 weights of the mixed-precision step straight from the live fp32 Parameter (gather + round to nearest even in one pass)
__global__ void gather_bf16_kernel(const float* __restrict__ src, const int* __restrict__ idx, bf16_t* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        const int j = idx[i];
        dst[i] = f32_to_bf16(j >= 0 ? src[j] : 0.f);
    }
}

struct WgradPlan { int variant, n_co_t, n_k_t, S, rows_per_slab; };

// tile shape by layer shape, slab count so that ~1024 workgroups are in flight (two per CU at two waves per SIMD), partial sums capped at 48 MiB
WgradPlan wgrad_plan(long long M, int cout_pad, int k_pad) {
    WgradPlan p;
    p.variant = k_pad <= 64 ? 0 : cout_pad <= 32 ? 2 : 1;
    const int ct = p.variant == 0 ? 4 : p.variant == 1 ? 2 : 1, kt = 8 / ct;
    p.n_co_t = (int)cdiv(cout_pad, 32 * ct); p.n_k_t = (int)cdiv(k_pad, 32 * kt);
    const long long wgs = cdiv((long long)p.n_co_t * p.n_k_t, 4);
    long long S = cdiv(1024, wgs);
    const long long cap = (48ll << 20) / ((long long)cout_pad * k_pad * 4);
    if (S > cap) S = cap;
    if (S > M / 64) S = M / 64;
    if (S < 1) S = 1;
    if (S >= 8) S &= ~7ll;          // a multiple of 8: the kernel gives each XCD a contiguous range of slabs
    long long rps = cdiv(M, S);
    rps += rps & 1;
    p.rows_per_slab = (int)rps;
    p.S = (int)S;                   // trailing slabs may be empty (they write zeros)
    return p;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float grad = g[i];
        if (weight_decay != 0.f) grad += weight_decay * p[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * grad;
        const float vi = beta2 * v[i] + (1.f - beta2) * grad * grad;
        m[i] = mi; v[i] = vi;
        // torch.optim.Adam (single tensor): denom = sqrt(v) / sqrt(bias_correction2) + eps; p -= lr / bias_correction1 * m / denom
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

struct AdamJob { float* p; const float* g; float* m; float* v; long long n; float lr; int first_block; };

// every parameter tensor of the model in ONE launch: a workgroup updates 1024 consecutive elements of the job its index falls into
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamJob* __restrict__ jobs, int njobs, float beta1, float beta2, float eps, float weight_decay,
                                                         float bc1, float bc2) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {          // last job whose first_block <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamJob j = jobs[lo];
    const long long base = (long long)(blockIdx.x - j.first_block) * 1024;
    const float sq2 = sqrtf(bc2);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i >= j.n) break;
        float grad = j.g[i];
        if (weight_decay != 0.f) grad += weight_decay * j.p[i];
        const float mi = beta1 * j.m[i] + (1.f - beta1) * grad;
        const float vi = beta2 * j.v[i] + (1.f - beta2) * grad * grad;
        j.m[i] = mi; j.v[i] = vi;
        const float denom = sqrtf(vi) / sq2 + eps;
        j.p[i] -= (j.lr / bc1) * (mi / denom);
    }
}

int slabs_for(long long rows) { return (int)(rows < 1024 ? 1 : (rows / 256 < 1024 ? rows / 256 : 1024)); }

}  // namespace

extern "C" int lt_bn_act_fwd(const void* y, const float* mean, const float* var, const float* gamma, const float* beta, const void* residual,
                             void* z, void* z_bf16, int64_t rows, int32_t C, float eps, int32_t flags, void* stream) {
    LT_REQUIRE(y && mean && var && gamma && beta && z, LT_ERR_INVALID, "lt_bn_act_fwd: null argument");
    LT_REQUIRE(!(flags & LT_ACT_BF16) || !z_bf16, LT_ERR_INVALID, "lt_bn_act_fwd: with LT_ACT_BF16 z itself is the bf16 tensor (z_bf16 must be NULL)");
    LT_REQUIRE(rows >= 1 && C >= 4 && C % 4 == 0, LT_ERR_UNSUPPORTED, "lt_bn_act_fwd: C %% 4 == 0 required (C=%d)", C);
    BnActArgs a;
    a.y = y; a.mean = mean; a.var = var; a.gamma = gamma; a.beta = beta; a.res = residual; a.z = z; a.z16 = (bf16_t*)z_bf16; a.eps = eps; a.flags = flags; a.C = C; a.rows = rows;
    a.y16 = (flags & LT_BN_Y_BF16) ? 1 : 0;
    a.a16 = (flags & LT_ACT_BF16) ? 1 : 0;
    long long blocks = cdiv(rows * (C / 4), 256);
    if (blocks > 8192) blocks = 8192;
    {          // blocks * 256 a multiple of C / 4: every thread of the grid-stride loop stays on its four channels
        long long c4n = C / 4, gg = c4n, hh = 256;
        while (hh) { const long long t = gg % hh; gg = hh; hh = t; }          // gcd(C / 4, 256)
        const long long m = c4n / gg;
        blocks = cdiv(blocks, m) * m;
    }
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    LT_CHECK_LAUNCH("lt_bn_act_fwd");
    return LT_OK;
}

extern "C" size_t lt_bn_act_bwd_workspace(int64_t rows, int32_t C) {
    const size_t generic = (size_t)slabs_for(rows) * C * 2 * sizeof(double), fast = colsum_workspace(rows, C, 2);
    return generic > fast ? generic : fast;
}

extern "C" int lt_bn_act_bwd(const void* dz, const void* y, const void* residual, const float* mean, const float* var, const float* gamma,
                             const float* beta, void* dy, void* dy_bf16, float* dgamma, float* dbeta, void* dres, int32_t accumulate_res, int64_t rows,
                             int32_t C, float eps, int32_t flags, void* workspace, void* stream) {
    LT_REQUIRE(dz && y && mean && var && gamma && beta && dy && dgamma && dbeta && workspace, LT_ERR_INVALID, "lt_bn_act_bwd: null argument");
    LT_REQUIRE(rows >= 1 && C >= 1 && C <= 4096, LT_ERR_INVALID, "lt_bn_act_bwd: bad shape");
    LT_REQUIRE(!dres || residual, LT_ERR_INVALID, "lt_bn_act_bwd: a residual gradient needs the residual");
    BnBwdArgs a;
    a.dz = dz; a.y = y; a.res = residual; a.mean = mean; a.var = var; a.gamma = gamma; a.beta = beta; a.part = (double*)workspace;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dy = dy; a.dy16 = (bf16_t*)dy_bf16; a.dres = dres; a.eps = eps; a.flags = flags; a.C = C; a.nslab = slabs_for(rows);
    a.accumulate_res = accumulate_res; a.rows = rows;
    a.y16 = (flags & LT_BN_Y_BF16) ? 1 : 0;
    a.a16 = (flags & LT_ACT_BF16) ? 1 : 0;
    LT_REQUIRE(!a.y16 || C % 4 == 0, LT_ERR_UNSUPPORTED, "lt_bn_act_bwd: a bf16 y needs C %% 4 == 0 (C=%d)", C);
    LT_REQUIRE(!a.a16 || (C % 4 == 0 && !dy_bf16), LT_ERR_UNSUPPORTED, "lt_bn_act_bwd: LT_ACT_BF16 needs C %% 4 == 0 and no separate bf16 copy of dy (C=%d)", C);
    hipStream_t st = (hipStream_t)stream;
    if (colsum_fast(C)) {
        const ColsumPlan p = colsum_plan(rows, C);
        a.nslab = p.nslab;
        // rows in flight per thread: 4 / 8 / 2 measured 130.9 / 128.9 / 129.5 samples/s on the act16 step at 8 samples (one session); LT_BNBWD_ROWS=8 keeps the wider one
        static const int nrows = [] { const char* e = getenv("LT_BNBWD_ROWS"); return e ? atoi(e) : 4; }();
        if (a.a16 && a.y16) {
            if (nrows == 8) hipLaunchKernelGGL((bn_bwd_reduce_vec_kernel<true, 8>), dim3(p.nslab, p.ncb), dim3(256), 0, st, a, p.cw4, p.rl);
            else if (nrows == 2) hipLaunchKernelGGL((bn_bwd_reduce_vec_kernel<true, 2>), dim3(p.nslab, p.ncb), dim3(256), 0, st, a, p.cw4, p.rl);
            else hipLaunchKernelGGL((bn_bwd_reduce_vec_kernel<true, 4>), dim3(p.nslab, p.ncb), dim3(256), 0, st, a, p.cw4, p.rl);
        } else
            hipLaunchKernelGGL((bn_bwd_reduce_vec_kernel<false, 4>), dim3(p.nslab, p.ncb), dim3(256), 0, st, a, p.cw4, p.rl);
        LT_CHECK_LAUNCH("lt_bn_act_bwd(reduce)");
        hipLaunchKernelGGL(bn_bwd_finalize_vec_kernel, dim3((unsigned)cdiv(C, COLSUM_FIN_C)), dim3(256), 0, st, a);
        LT_CHECK_LAUNCH("lt_bn_act_bwd(finalize)");
        long long ns = rows / ((long long)p.rl * 2);          // >= 2 rows per thread, <= 4096 workgroups
        ns = ns < 1 ? 1 : ns > 4096 / p.ncb ? 4096 / p.ncb : ns;
        hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3((unsigned)ns, p.ncb), dim3(256), 0, st, a, (int)ns, p.cw4, p.rl);
        LT_CHECK_LAUNCH("lt_bn_act_bwd(apply)");
        return LT_OK;
    }
    LT_REQUIRE(!dy_bf16, LT_ERR_UNSUPPORTED, "lt_bn_act_bwd: the bf16 copy of dy needs a channel count of the vector path (C=%d)", C);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(a.nslab), dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_bn_act_bwd(reduce)");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_bn_act_bwd(finalize)");
    const long long blocks = cdiv(rows * C, 256);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_bn_act_bwd(apply)");
    return LT_OK;
}

extern "C" int lt_act_bwd(const void* dz, const void* z, const void* residual, void* dy, void* dres, int32_t accumulate_res, int64_t total,
                          int32_t flags, void* stream) {
    LT_REQUIRE(dz && z && dy && total >= 1, LT_ERR_INVALID, "lt_act_bwd: bad argument");
    // z = relu(v) + res: (z - res) > 0 loses a live gradient whenever 0 < relu(v) < ulp(res) / 2 -- the mask has to come from v, which a layer
    // without BatchNorm does not keep (the tape never records this shape: every such layer of the reference's networks carries BatchNorm)
    LT_REQUIRE(!((flags & LT_EPI_RELU_PRE) && !(flags & (LT_EPI_RELU_POST | LT_EPI_SIGMOID)) && residual), LT_ERR_UNSUPPORTED,
               "lt_act_bwd: ReLU in front of a residual add without BatchNorm (the mask of relu(v) cannot be rebuilt from relu(v) + res)");
    const long long blocks = cdiv(total, 256);
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, dz, z, residual, dy, dres, flags,
                       accumulate_res, (long long)total);
    LT_CHECK_LAUNCH("lt_act_bwd");
    return LT_OK;
}

extern "C" size_t lt_channel_sum_workspace(int64_t rows, int32_t C) {
    const size_t generic = (size_t)slabs_for(rows) * C * sizeof(double), fast = colsum_workspace(rows, C, 1);
    return generic > fast ? generic : fast;
}

extern "C" int lt_channel_sum(const float* x, int64_t rows, int32_t C, float* out, int32_t accumulate, void* workspace, void* stream) {
    return lt_channel_sum_dt(LT_F32, x, rows, C, out, accumulate, workspace, stream);
}

extern "C" int lt_channel_sum_dt(int32_t dtype, const void* x, int64_t rows, int32_t C, float* out, int32_t accumulate, void* workspace, void* stream) {
    LT_REQUIRE(x && out && workspace && rows >= 1 && C >= 1, LT_ERR_INVALID, "lt_channel_sum: bad argument");
    LT_REQUIRE(dtype == LT_F32 || dtype == LT_BF16, LT_ERR_INVALID, "lt_channel_sum_dt: bad dtype %d", dtype);
    const int x16 = dtype == LT_BF16 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (colsum_fast(C)) {
        const ColsumPlan p = colsum_plan(rows, C);
        hipLaunchKernelGGL(channel_sum_vec_kernel, dim3(p.nslab, p.ncb), dim3(256), 0, st, x, x16, (long long)rows, C, p.nslab, p.cw4, p.rl, (double*)workspace);
        LT_CHECK_LAUNCH("lt_channel_sum(partial)");
        hipLaunchKernelGGL(channel_sum_finalize_vec_kernel, dim3((unsigned)cdiv(C, COLSUM_FIN_C)), dim3(256), 0, st, (const double*)workspace, C, p.nslab, ChanSumFin{out, accumulate});
        LT_CHECK_LAUNCH("lt_channel_sum(finalize)");
        return LT_OK;
    }
    const int ns = slabs_for(rows);
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(ns), dim3(256), 0, st, x, x16, (long long)rows, C, ns, (double*)workspace);
    LT_CHECK_LAUNCH("lt_channel_sum(partial)");
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, (const double*)workspace, C, ns, out, accumulate);
    LT_CHECK_LAUNCH("lt_channel_sum(finalize)");
    return LT_OK;
}

extern "C" int lt_maxpool_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, const int32_t k[3],
                              const int32_t s[3], const int32_t p[3], void* stream) {
    return lt_maxpool_bwd_dt(LT_F32, x, dy, dx, N, D, H, W, C, k, s, p, stream);
}

extern "C" int lt_maxpool_bwd_dt(int32_t dtype, const void* x, const void* dy, void* dx, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, const int32_t k[3],
                                 const int32_t s[3], const int32_t p[3], void* stream) {
    LT_REQUIRE(x && dy && dx && k && s && p, LT_ERR_INVALID, "lt_maxpool_bwd: null argument");
    LT_REQUIRE(dtype == LT_F32 || dtype == LT_BF16, LT_ERR_INVALID, "lt_maxpool_bwd_dt: bad dtype %d", dtype);
    const int a16 = dtype == LT_BF16 ? 1 : 0;
    const int Do = (D + 2 * p[0] - k[0]) / s[0] + 1, Ho = (H + 2 * p[1] - k[1]) / s[1] + 1, Wo = (W + 2 * p[2] - k[2]) / s[2] + 1;
    LT_REQUIRE(N >= 1 && C >= 1 && Do >= 1 && Ho >= 1 && Wo >= 1, LT_ERR_INVALID, "lt_maxpool_bwd: bad shape");
    const int md = (int)cdiv(k[0], s[0]), mh = (int)cdiv(k[1], s[1]), mw = (int)cdiv(k[2], s[2]);
    for (int cd = 0; cd < md && cd < Do; ++cd)
        for (int ch = 0; ch < mh && ch < Ho; ++ch)
            for (int cw = 0; cw < mw && cw < Wo; ++cw) {
                PoolClass pc;
                pc.cd = cd; pc.ch = ch; pc.cw = cw; pc.md = md; pc.mh = mh; pc.mw = mw;
                pc.nd = (Do - cd + md - 1) / md; pc.nh = (Ho - ch + mh - 1) / mh; pc.nw = (Wo - cw + mw - 1) / mw;
                const long long total = (long long)N * pc.nd * pc.nh * pc.nw * C;
                const long long blocks = cdiv(total, 256);
                hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, a16, N, D, H, W, C,
                                   Do, Ho, Wo, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2], pc);
                LT_CHECK_LAUNCH("lt_maxpool_bwd");
            }
    LT_CHECK_LAUNCH("lt_maxpool_bwd");
    return LT_OK;
}

struct BrickPlan { int nbricks, bricks_per_slab, S; };

// 3x3x3, stride 1, pad 1, channel counts multiples of 32 that are their own padding, the volume a whole number of bricks
static bool brick_ok(int N, int D, int H, int W, int Cin, int Do, int Ho, int Wo, const int32_t* stride, const int32_t* pad, int Cout, int cout_pad, int k_pad, int ntaps) {
    if (ntaps != 27 || D != Do || H != Ho || W != Wo || stride[0] != 1 || stride[1] != 1 || stride[2] != 1 || pad[0] != 1 || pad[1] != 1 || pad[2] != 1) return false;
    if (Cin % 32 || Cout % 32 || cout_pad != Cout || k_pad != 27 * Cin) return false;
    return D % BRK_D == 0 && H % BRK_H == 0 && W % BRK_W == 0 && N >= 1;
}

static BrickPlan brick_plan(int N, int D, int H, int W, int Cin, int Cout, int cout_pad, int k_pad) {
    BrickPlan p;
    p.nbricks = N * (D / BRK_D) * (H / BRK_H) * (W / BRK_W);
    const long long blocks = (long long)(Cout / 32) * (Cin / 32);
    long long S = cdiv(1024, blocks);
    const long long cap = (48ll << 20) / ((long long)cout_pad * k_pad * 4);
    if (S > cap) S = cap;
    if (S > p.nbricks) S = p.nbricks;
    if (S < 1) S = 1;
    S = balance_slabs(blocks, S);
    p.bricks_per_slab = (int)cdiv(p.nbricks, S);
    p.S = (int)cdiv(p.nbricks, p.bricks_per_slab);
    return p;
}

extern "C" size_t lt_conv_wgrad_workspace(int64_t rows, int32_t cout_pad, int32_t k_pad) {
    if (rows < 1 || cout_pad < 1 || k_pad < 1) return 0;
    const WgradPlan p = wgrad_plan(rows, cout_pad, k_pad);
    size_t need = p.S > 1 ? (size_t)p.S * cout_pad * k_pad * sizeof(float) : 0;
    {          // the brick / pointwise kernels: their slab count is bounded by the same 48 MiB / 1024
        long long S = (48ll << 20) / ((long long)cout_pad * k_pad * 4);
        if (S > 1024) S = 1024;
        if (S < 1) S = 1;
        const size_t brick = (size_t)S * cout_pad * k_pad * sizeof(float);
        if (brick > need) need = brick;
    }
    return need;
}

extern "C" int lt_conv_wgrad(const float* dy, const float* x, const int32_t* taps, float* dw, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                             int32_t Do, int32_t Ho, int32_t Wo, const int32_t stride[3], const int32_t pad[3], int32_t Cout, int32_t ldy,
                             int32_t cout_pad, int32_t k_pad, int32_t ntaps, int32_t accumulate, void* workspace, void* stream) {
    LT_REQUIRE(dy && x && taps && dw && stride && pad, LT_ERR_INVALID, "lt_conv_wgrad: null argument");
    const int l2 = ilog2_exact(Cin);
    LT_REQUIRE(l2 >= 0, LT_ERR_UNSUPPORTED, "lt_conv_wgrad: Cin=%d must be a power of two", Cin);
    LT_REQUIRE(Cout >= 1 && ldy >= Cout && cout_pad >= Cout && k_pad >= ntaps * Cin && ntaps >= 1, LT_ERR_INVALID, "lt_conv_wgrad: bad sizes");
    const long long M = (long long)N * Do * Ho * Wo;
    LT_REQUIRE(M >= 1 && M < (1ll << 31) && (long long)N * D * H * W * Cin < (1ll << 31) && M * ldy < (1ll << 40), LT_ERR_UNSUPPORTED,
               "lt_conv_wgrad: too many rows / elements for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    if (ldy % 4 == 0 && brick_ok(N, D, H, W, Cin, Do, Ho, Wo, stride, pad, Cout, cout_pad, k_pad, ntaps)) {          // (float4 rows of dY)
        const BrickPlan bp = brick_plan(N, D, H, W, Cin, Cout, cout_pad, k_pad);
        LT_REQUIRE(workspace, LT_ERR_INVALID, "lt_conv_wgrad: this shape needs a workspace of lt_conv_wgrad_workspace() bytes");
        BrickArgs b;
        b.dy = dy; b.x = x; b.taps = (const int4*)taps; b.out = (float*)workspace;
        b.N = N; b.D = D; b.H = H; b.W = W; b.Cin = Cin; b.ldy = ldy; b.cout_pad = cout_pad; b.k_pad = k_pad;
        b.nbd = D / BRK_D; b.nbh = H / BRK_H; b.nbw = W / BRK_W; b.nbricks = bp.nbricks; b.bricks_per_slab = bp.bricks_per_slab;
        hipLaunchKernelGGL(conv3d_wgrad_brick_kernel, dim3(Cout / 32, Cin / 32, bp.S), dim3(256), 0, st, b);
        LT_CHECK_LAUNCH("lt_conv_wgrad(brick)");
        const long long n = (long long)cout_pad * k_pad;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid(n)), dim3(256), 0, st, (const float*)workspace, dw, n, bp.S,
                           accumulate);
        LT_CHECK_LAUNCH("lt_conv_wgrad(reduce)");
        return LT_OK;
    }
    const bool unit = stride[0] == 1 && stride[1] == 1 && stride[2] == 1 && D == Do && H == Ho && W == Wo;
    const long long cap_slabs = (48ll << 20) / ((long long)cout_pad * k_pad * 4);
    if (unit && ntaps == 1 && pad[0] == 0 && pad[1] == 0 && pad[2] == 0 && Cout % 4 == 0 && Cin % 4 == 0 && ldy % 4 == 0 && k_pad == Cin && Cout >= 64 && Cin >= 64 &&
        workspace) {
        // the bottleneck 1x1 layers: LDS-tiled GEMM
        PwArgs b;
        b.dy = dy; b.x = x; b.out = (float*)workspace; b.M = (int)M; b.Cout = Cout; b.Cin = Cin; b.ldy = ldy; b.cout_pad = cout_pad; b.k_pad = k_pad;
        b.chunks = (int)cdiv(M, PW_R);
        const long long tiles = cdiv(cout_pad, 128) * cdiv(k_pad, 128);
        long long S = cdiv(768, tiles);
        S = S > cap_slabs ? cap_slabs : S;
        S = S > b.chunks ? b.chunks : S;
        S = S < 1 ? 1 : S;
        S = balance_slabs(tiles, S);
        b.chunks_per_slab = (int)cdiv(b.chunks, S);
        S = cdiv(b.chunks, b.chunks_per_slab);
        hipLaunchKernelGGL(wgrad_pw_kernel, dim3((unsigned)cdiv(cout_pad, 128), (unsigned)cdiv(k_pad, 128), (unsigned)S), dim3(256), 0, st, b);
        LT_CHECK_LAUNCH("lt_conv_wgrad(pointwise)");
        const long long n = (long long)cout_pad * k_pad;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid(n)), dim3(256), 0, st, (const float*)workspace, dw, n, (int)S, accumulate);
        LT_CHECK_LAUNCH("lt_conv_wgrad(reduce)");
        return LT_OK;
    }
    if (unit && ntaps == 9 && D == 1 && pad[0] == 0 && pad[1] == 1 && pad[2] == 1 && H % B2_H == 0 && W % B2_W == 0 && Cout % 32 == 0 && Cin % 32 == 0 && ldy % 4 == 0 &&
        cout_pad >= Cout && k_pad == 9 * Cin && workspace) {
        // the backbone's 3x3 layers (taps in (kh, kw) order): LDS bricks
        Brick2Args b;
        b.dy = dy; b.x = x; b.out = (float*)workspace; b.N = N; b.H = H; b.W = W; b.Cin = Cin; b.Cout = Cout; b.ldy = ldy; b.cout_pad = cout_pad; b.k_pad = k_pad;
        b.nbh = H / B2_H; b.nbw = W / B2_W; b.nbricks = N * b.nbh * b.nbw;
        const long long blocks = (long long)(Cout / 32) * cdiv(Cin, 128);
        long long S = cdiv(768, blocks);
        S = S > cap_slabs ? cap_slabs : S;
        S = S > b.nbricks ? b.nbricks : S;
        S = S < 1 ? 1 : S;
        S = balance_slabs(blocks, S);
        b.bricks_per_slab = (int)cdiv(b.nbricks, S);
        S = cdiv(b.nbricks, b.bricks_per_slab);
        if (cout_pad > Cout)          // rows past Cout are not written by the kernel
            LT_REQUIRE(hipMemsetAsync(workspace, 0, (size_t)S * cout_pad * k_pad * sizeof(float), st) == hipSuccess, LT_ERR_LAUNCH, "lt_conv_wgrad: memset failed");
        hipLaunchKernelGGL(conv2d_wgrad_brick_kernel, dim3(Cout / 32, (unsigned)cdiv(Cin, 128), (unsigned)S), dim3(256), 0, st, b);
        LT_CHECK_LAUNCH("lt_conv_wgrad(brick2d)");
        const long long n = (long long)cout_pad * k_pad;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid(n)), dim3(256), 0, st, (const float*)workspace, dw, n, (int)S, accumulate);
        LT_CHECK_LAUNCH("lt_conv_wgrad(reduce)");
        return LT_OK;
    }
    if (unit && ntaps == 343 && pad[0] == 3 && pad[1] == 3 && pad[2] == 3 && Cin == 32 && Cout == 16 && cout_pad == 16 && k_pad == 343 * 32 && ldy >= 16 && ldy % 4 == 0 &&
        D % BRK_D == 0 && H % BRK_H == 0 && W % BRK_W == 0 && workspace) {
        BrickArgs b;
        b.dy = dy; b.x = x; b.taps = (const int4*)taps; b.out = (float*)workspace;
        b.N = N; b.D = D; b.H = H; b.W = W; b.Cin = Cin; b.ldy = ldy; b.cout_pad = cout_pad; b.k_pad = k_pad;
        b.nbd = D / BRK_D; b.nbh = H / BRK_H; b.nbw = W / BRK_W; b.nbricks = N * b.nbd * b.nbh * b.nbw;
        long long S = 73;                              // 7 kd planes x 73 slabs = 511 workgroups, two per CU
        S = S > cap_slabs ? cap_slabs : S;
        S = S > b.nbricks ? b.nbricks : S;
        b.bricks_per_slab = (int)cdiv(b.nbricks, S);
        S = cdiv(b.nbricks, b.bricks_per_slab);
        hipLaunchKernelGGL(conv3d_wgrad_k7_kernel, dim3(7, 1, (unsigned)S), dim3(256), 0, st, b);
        LT_CHECK_LAUNCH("lt_conv_wgrad(7^3)");
        const long long n = (long long)cout_pad * k_pad;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid(n)), dim3(256), 0, st, (const float*)workspace, dw, n, (int)S, accumulate);
        LT_CHECK_LAUNCH("lt_conv_wgrad(reduce)");
        return LT_OK;
    }
    const WgradPlan p = wgrad_plan(M, cout_pad, k_pad);
    LT_REQUIRE(p.S == 1 || workspace, LT_ERR_INVALID, "lt_conv_wgrad: this shape needs a workspace of lt_conv_wgrad_workspace() bytes");
    WgradArgs a;
    a.dy = dy; a.x = x; a.taps = (const int4*)taps; a.out = p.S > 1 ? (float*)workspace : dw;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.log2Cin = l2; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.sd = stride[0]; a.sh = stride[1]; a.sw = stride[2]; a.pd = pad[0]; a.ph = pad[1]; a.pw = pad[2];
    a.Cout = Cout; a.ldy = ldy; a.k_pad = k_pad; a.ntaps = ntaps; a.M = (int)M; a.accumulate = accumulate; a.cout_pad = cout_pad;
    a.n_k_t = p.n_k_t; a.n_tiles = p.n_co_t * p.n_k_t; a.rows_per_slab = p.rows_per_slab;

    const dim3 grid((unsigned)cdiv(a.n_tiles, 4), (unsigned)p.S);
    if (p.variant == 0) hipLaunchKernelGGL((conv_wgrad_kernel<4, 2>), grid, dim3(256), 0, st, a);
    else if (p.variant == 1) hipLaunchKernelGGL((conv_wgrad_kernel<2, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<1, 8>), grid, dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_conv_wgrad");
    if (p.S > 1) {
        const long long n = (long long)cout_pad * k_pad;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid(n)), dim3(256), 0, st, (const float*)workspace, dw, n, p.S,
                           accumulate);
        LT_CHECK_LAUNCH("lt_conv_wgrad(reduce)");
    }
    return LT_OK;
}

extern "C" int lt_adam_step_multi(const void* jobs, int32_t njobs, int32_t total_blocks, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                  void* stream) {
    LT_REQUIRE(jobs && njobs >= 1 && total_blocks >= 1 && step >= 1, LT_ERR_INVALID, "lt_adam_step_multi: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const AdamJob*)jobs, njobs, beta1, beta2, eps, weight_decay,
                       bc1, bc2);
    LT_CHECK_LAUNCH("lt_adam_step_multi");
    return LT_OK;
}

extern "C" int lt_add_f32(float* y, const float* x, int64_t n, void* stream) {
    LT_REQUIRE(y && x && n >= 1 && ((size_t)y % 16 == 0) && ((size_t)x % 16 == 0), LT_ERR_INVALID, "lt_add_f32: bad argument (16-byte aligned pointers)");
    const long long blocks = cdiv(n >> 2, 256);
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, y, x, (long long)n);
    LT_CHECK_LAUNCH("lt_add_f32");
    return LT_OK;
}

namespace {
// dst[r][c] = c < C ? src[r][c] : 0 with a change of element type on the way (fp32 <-> bf16, round to nearest even)
__global__ __launch_bounds__(256) void convert_pad_kernel(const void* __restrict__ src, int s16, void* __restrict__ dst, int d16, long long rows, int C, int Cpad) {
    const long long total = rows * Cpad;
    if (C == Cpad && (total & 3) == 0) {          // plain casts: four elements per thread
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (total >> 2); i += (long long)gridDim.x * 256) {
            const float4 v = ld4_f32_or_bf16(src, (size_t)i * 4, s16);
            st4_f32_or_bf16(dst, (size_t)i * 4, d16, v.x, v.y, v.z, v.w);
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / Cpad;
        const int c = (int)(i - r * Cpad);
        st1_f32_or_bf16(dst, (size_t)i, d16, c < C ? ld1_f32_or_bf16(src, (size_t)(r * C + c), s16) : 0.f);
    }
}
}  // namespace

extern "C" int lt_convert_pad(int32_t src_dtype, const void* src, int32_t dst_dtype, void* dst, int64_t rows, int32_t C, int32_t c_pad, void* stream) {
    LT_REQUIRE(src && dst && rows >= 1 && C >= 1 && c_pad >= C, LT_ERR_INVALID, "lt_convert_pad: bad argument");
    LT_REQUIRE((src_dtype == LT_F32 || src_dtype == LT_BF16) && (dst_dtype == LT_F32 || dst_dtype == LT_BF16), LT_ERR_INVALID, "lt_convert_pad: bad dtype");
    LT_REQUIRE(((size_t)src % 16 == 0) && ((size_t)dst % 16 == 0), LT_ERR_INVALID, "lt_convert_pad: 16-byte aligned pointers");
    const long long blocks = cdiv(rows * c_pad, 1024);
    hipLaunchKernelGGL(convert_pad_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, src,
                       src_dtype == LT_BF16 ? 1 : 0, dst, dst_dtype == LT_BF16 ? 1 : 0, (long long)rows, C, c_pad);
    LT_CHECK_LAUNCH("lt_convert_pad");
    return LT_OK;
}

extern "C" int lt_pad_channels_f32(const float* src, float* dst, int64_t rows, int32_t C, int32_t c_pad, void* stream) {
    LT_REQUIRE(src && dst && rows >= 1 && C >= 1 && c_pad >= C, LT_ERR_INVALID, "lt_pad_channels_f32: bad argument");
    const long long blocks = cdiv(rows * c_pad, 256);
    hipLaunchKernelGGL(pad_channels_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)rows, C, c_pad);
    LT_CHECK_LAUNCH("lt_pad_channels_f32");
    return LT_OK;
}

// d mean-over-the-map / d x: every pixel of sample n gets dy[n][c] / HW
__global__ __launch_bounds__(256) void global_avgpool_bwd_kernel(const void* __restrict__ dy, void* __restrict__ dx, int a16, int HW, int C, int accumulate, long long total) {
    const float inv = 1.f / (float)HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long np = i / C;
        const int c = (int)(i - np * C);
        const float g = ld1_f32_or_bf16(dy, (size_t)((np / HW) * C + c), a16) * inv;
        st1_f32_or_bf16(dx, (size_t)i, a16, accumulate ? ld1_f32_or_bf16(dx, (size_t)i, a16) + g : g);
    }
}

// BatchNorm's num_batches_tracked counters (int64 scalars scattered over the module tree): ONE launch adds delta to all of them
__global__ void add_i64_multi_kernel(long long* const* __restrict__ ptrs, int n, long long delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *ptrs[i] += delta;
}

extern "C" int lt_global_avgpool_bwd(const float* dy, float* dx, int32_t N, int32_t HW, int32_t C, int32_t accumulate, void* stream) {
    return lt_global_avgpool_bwd_dt(LT_F32, dy, dx, N, HW, C, accumulate, stream);
}

extern "C" int lt_global_avgpool_bwd_dt(int32_t dtype, const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, int32_t accumulate, void* stream) {
    LT_REQUIRE(dy && dx && N >= 1 && HW >= 1 && C >= 1 && (dtype == LT_F32 || dtype == LT_BF16), LT_ERR_INVALID, "lt_global_avgpool_bwd: bad argument");
    const long long total = (long long)N * HW * C, blocks = cdiv(total, 256);
    hipLaunchKernelGGL(global_avgpool_bwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, dy, dx, dtype == LT_BF16 ? 1 : 0, HW, C,
                       accumulate, total);
    LT_CHECK_LAUNCH("lt_global_avgpool_bwd");
    return LT_OK;
}

extern "C" int lt_add_i64_multi(const void* ptrs, int32_t n, int64_t delta, void* stream) {
    LT_REQUIRE(ptrs && n >= 1, LT_ERR_INVALID, "lt_add_i64_multi: bad argument");
    hipLaunchKernelGGL(add_i64_multi_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (long long* const*)ptrs, n, (long long)delta);
    LT_CHECK_LAUNCH("lt_add_i64_multi");
    return LT_OK;
}

extern "C" int lt_zero(void* p, int64_t nbytes, void* stream) {
    LT_REQUIRE(p && nbytes >= 1, LT_ERR_INVALID, "lt_zero: bad argument");
    LT_REQUIRE(hipMemsetAsync(p, 0, (size_t)nbytes, (hipStream_t)stream) == hipSuccess, LT_ERR_LAUNCH, "lt_zero: hipMemsetAsync failed");
    return LT_OK;
}

extern "C" int lt_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    LT_REQUIRE(src && dst && n >= 1 && ((size_t)src % 16 == 0) && ((size_t)dst % 16 == 0), LT_ERR_INVALID, "lt_cast_f32_bf16: bad argument (16-byte aligned pointers)");
    const long long blocks = cdiv(n >> 3, 256);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long long)n);
    LT_CHECK_LAUNCH("lt_cast_f32_bf16");
    return LT_OK;
}

extern "C" int lt_gather_f32_multi(const void* jobs, int32_t njobs, int32_t total_blocks, void* stream) {
    LT_REQUIRE(jobs && njobs >= 1 && total_blocks >= 1, LT_ERR_INVALID, "lt_gather_f32_multi: bad argument");
    hipLaunchKernelGGL(gather_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const GatherJob*)jobs, njobs);
    LT_CHECK_LAUNCH("lt_gather_f32_multi");
    return LT_OK;
}

extern "C" int lt_gather_f32_bf16(const float* src, const int32_t* idx, void* dst_bf16, int64_t n, void* stream) {
    LT_REQUIRE(src && idx && dst_bf16 && n >= 1, LT_ERR_INVALID, "lt_gather_f32_bf16: bad argument");
    const long long blocks = cdiv(n, 256);
    hipLaunchKernelGGL(gather_bf16_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, src, (const int*)idx, (bf16_t*)dst_bf16,
                       (long long)n);
    LT_CHECK_LAUNCH("lt_gather_f32_bf16");
    return LT_OK;
}

extern "C" int lt_gather_f32(const float* src, const int32_t* idx, float* dst, int64_t n, void* stream) {
    LT_REQUIRE(src && idx && dst && n >= 1, LT_ERR_INVALID, "lt_gather_f32: bad argument");
    const long long blocks = cdiv(n, 256);
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, src, (const int*)idx, dst, (long long)n);
    LT_CHECK_LAUNCH("lt_gather_f32");
    return LT_OK;
}

extern "C" int lt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t step, void* stream) {
    LT_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 1 && step >= 1, LT_ERR_INVALID, "lt_adam_step: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    const long long blocks = cdiv(n, 256);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    LT_CHECK_LAUNCH("lt_adam_step");
    return LT_OK;
}
