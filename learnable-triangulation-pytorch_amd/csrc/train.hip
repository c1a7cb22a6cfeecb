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
//   lt_adam_step       torch.optim.Adam's update (train.py:430-437), one launch per parameter tensor
// The convolution dgrad needs no kernel of its own: it is lt_conv_fwd over dY with the weights transposed / flipped (stride 1), as a
// parity-phase transposed convolution (stride-2 layers) or as a strided convolution (the transposed layers) -- see lt_train.py.
#include "conv_common.h"

using namespace lt;

namespace {

__device__ __forceinline__ float relu_mask_post(float v) { return v > 0.f ? 1.f : 0.f; }

struct BnActArgs {
    const float* y;        // rows x C: convolution output (bias included)
    const float* mean; const float* var; const float* gamma; const float* beta;
    const float* res;      // rows x C or null
    float* z;
    float eps;
    int flags, C;
    long long rows;
};

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const BnActArgs a) {
    const long long total = a.rows * (a.C >> 2);
    const EpiFloors fl = epi_floors(a.flags);
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const long long row = g / (a.C >> 2);
        const int c = (int)(g - row * (a.C >> 2)) * 4;
        const size_t off = (size_t)row * a.C + c;
        const float4 y4 = *(const float4*)(a.y + off);
        const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
        float rr[4] = {-0.0f, -0.0f, -0.0f, -0.0f};
        if (a.res) { const float4 r4 = *(const float4*)(a.res + off); rr[0] = r4.x; rr[1] = r4.y; rr[2] = r4.z; rr[3] = r4.w; }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // ATen's batch_norm (training): (x - mean) * invstd * weight + bias with invstd = 1 / sqrt(var + eps) in fp32
            const float invstd = 1.0f / sqrtf(a.var[c + e] + a.eps);
            const float v = (yy[e] - a.mean[c + e]) * invstd * a.gamma[c + e] + a.beta[c + e];
            o[e] = epi_apply(v, fl, rr[e]);
        }
        *(float4*)(a.z + off) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

struct BnBwdArgs {
    const float* dz; const float* y; const float* res;     // rows x C
    const float* mean; const float* var; const float* gamma; const float* beta;
    double* part;            // [nslab][C][2]: sum g, sum g x^
    float* dgamma; float* dbeta;
    float* dy;               // rows x C
    float* dres;             // rows x C or null: gradient of the residual input (accumulated when accumulate_res)
    float eps;
    int flags, C, nslab, accumulate_res;
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
                const float g = bn_g(a, a.dz[off], a.y[off], a.res ? a.res[off] : 0.f, c, xh);
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
    const float inv_n = 1.0f / (float)a.rows;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % a.C);
        float xh;
        const float r = a.res ? a.res[i] : 0.f;
        const float dzv = a.dz[i];
        const float g = bn_g(a, dzv, a.y[i], r, c, xh);
        const float invstd = 1.0f / sqrtf(a.var[c] + a.eps);
        a.dy[i] = a.gamma[c] * invstd * (g - a.dbeta[c] * inv_n - xh * a.dgamma[c] * inv_n);
        if (a.dres) {
            const float dr = (a.flags & LT_EPI_RELU_POST) ? g : dzv;     // RELU_PRE / none: the residual is added after the activation
            a.dres[i] = a.accumulate_res ? a.dres[i] + dr : dr;
        }
    }
}

// layers without BatchNorm: z = act(y, res) with y = conv + bias: dy = dz * mask, dres likewise
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ res,
                                                      float* __restrict__ dy, float* __restrict__ dres, int flags, int accumulate_res, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float m = 1.f;
        if (flags & LT_EPI_RELU_POST) m = z[i] > 0.f ? 1.f : 0.f;                          // z = relu(v + res)
        else if (flags & LT_EPI_RELU_PRE) m = (z[i] - (res ? res[i] : 0.f)) > 0.f ? 1.f : 0.f;   // z = relu(v) + res
        const float g = dz[i] * m;
        dy[i] = g;
        if (dres) {
            const float dr = (flags & LT_EPI_RELU_POST) ? g : dz[i];
            dres[i] = accumulate_res ? dres[i] + dr : dr;
        }
    }
}

__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ x, long long rows, int C, int nslab, double* __restrict__ part) {
    __shared__ double ss[256];
    const int slab = blockIdx.x;
    const long long r0 = rows * slab / nslab, r1 = rows * (slab + 1) / nslab;
    const int Cw = C < 256 ? C : 256, RL = 256 / Cw;
    const int rl = threadIdx.x / Cw, cl = threadIdx.x - rl * Cw;
    for (int c0 = 0; c0 < C; c0 += Cw) {
        const int c = c0 + cl;
        double s = 0.0;
        if (rl < RL && c < C)
            for (long long r = r0 + rl; r < r1; r += RL) s += (double)x[(size_t)r * C + c];
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
// ATen's CPU max_pool) and atomically adds dy there (windows overlap for k > s)
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int N, int D, int H, int W, int C,
                                   int Do, int Ho, int Wo, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw) {
    const long long total = (long long)N * Do * Ho * Wo * C;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(g % C);
        long long r = g / C;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int od = (int)(r % Do);
        const int n = (int)(r / Do);
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
                    const float v = x[idx];
                    if (v > best || bi < 0) { best = v; bi = idx; }
                }
            }
        }
        if (bi >= 0) unsafeAtomicAdd(dx + bi, dy[g]);
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
    const int4* taps;        // [ntaps] = (dd, dh, dw, element offset) as in lt_conv_fwd
    float* dw;               // [cout_pad][k_pad] fp32 (rows >= Cout / columns >= ntaps * Cin are written as 0)
    int N, D, H, W, Cin, log2Cin, Do, Ho, Wo, sd, sh, sw, pd, ph, pw;
    int Cout, ldy, k_pad, ntaps, M, accumulate, cout_pad;
};

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    __shared__ float red[3][4][16][64];                   // waves 1-3: [k block][acc element][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co0 = blockIdx.x * 32, k0 = blockIdx.y * 128;
    const int col = lane & 31, half = lane >> 5;          // A: co = co0 + col, row m + half;  B: k = k0 + 32 j + col, row m + half
    const bool co_ok = co0 + col < a.Cout;
    int tap_dd[4], tap_dh[4], tap_dw[4], ci[4];
    bool k_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + 32 * j + col;
        const int tap = k >> a.log2Cin;
        k_ok[j] = tap < a.ntaps;
        const int4 tp = k_ok[j] ? a.taps[tap] : make_int4(0, 0, 0, 0);
        tap_dd[j] = tp.x; tap_dh[j] = tp.y; tap_dw[j] = tp.z;
        ci[j] = k & (a.Cin - 1);
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int hw = a.Ho * a.Wo, dhw = a.Do * hw;
    for (int m2 = wave * 2; m2 < a.M; m2 += 8) {
        const int m = m2 + half;
        const bool m_ok = m < a.M;
        float av = 0.f, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (m_ok) {
            if (co_ok) av = a.dy[(size_t)m * a.ldy + co0 + col];
            const int n = m / dhw;
            int r = m - n * dhw;
            const int od = r / hw; r -= od * hw;
            const int oh = r / a.Wo, ow = r - oh * a.Wo;
            const int id0 = od * a.sd - a.pd, ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = id0 + tap_dd[j], ih = ih0 + tap_dh[j], iw = iw0 + tap_dw[j];
                if (k_ok[j] && (unsigned)id < (unsigned)a.D && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
                    bv[j] = a.x[((((size_t)n * a.D + id) * a.H + ih) * a.W + iw) * a.Cin + ci[j]];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave - 1][j][e][lane] = acc[j][e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = acc[j][e] + red[0][j][e][lane] + red[1][j][e][lane] + red[2][j][e][lane];
                const int row = co0 + 8 * (e >> 2) + 4 * half + (e & 3);       // C layout of the 32x32 MFMA: row = co, column = k
                const int k = k0 + 32 * j + col;
                if (k < a.k_pad && row < a.cout_pad) {
                    float* dst = a.dw + (size_t)row * a.k_pad + k;
                    *dst = a.accumulate ? *dst + v : v;
                }
            }
    }
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

int slabs_for(long long rows) { return (int)(rows < 1024 ? 1 : (rows / 256 < 1024 ? rows / 256 : 1024)); }

}  // namespace

extern "C" int lt_bn_act_fwd(const float* y, const float* mean, const float* var, const float* gamma, const float* beta, const float* residual,
                             float* z, int64_t rows, int32_t C, float eps, int32_t flags, void* stream) {
    LT_REQUIRE(y && mean && var && gamma && beta && z, LT_ERR_INVALID, "lt_bn_act_fwd: null argument");
    LT_REQUIRE(rows >= 1 && C >= 4 && C % 4 == 0, LT_ERR_UNSUPPORTED, "lt_bn_act_fwd: C %% 4 == 0 required (C=%d)", C);
    BnActArgs a;
    a.y = y; a.mean = mean; a.var = var; a.gamma = gamma; a.beta = beta; a.res = residual; a.z = z; a.eps = eps; a.flags = flags; a.C = C; a.rows = rows;
    const long long blocks = cdiv(rows * (C / 4), 256);
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, a);
    LT_CHECK_LAUNCH("lt_bn_act_fwd");
    return LT_OK;
}

extern "C" size_t lt_bn_act_bwd_workspace(int64_t rows, int32_t C) { return (size_t)slabs_for(rows) * C * 2 * sizeof(double); }

extern "C" int lt_bn_act_bwd(const float* dz, const float* y, const float* residual, const float* mean, const float* var, const float* gamma,
                             const float* beta, float* dy, float* dgamma, float* dbeta, float* dres, int32_t accumulate_res, int64_t rows, int32_t C,
                             float eps, int32_t flags, void* workspace, void* stream) {
    LT_REQUIRE(dz && y && mean && var && gamma && beta && dy && dgamma && dbeta && workspace, LT_ERR_INVALID, "lt_bn_act_bwd: null argument");
    LT_REQUIRE(rows >= 1 && C >= 1 && C <= 4096, LT_ERR_INVALID, "lt_bn_act_bwd: bad shape");
    LT_REQUIRE(!dres || residual, LT_ERR_INVALID, "lt_bn_act_bwd: a residual gradient needs the residual");
    BnBwdArgs a;
    a.dz = dz; a.y = y; a.res = residual; a.mean = mean; a.var = var; a.gamma = gamma; a.beta = beta; a.part = (double*)workspace;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dy = dy; a.dres = dres; a.eps = eps; a.flags = flags; a.C = C; a.nslab = slabs_for(rows);
    a.accumulate_res = accumulate_res; a.rows = rows;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(a.nslab), dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_bn_act_bwd(reduce)");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_bn_act_bwd(finalize)");
    const long long blocks = cdiv(rows * C, 256);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_bn_act_bwd(apply)");
    return LT_OK;
}

extern "C" int lt_act_bwd(const float* dz, const float* z, const float* residual, float* dy, float* dres, int32_t accumulate_res, int64_t total,
                          int32_t flags, void* stream) {
    LT_REQUIRE(dz && z && dy && total >= 1, LT_ERR_INVALID, "lt_act_bwd: bad argument");
    const long long blocks = cdiv(total, 256);
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, dz, z, residual, dy, dres, flags,
                       accumulate_res, (long long)total);
    LT_CHECK_LAUNCH("lt_act_bwd");
    return LT_OK;
}

extern "C" size_t lt_channel_sum_workspace(int64_t rows, int32_t C) { return (size_t)slabs_for(rows) * C * sizeof(double); }

extern "C" int lt_channel_sum(const float* x, int64_t rows, int32_t C, float* out, int32_t accumulate, void* workspace, void* stream) {
    LT_REQUIRE(x && out && workspace && rows >= 1 && C >= 1, LT_ERR_INVALID, "lt_channel_sum: bad argument");
    const int ns = slabs_for(rows);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(ns), dim3(256), 0, st, x, (long long)rows, C, ns, (double*)workspace);
    LT_CHECK_LAUNCH("lt_channel_sum(partial)");
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, (const double*)workspace, C, ns, out, accumulate);
    LT_CHECK_LAUNCH("lt_channel_sum(finalize)");
    return LT_OK;
}

extern "C" int lt_maxpool_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, const int32_t k[3],
                              const int32_t s[3], const int32_t p[3], void* stream) {
    LT_REQUIRE(x && dy && dx && k && s && p, LT_ERR_INVALID, "lt_maxpool_bwd: null argument");
    const int Do = (D + 2 * p[0] - k[0]) / s[0] + 1, Ho = (H + 2 * p[1] - k[1]) / s[1] + 1, Wo = (W + 2 * p[2] - k[2]) / s[2] + 1;
    LT_REQUIRE(N >= 1 && C >= 1 && Do >= 1 && Ho >= 1 && Wo >= 1, LT_ERR_INVALID, "lt_maxpool_bwd: bad shape");
    const long long total = (long long)N * Do * Ho * Wo * C;
    const long long blocks = cdiv(total, 256);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, D, H, W, C, Do, Ho,
                       Wo, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2]);
    LT_CHECK_LAUNCH("lt_maxpool_bwd");
    return LT_OK;
}

extern "C" int lt_conv_wgrad(const float* dy, const float* x, const int32_t* taps, float* dw, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin,
                             int32_t Do, int32_t Ho, int32_t Wo, const int32_t stride[3], const int32_t pad[3], int32_t Cout, int32_t ldy,
                             int32_t cout_pad, int32_t k_pad, int32_t ntaps, int32_t accumulate, void* stream) {
    LT_REQUIRE(dy && x && taps && dw && stride && pad, LT_ERR_INVALID, "lt_conv_wgrad: null argument");
    const int l2 = ilog2_exact(Cin);
    LT_REQUIRE(l2 >= 0, LT_ERR_UNSUPPORTED, "lt_conv_wgrad: Cin=%d must be a power of two", Cin);
    LT_REQUIRE(Cout >= 1 && ldy >= Cout && cout_pad >= Cout && k_pad >= ntaps * Cin && ntaps >= 1, LT_ERR_INVALID, "lt_conv_wgrad: bad sizes");
    const long long M = (long long)N * Do * Ho * Wo;
    LT_REQUIRE(M >= 1 && M < (1ll << 31), LT_ERR_UNSUPPORTED, "lt_conv_wgrad: too many rows");
    WgradArgs a;
    a.dy = dy; a.x = x; a.taps = (const int4*)taps; a.dw = dw;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.log2Cin = l2; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.sd = stride[0]; a.sh = stride[1]; a.sw = stride[2]; a.pd = pad[0]; a.ph = pad[1]; a.pw = pad[2];
    a.Cout = Cout; a.ldy = ldy; a.k_pad = k_pad; a.ntaps = ntaps; a.M = (int)M; a.accumulate = accumulate; a.cout_pad = cout_pad;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)cdiv(cout_pad, 32), (unsigned)cdiv(k_pad, 128)), dim3(256), 0, (hipStream_t)stream, a);
    LT_CHECK_LAUNCH("lt_conv_wgrad");
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
