// Backward kernels of the non-convolution ops of the volumetric path and the training-mode BatchNorm statistics
// (BASELINE config 5): what torch.autograd derives for the reference from
//   op.unproject_heatmaps (mvn/utils/op.py:99-166)            -> lt_unproject_bwd (csrc/unproject_bwd.hip)
//   op.integrate_tensor_3d_with_coordinates (op.py:84-96)      -> lt_softargmax3d_bwd
//   VolumetricCELoss (mvn/models/loss.py:52-80)                -> lt_volumetric_ce_fwd (+ its sparse gradient, consumed by lt_softargmax3d_bwd)
//   nn.BatchNorm{2,3}d in training mode (batch statistics)     -> lt_bn_stats_fwd
// Gradients are fp32.  Layouts are the forward's: channels-last feature maps / volumes, planar probabilities (B, J, V^3).
#include <type_traits>

#include "colsum.h"

using namespace lt;

namespace {

// ---- 3D soft-argmax backward -----------------------------------------------------------------------------------------------
// forward: p = softmax_i(mult * l_i), kp = sum_i p_i X_i  (softmax = 0: p_i = relu(mult * l_i), no normalisation)
// given g_kp (B, J, 3) and an optional SPARSE gradient on the returned probabilities (what VolumetricCELoss produces: one voxel per
// (b, j)): a_i = g_kp . X_i + gp_i;  d L / d l_i = mult * p_i * (a_i - sum_j p_j a_j) = mult * p_i * (a_i - g_kp . kp - p[idx] gp)
// (ReLU mode: mult * [l_i > 0] * a_i).  One elementwise pass over the planar probabilities; the logits gradient is written in
// the layout the V2V backward wants (planar (B, J, nvox) or channels-last (B, nvox, J)).
struct SA3BwdArgs {
    const float* probs;      // (B, J, nvox): the forward's second output
    const float* coords;     // (B, nvox, 3)
    const float* kp;         // (B, J, 3): the forward's first output
    const float* gkp;        // (B, J, 3)
    const int* gp_idx;       // (B, J) voxel index, or null
    const float* gp_val;     // (B, J) gradient on probs[b, j, idx]
    const float* gp_dense;   // (B, J, nvox) DENSE gradient on the returned probabilities (any loss on the volumes, train.py:222-230), or null
    const float* pgsum;      // (B, J): sum_i p_i gp_dense_i (sa3_pg_kernel), softmax mode with gp_dense
    float* glogits;
    float mult;
    int softmax, J, channels_last;
    long long nvox;
};

__global__ __launch_bounds__(256) void sa3_bwd_kernel(const SA3BwdArgs a) {
    const int bj = blockIdx.y, b = bj / a.J, j = bj - b * a.J;
    const float* p = a.probs + (long long)bj * a.nvox;
    const float* cd = a.coords + (long long)b * a.nvox * 3;
    const float g0 = a.gkp[bj * 3], g1 = a.gkp[bj * 3 + 1], g2 = a.gkp[bj * 3 + 2];
    long long idx = -1;
    float gv = 0.f;
    if (a.gp_idx) { idx = a.gp_idx[bj]; gv = a.gp_val[bj]; }
    float sub = 0.f;
    if (a.softmax) {
        sub = g0 * a.kp[bj * 3] + g1 * a.kp[bj * 3 + 1] + g2 * a.kp[bj * 3 + 2];
        if (idx >= 0) sub += p[idx] * gv;
        if (a.gp_dense) sub += a.pgsum[bj];
    }
    const float* gd = a.gp_dense ? a.gp_dense + (long long)bj * a.nvox : nullptr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.nvox; i += (long long)gridDim.x * 256) {
        const float pi = p[i];
        float ai = g0 * cd[i * 3] + g1 * cd[i * 3 + 1] + g2 * cd[i * 3 + 2];
        if (i == idx) ai += gv;
        if (gd) ai += gd[i];
        float gl;
        if (a.softmax) gl = a.mult * pi * (ai - sub);
        else gl = pi > 0.f ? a.mult * ai : 0.f;
        if (a.channels_last) a.glogits[((long long)b * a.nvox + i) * a.J + j] = gl;
        else a.glogits[(long long)bj * a.nvox + i] = gl;
    }
}

// sum_i p_i g_i per (b, j), fp64 accumulation in a fixed order (one workgroup per (b, j): lanes stride the voxels, LDS tree)
__global__ __launch_bounds__(256) void sa3_pg_kernel(const float* __restrict__ probs, const float* __restrict__ gp, long long nvox, float* __restrict__ out) {
    __shared__ double red[256];
    const long long base = (long long)blockIdx.x * nvox;
    double s = 0.0;
    for (long long i = threadIdx.x; i < nvox; i += 256) s += (double)probs[base + i] * (double)gp[base + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)red[0];
}

// ---- VolumetricCELoss, fused (loss.py:52-80) -----------------------------------------------------------------------------------
// per (b, j): idx = argmin_i |X_i - gt_bj| (first minimum, like torch.argmin), term = validity * -log(p[idx] + 1e-6);
// loss = sum(terms) / (B * J) (n_losses counts every joint, valid or not).  The reference syncs the host once per sample
// (.cpu() at loss.py:70) and walks the joints in Python; here: one workgroup per (b, j), no host round trip.
struct CEArgs {
    const float* coords;     // (B, nvox, 3)
    const float* probs;      // (B, J, nvox)
    const float* gt;         // (B, J, 3)
    const float* validity;   // (B, J)
    float* terms;            // (B, J): validity * -log(p + 1e-6)
    int* idx;                // (B, J)
    float* gval;             // (B, J): d loss / d probs[b, j, idx] for an upstream gradient of 1 = -validity / (p + 1e-6) / (B J)
    int J, n_losses;
    long long nvox;
};

__global__ __launch_bounds__(256) void vol_ce_kernel(const CEArgs a) {
    __shared__ float sd[256];
    __shared__ long long si[256];
    const int bj = blockIdx.x, b = bj / a.J;
    const float* cd = a.coords + (long long)b * a.nvox * 3;
    const float gx = a.gt[bj * 3], gy = a.gt[bj * 3 + 1], gz = a.gt[bj * 3 + 2];
    float best = INFINITY;
    long long bi = 0;
    for (long long i = threadIdx.x; i < a.nvox; i += 256) {
        // torch: sqrt(((X - gt) ** 2).sum(-1)); the argmin of the sqrt is the argmin of the sum up to ties created by the
        // rounding of sqrt -- compare what torch compares
        const float dx = __fsub_rn(cd[i * 3], gx), dy = __fsub_rn(cd[i * 3 + 1], gy), dz = __fsub_rn(cd[i * 3 + 2], gz);
        const float d = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        if (d < best) { best = d; bi = i; }               // ascending i per lane: keeps the first minimum
    }
    sd[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) {
            const float d2 = sd[threadIdx.x + s];
            const long long i2 = si[threadIdx.x + s];
            if (d2 < sd[threadIdx.x] || (d2 == sd[threadIdx.x] && i2 < si[threadIdx.x])) { sd[threadIdx.x] = d2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const long long i = si[0];
        const float p = a.probs[(long long)bj * a.nvox + i];
        const float v = a.validity[bj];
        a.idx[bj] = (int)i;
        a.terms[bj] = v * -logf(p + 1e-6f);
        a.gval[bj] = -v / (p + 1e-6f) / (float)a.n_losses;
    }
}

// ---- BatchNorm batch statistics (training mode) -----------------------------------------------------------------------------------
// x: channels-last rows x C (T); per channel mean and BIASED variance over the rows (what F.batch_norm normalises with in training mode),
// accumulated in fp64 (two-pass-free: sum and sum of squares of fp32 values in fp64 is exact enough for 1e7 rows).
// Stage 1: workgroups own row slabs and write per-slab partials; stage 2 combines.  C <= 2048.
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, long long rows, int C, int nslab, double* __restrict__ part) {
    // the 256 threads of a slab are laid out as RL row lanes x Cw channels (Cw = min(C, 256)): consecutive threads read consecutive
    // channels of one row (coalesced), the row lanes walk the slab's rows RL apart and are combined through LDS at the end
    __shared__ double ss[256], sq[256];
    const int slab = blockIdx.x;
    const long long r0 = rows * slab / nslab, r1 = rows * (slab + 1) / nslab;
    const int Cw = C < 256 ? C : 256, RL = 256 / Cw;
    const int rl = threadIdx.x / Cw, cl = threadIdx.x - rl * Cw;
    for (int c0 = 0; c0 < C; c0 += Cw) {
        const int c = c0 + cl;
        double s = 0.0, q = 0.0;
        if (rl < RL && c < C)
            for (long long r = r0 + rl; r < r1; r += RL) {
                const double v = (double)elt<T>::ld(x + r * C + c);
                s += v; q += v * v;
            }
        ss[threadIdx.x] = s; sq[threadIdx.x] = q;
        __syncthreads();
        if (rl == 0 && c < C) {
            for (int k = 1; k < RL; ++k) { s += ss[k * Cw + cl]; q += sq[k * Cw + cl]; }
            part[((long long)slab * C + c) * 2] = s;
            part[((long long)slab * C + c) * 2 + 1] = q;
        }
        __syncthreads();
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ part, long long rows, int C, int nslab, float* __restrict__ mean, float* __restrict__ var,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < nslab; ++k) { s += part[((long long)k * C + c) * 2]; q += part[((long long)k * C + c) * 2 + 1]; }
    const double m = s / (double)rows;
    double v = q / (double)rows - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m;
    var[c] = (float)v;
    if (running_mean) {   // torch: running = (1 - momentum) * running + momentum * stat, the variance UNBIASED (n / (n - 1))
        const double unb = rows > 1 ? v * (double)rows / (double)(rows - 1) : v;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
}

template <bool BF16, int NROWS>
struct BnStatLoad {
    typedef typename std::conditional<BF16, uint2, float4>::type Raw;
    static constexpr int ROWS = NROWS;
    const void* x; int C;
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ Raw fetch(long long row, int c) const { return *(const Raw*)((const char*)x + ((size_t)row * C + c) * (BF16 ? 2 : 4)); }
    __device__ __forceinline__ void eval(const Raw& r, int, float (&q)[2][4]) const {
        float4 v;
        if constexpr (BF16) v = bf16x4_to_f32(r);
        else v = r;
        q[0][0] = v.x; q[0][1] = v.y; q[0][2] = v.z; q[0][3] = v.w;
        q[1][0] = v.x * v.x; q[1][1] = v.y * v.y; q[1][2] = v.z * v.z; q[1][3] = v.w * v.w;
    }
};

template <bool BF16, int NROWS>
__global__ __launch_bounds__(256) void bn_partial_vec_kernel(const void* __restrict__ x, long long rows, int C, int nslab, int cw4, int rl, double* __restrict__ part) {
    colsum_partial<2>(rows, C, nslab, cw4, rl, part, BnStatLoad<BF16, NROWS>{x, C});
}

struct BnStatFin {
    long long rows; float* mean; float* var; float* running_mean; float* running_var; float momentum;
    __device__ __forceinline__ void operator()(int c, const double (&t)[2]) const {
        const double m = t[0] / (double)rows;
        double v = t[1] / (double)rows - m * m;
        if (v < 0.0) v = 0.0;
        mean[c] = (float)m;
        var[c] = (float)v;
        if (running_mean) {   // torch: running = (1 - momentum) * running + momentum * stat, the variance UNBIASED (n / (n - 1))
            const double unb = rows > 1 ? v * (double)rows / (double)(rows - 1) : v;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        }
    }
};

__global__ __launch_bounds__(256) void bn_finalize_vec_kernel(const double* __restrict__ part, int C, int nslab, BnStatFin fin) {
    colsum_finalize<2>(part, C, nslab, fin);
}

}  // namespace

extern "C" int lt_softargmax3d_bwd(const float* probs, const float* coords, const float* kp, const float* grad_kp, const int32_t* gp_idx,
                                   const float* gp_val, float multiplier, int32_t softmax, int32_t channels_last, float* grad_logits, int32_t B,
                                   int32_t J, int64_t nvox, void* stream) {
    return lt_softargmax3d_bwd_dense(probs, coords, kp, grad_kp, gp_idx, gp_val, nullptr, nullptr, multiplier, softmax, channels_last, grad_logits, B, J, nvox, stream);
}

extern "C" int lt_softargmax3d_bwd_dense(const float* probs, const float* coords, const float* kp, const float* grad_kp, const int32_t* gp_idx,
                                         const float* gp_val, const float* gp_dense, float* workspace_bj, float multiplier, int32_t softmax,
                                         int32_t channels_last, float* grad_logits, int32_t B, int32_t J, int64_t nvox, void* stream) {
    LT_REQUIRE(probs && coords && kp && grad_kp && grad_logits, LT_ERR_INVALID, "lt_softargmax3d_bwd: null argument");
    LT_REQUIRE(!gp_dense || !softmax || workspace_bj, LT_ERR_INVALID, "lt_softargmax3d_bwd_dense: the softmax mode needs B * J floats of workspace");
    LT_REQUIRE((gp_idx == nullptr) == (gp_val == nullptr), LT_ERR_INVALID, "lt_softargmax3d_bwd: gp_idx and gp_val come together");
    LT_REQUIRE(B >= 1 && J >= 1 && nvox >= 1 && (long long)B * J < 65536, LT_ERR_INVALID, "lt_softargmax3d_bwd: bad shape");
    SA3BwdArgs a;
    a.probs = probs; a.coords = coords; a.kp = kp; a.gkp = grad_kp; a.gp_idx = gp_idx; a.gp_val = gp_val; a.glogits = grad_logits;
    a.gp_dense = gp_dense; a.pgsum = workspace_bj;
    if (gp_dense && softmax) {
        hipLaunchKernelGGL(sa3_pg_kernel, dim3((unsigned)(B * J)), dim3(256), 0, (hipStream_t)stream, probs, gp_dense, (long long)nvox, workspace_bj);
        LT_CHECK_LAUNCH("lt_softargmax3d_bwd_dense(sum)");
    }
    a.mult = multiplier; a.softmax = softmax; a.J = J; a.channels_last = channels_last; a.nvox = nvox;
    const long long blocks = cdiv(nvox, 256 * 8);
    hipLaunchKernelGGL(sa3_bwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), (unsigned)(B * J)), dim3(256), 0, (hipStream_t)stream, a);
    LT_CHECK_LAUNCH("lt_softargmax3d_bwd");
    return LT_OK;
}

extern "C" int lt_volumetric_ce_fwd(const float* coords, const float* probs, const float* keypoints_gt, const float* validity, float* terms,
                                    int32_t* idx, float* grad_val, int32_t B, int32_t J, int64_t nvox, void* stream) {
    LT_REQUIRE(coords && probs && keypoints_gt && validity && terms && idx && grad_val, LT_ERR_INVALID, "lt_volumetric_ce_fwd: null argument");
    LT_REQUIRE(B >= 1 && J >= 1 && nvox >= 1 && nvox < (1ll << 31), LT_ERR_INVALID, "lt_volumetric_ce_fwd: bad shape");
    CEArgs a;
    a.coords = coords; a.probs = probs; a.gt = keypoints_gt; a.validity = validity; a.terms = terms; a.idx = idx; a.gval = grad_val;
    a.J = J; a.n_losses = B * J; a.nvox = nvox;
    hipLaunchKernelGGL(vol_ce_kernel, dim3((unsigned)(B * J)), dim3(256), 0, (hipStream_t)stream, a);
    LT_CHECK_LAUNCH("lt_volumetric_ce_fwd");
    return LT_OK;
}

extern "C" size_t lt_bn_stats_workspace(int64_t rows, int32_t C) {
    const long long nslab = rows < 1024 ? 1 : (rows / 256 < 1024 ? rows / 256 : 1024);
    const size_t generic = (size_t)nslab * C * 2 * sizeof(double), fast = colsum_workspace(rows, C, 2);
    return generic > fast ? generic : fast;
}

extern "C" int lt_bn_stats_fwd(int32_t dtype, const void* x, int64_t rows, int32_t C, float* mean, float* var, float* running_mean,
                               float* running_var, float momentum, void* workspace, void* stream) {
    LT_REQUIRE(x && mean && var && workspace, LT_ERR_INVALID, "lt_bn_stats_fwd: null argument");
    LT_REQUIRE(dtype == LT_F32 || dtype == LT_BF16, LT_ERR_INVALID, "lt_bn_stats_fwd: bad dtype %d", dtype);
    LT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), LT_ERR_INVALID, "lt_bn_stats_fwd: running_mean and running_var come together");
    LT_REQUIRE(rows >= 1 && C >= 1 && C <= 4096, LT_ERR_INVALID, "lt_bn_stats_fwd: bad shape rows=%lld C=%d", (long long)rows, C);
    hipStream_t st = (hipStream_t)stream;
    if (colsum_fast(C)) {       // the training path: four-channel lanes, four rows in flight, parallel finalize (colsum.h); fp32 or bf16 input
        const ColsumPlan p = colsum_plan(rows, C);
        static const int nrows = [] { const char* e = getenv("LT_BNSTAT_ROWS"); return e ? atoi(e) : 8; }();
        if (dtype == LT_BF16 && nrows == 4) hipLaunchKernelGGL((bn_partial_vec_kernel<true, 4>), dim3(p.nslab, p.ncb), dim3(256), 0, st, x, (long long)rows, C, p.nslab, p.cw4, p.rl, (double*)workspace);
        else if (dtype == LT_BF16) hipLaunchKernelGGL((bn_partial_vec_kernel<true, 8>), dim3(p.nslab, p.ncb), dim3(256), 0, st, x, (long long)rows, C, p.nslab, p.cw4, p.rl, (double*)workspace);
        else hipLaunchKernelGGL((bn_partial_vec_kernel<false, 8>), dim3(p.nslab, p.ncb), dim3(256), 0, st, x, (long long)rows, C, p.nslab, p.cw4, p.rl, (double*)workspace);
        LT_CHECK_LAUNCH("lt_bn_stats_fwd(partial)");
        hipLaunchKernelGGL(bn_finalize_vec_kernel, dim3((unsigned)cdiv(C, COLSUM_FIN_C)), dim3(256), 0, st, (const double*)workspace, C, p.nslab,
                           BnStatFin{(long long)rows, mean, var, running_mean, running_var, momentum});
        LT_CHECK_LAUNCH("lt_bn_stats_fwd(finalize)");
        return LT_OK;
    }
    const int nslab = (int)(rows < 1024 ? 1 : (rows / 256 < 1024 ? rows / 256 : 1024));
    if (dtype == LT_F32) hipLaunchKernelGGL(bn_partial_kernel<float>, dim3(nslab), dim3(256), 0, st, (const float*)x, (long long)rows, C, nslab, (double*)workspace);
    else hipLaunchKernelGGL(bn_partial_kernel<bf16_t>, dim3(nslab), dim3(256), 0, st, (const bf16_t*)x, (long long)rows, C, nslab, (double*)workspace);
    LT_CHECK_LAUNCH("lt_bn_stats_fwd(partial)");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, (const double*)workspace, (long long)rows, C, nslab, mean, var,
                       running_mean, running_var, momentum);
    LT_CHECK_LAUNCH("lt_bn_stats_fwd(finalize)");
    return LT_OK;
}
