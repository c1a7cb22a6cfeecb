// Voxel-grid construction and the fused unprojection (projective bilinear gather + view aggregation).
//
// Unprojection is HBM/L2-bound: per sample it must read the NV feature maps once and write the
// C x V^3 volume once (38.27 MB fp32 at config 2); everything else (projection,
// 4-tap weights, view softmax) lives in registers.  Layout choices that make it stream:
//   * feature maps are channels-last, so ONE bilinear tap of ONE voxel is a contiguous C-vector
//     (128 B fp32 / 64 B bf16 at C=32): the C/CH lanes of a voxel issue one coalesced 16-byte load each;
//   * the volume is written channels-last too (what the V2V implicit GEMM wants): 16 B per lane, a
//     brick's 16 consecutive voxels form one 2 KiB run;
//   * a workgroup owns a 4x4x16 voxel brick: its projection into every view is a ~10-px patch that
//     stays L1/L2 resident across the brick's 1024 taps per view (8x reuse);
//   * with B % 8 == 0, sample b is processed only by workgroups dispatched to XCD b % 8, so a sample's
//     feature maps occupy one private L2 instead of being replicated into all eight.
#include <stdlib.h>

#include <type_traits>

#include "lt_common.h"

using namespace lt;

namespace {

// One voxel centre of the (rotated, optionally CMU-permuted) cuboid grid: the loop body of triangulation.py:298-339, fp32 in the
// reference's operation order (bit-exact in eval mode).  Shared by coord_volumes_kernel and the fused unprojection below.
__device__ __forceinline__ void voxel_coord(const float* __restrict__ p, const float* __restrict__ c, const float* __restrict__ R, float step,
                                            int i, int j, int k, int V, int cmu, float& o0, float& o1, float& o2) {
    // triangulation.py:336-339: permute(0,2,1,3) then flip axis 1  ==> out[i][j][k] = cv[i][k][V-1-j]
    const int a0 = i, a1 = cmu ? k : j, a2 = cmu ? (V - 1 - j) : k;
    // f32(position) + f32(step) * idx, two roundings as in triangulation.py:311-313 (no FMA contraction)
    const float x0 = __fadd_rn(p[0], __fmul_rn(step, (float)a0));
    const float x1 = __fadd_rn(p[1], __fmul_rn(step, (float)a1));
    const float x2 = __fadd_rn(p[2], __fmul_rn(step, (float)a2));
    const float d0 = __fsub_rn(x0, c[0]), d1 = __fsub_rn(x1, c[1]), d2 = __fsub_rn(x2, c[2]);
    // rot.mm(d): k-ordered multiply-adds (exact for theta = 0, where R is the identity)
    const float r0 = fmaf(R[2], d2, fmaf(R[1], d1, __fmul_rn(R[0], d0)));
    const float r1 = fmaf(R[5], d2, fmaf(R[4], d1, __fmul_rn(R[3], d0)));
    const float r2 = fmaf(R[8], d2, fmaf(R[7], d1, __fmul_rn(R[6], d0)));
    o0 = __fadd_rn(r0, c[0]);
    o1 = __fadd_rn(r1, c[1]);
    o2 = __fadd_rn(r2, c[2]);
}

__global__ void coord_volumes_kernel(const float* __restrict__ pos, const float* __restrict__ center,
                                     const float* __restrict__ rot, float step, int B, int V, int cmu, float* __restrict__ out) {
    const long long total = (long long)B * V * V * V;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        long long r = g;
        const int k = (int)(r % V); r /= V;
        const int j = (int)(r % V); r /= V;
        const int i = (int)(r % V);
        const int b = (int)(r / V);
        float* o = out + g * 3;
        voxel_coord(pos + 3 * b, center + 3 * b, rot + 9 * b, step, i, j, k, V, cmu, o[0], o[1], o[2]);
    }
}

template <typename T, int CH> struct ChVec;  // CH channels of one pixel <-> CH floats
template <> struct ChVec<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[4]) {
        const float4 v = *(const float4*)p;
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&f)[4]) { *(float4*)p = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct ChVec<float, 1> {
    static __device__ __forceinline__ void ld(const float* p, float (&f)[1]) { f[0] = *p; }
    static __device__ __forceinline__ void st(float* p, const float (&f)[1]) { *p = f[0]; }
};
template <> struct ChVec<bf16_t, 8> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&f)[8]) {
        const uint4 v = *(const uint4*)p;
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(u[i] << 16);
            f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&f)[8]) {
        *(uint4*)p = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
};
template <> struct ChVec<bf16_t, 1> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&f)[1]) { f[0] = bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&f)[1]) { *p = f32_to_bf16(f[0]); }
};

template <int I0, int I1, typename F>
__device__ __forceinline__ void static_for_views(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for_views<I0 + 1, I1>(f);
    }
}

struct UnprojArgs {
    const void* feats;
    const float* proj;
    const float* coords;
    const float* conf;
    void* out;
    int B, NV, C, h, w, v0, v1, v2, agg;
    int bricked;        // 4x4x16 bricks (v0%4 == v1%4 == v2%16 == 0) or linear 256-voxel chunks
    int blocked;        // bricks walked in 32^3-voxel super-blocks (8 x 8 x 2 bricks) instead of in raster order -- see brick_of()
    int chunks;         // workgroups per sample
    int xcd_pin;        // B % 8 == 0
    // lt_unproject_grid_fwd: the voxel centres are COMPUTED (voxel_coord) from the cuboid description instead of read, and written to
    // coords_out when that is not null (the coordinate volume is a returned tensor of the forward: written once, never read back)
    const float* g_pos; const float* g_center; const float* g_rot;
    float g_step; int g_cmu; float* coords_out;
};

// Brick (bi, bj, bk) of chunk c of a sample.  Raster order (k fastest) makes every 4-voxel i-slab of bricks sweep the WHOLE projected cube in every view:
// at 8 views the per-slab working set (2.7 MB of feature rows) plus the output stream does not fit the 4 MB L2 of the XCD the sample is pinned to, and the
// next slab -- which projects 2 pixels next to this one -- finds its rows evicted: PMC traffic 1.33 x the algorithmic bytes at BASELINE config 4 (rounds 4-5,
// unexplained there).  Blocked order (round 6): consecutive chunks fill one 32^3-voxel super-block (8 x 8 x 2 bricks = 128 workgroups, about what one XCD
// runs at a time), whose footprint is ~20 x 20 pixels per view; a pixel row is re-read once per super-block along the view's depth direction instead of
// once per slab.  Per-voxel arithmetic is untouched: results are bit-identical.  Measured at config 4, 16 samples (live PMC, profiles/r06_ab_unproject_brick_order.log):
// 3.48 -> 2.80 GB per launch = 1.33 -> 1.07 x the algorithmic 2.63 GB; the kernel's TIME does not move (3.62-3.67 ms either way: it is issue-bound, not
// traffic-bound); config 2 (4 views: 2.4 MB of maps, they fit): 1.10 -> 1.07 ms.
__device__ __forceinline__ void brick_of(const UnprojArgs& a, int chunk, int& bi, int& bj, int& bk) {
    const int nk = a.v2 >> 4, nj = a.v1 >> 2;
    if (a.blocked) {
        const int g = chunk >> 7, l = chunk & 127;
        const int ngk = nk >> 1, ngj = nj >> 3;
        const int gk = g % ngk, gj = (g / ngk) % ngj, gi = g / (ngk * ngj);
        bk = (gk * 2 + (l & 1)) << 4;
        bj = (gj * 8 + ((l >> 1) & 7)) << 2;
        bi = (gi * 8 + (l >> 4)) << 2;
    } else {
        bk = (chunk % nk) << 4;
        bj = ((chunk / nk) % nj) << 2;
        bi = (chunk / (nk * nj)) << 2;
    }
}

// fp32 (parity) mode keeps IEEE divisions / expf; bf16 (throughput) mode uses v_rcp_f32 / v_exp_f32: the kernel was
// issue-bound (PMC: SQ_WAIT_INST_ANY 40 %, ~50 IEEE divisions + 32 expf per lane-item), not memory-bound.
template <bool FAST> __device__ __forceinline__ float div_(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : __fdiv_rn(a, b); }
template <bool FAST> __device__ __forceinline__ float exp_(float x) { return FAST ? __expf(x) : expf(x); }

// Bilinear sample of CH channels for one voxel in one view; mirrors ATen's CPU grid_sampler_2d
// (bilinear, zeros padding, align_corners=True) and op.py:116-141.
template <typename T, int CH>
__device__ __forceinline__ void sample_view(const T* __restrict__ fmap, const float* __restrict__ P, float X0, float X1,
                                            float X2, int h, int w, int C, int c0, float (&val)[CH]) {
    // multiview.py:105: [X,1] @ P^T, k-ordered
    const float px = __fadd_rn(fmaf(X2, P[2], fmaf(X1, P[1], __fmul_rn(X0, P[0]))), P[3]);
    const float py = __fadd_rn(fmaf(X2, P[6], fmaf(X1, P[5], __fmul_rn(X0, P[4]))), P[7]);
    float pz = __fadd_rn(fmaf(X2, P[10], fmaf(X1, P[9], __fmul_rn(X0, P[8]))), P[11]);
    const bool invalid = pz <= 0.0f;  // op.py:123
    if (pz == 0.0f) pz = 1.0f;        // op.py:125
    constexpr bool FAST = sizeof(T) == 2;
    const float u = div_<FAST>(px, pz), v = div_<FAST>(py, pz);
    // op.py:128-129: x normalised by heatmap_shape[0] (= h), y by heatmap_shape[1] (= w)
    const float gx = __fmul_rn(2.0f, __fsub_rn(div_<FAST>(u, (float)h), 0.5f));
    const float gy = __fmul_rn(2.0f, __fsub_rn(div_<FAST>(v, (float)w), 0.5f));
    // grid_sample, align_corners=True: pixel = (g + 1) * (size - 1) / 2
    const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), 0.5f * (float)(w - 1));
    const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), 0.5f * (float)(h - 1));
    const float xw = floorf(ix), yn = floorf(iy);
    const float we = __fsub_rn(ix, xw), ww = __fsub_rn(1.0f, we);
    const float ws = __fsub_rn(iy, yn), wn = __fsub_rn(1.0f, ws);
#pragma unroll
    for (int e = 0; e < CH; ++e) val[e] = 0.f;
    if (invalid) return;  // op.py:141 (the sample is computed, then zeroed)
    // in-range tests in float (huge / NaN coordinates never convert to int)
    const bool xw_ok = xw >= 0.f && xw <= (float)(w - 1), xe_ok = xw >= -1.f && xw <= (float)(w - 2);
    const bool yn_ok = yn >= 0.f && yn <= (float)(h - 1), ys_ok = yn >= -1.f && yn <= (float)(h - 2);
    if (!((xw_ok || xe_ok) && (yn_ok || ys_ok))) return;
    const int x0 = (int)xw, y0 = (int)yn;
    float t00[CH], t01[CH], t10[CH], t11[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) t00[e] = t01[e] = t10[e] = t11[e] = 0.f;
    const T* base = fmap + ((long long)y0 * w + x0) * C + c0;
    if (yn_ok && xw_ok) ChVec<T, CH>::ld(base, t00);
    if (yn_ok && xe_ok) ChVec<T, CH>::ld(base + C, t01);
    if (ys_ok && xw_ok) ChVec<T, CH>::ld(base + (long long)w * C, t10);
    if (ys_ok && xe_ok) ChVec<T, CH>::ld(base + (long long)w * C + C, t11);
    const float nw = __fmul_rn(wn, ww), ne = __fmul_rn(wn, we), sw = __fmul_rn(ws, ww), se = __fmul_rn(ws, we);
#pragma unroll
    for (int e = 0; e < CH; ++e) val[e] = t00[e] * nw + t01[e] * ne + t10[e] * sw + t11[e] * se;
}

template <typename T, int CH, bool SMALL_NV>
__global__ __launch_bounds__(256) void unproject_kernel(const UnprojArgs a) {
    constexpr bool FAST = sizeof(T) == 2;
    const int tpv = a.C / CH;  // lanes per voxel
    // workgroup -> (sample, chunk); XCD-pinned when B % 8 == 0 (block b runs on XCD b % 8)
    int b, chunk;
    if (a.xcd_pin) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        b = xcd + 8 * (j / a.chunks);
        chunk = j % a.chunks;
    } else {
        b = blockIdx.x / a.chunks;
        chunk = blockIdx.x % a.chunks;
    }
    const long long nvox = (long long)a.v0 * a.v1 * a.v2;
    const T* feats = (const T*)a.feats + (long long)b * a.NV * a.h * a.w * a.C;
    const float* P = a.proj + (long long)b * a.NV * 12;
    const float* coords = a.coords + (long long)b * nvox * 3;
    T* out = (T*)a.out + (long long)b * nvox * a.C;
    int bi = 0, bj = 0, bk = 0;
    if (a.bricked) brick_of(a, chunk, bi, bj, bk);
    const int items = 256 * tpv;
    for (int q = threadIdx.x; q < items; q += 256) {
        const int vb = q / tpv;            // voxel within the chunk, 0..255
        const int c0 = (q - vb * tpv) * CH;
        long long vox;
        if (a.bricked) vox = ((long long)(bi + (vb >> 6)) * a.v1 + bj + ((vb >> 4) & 3)) * a.v2 + bk + (vb & 15);
        else vox = (long long)chunk * 256 + vb;
        if (vox >= nvox) continue;
        const float X0 = coords[vox * 3], X1 = coords[vox * 3 + 1], X2 = coords[vox * 3 + 2];
        float res[CH];
        if (SMALL_NV) {  // NV <= 8: keep every view's sample in registers (two-pass softmax like torch)
            float vals[8][CH];
#pragma unroll
            for (int v = 0; v < 8; ++v)
                if (v < a.NV) sample_view<T, CH>(feats + (long long)v * a.h * a.w * a.C, P + v * 12, X0, X1, X2, a.h, a.w, a.C, c0, vals[v]);
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                float r;
                if (a.agg == LT_AGG_SOFTMAX) {
                    float m = vals[0][e];
#pragma unroll
                    for (int v = 1; v < 8; ++v) if (v < a.NV) m = fmaxf(m, vals[v][e]);
                    // sum_v x_v softmax_v(x) = (sum_v x_v e_v) / (sum_v e_v): one division per channel
                    float s = 0.f, tt = 0.f;
#pragma unroll
                    for (int v = 0; v < 8; ++v) if (v < a.NV) { const float ex = exp_<FAST>(vals[v][e] - m); s += ex; tt += vals[v][e] * ex; }
                    r = div_<FAST>(tt, s);
                } else if (a.agg == LT_AGG_MAX) {
                    r = vals[0][e];
#pragma unroll
                    for (int v = 1; v < 8; ++v) if (v < a.NV) r = fmaxf(r, vals[v][e]);
                } else if (a.agg == LT_AGG_CONF || a.agg == LT_AGG_CONF_NORM) {
                    float cs = 1.f;
                    if (a.agg == LT_AGG_CONF_NORM) {
                        cs = 0.f;
#pragma unroll
                        for (int v = 0; v < 8; ++v) if (v < a.NV) cs += a.conf[((long long)b * a.NV + v) * a.C + c0 + e];
                    }
                    r = 0.f;
#pragma unroll
                    for (int v = 0; v < 8; ++v) if (v < a.NV) r += vals[v][e] * __fdiv_rn(a.conf[((long long)b * a.NV + v) * a.C + c0 + e], cs);
                } else {
                    r = 0.f;
#pragma unroll
                    for (int v = 0; v < 8; ++v) if (v < a.NV) r += vals[v][e];
                }
                res[e] = r;
            }
        } else {  // any NV: recompute the samples instead of storing them (softmax needs the max first)
            float m[CH], s[CH], acc[CH], val[CH], cs[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                m[e] = -INFINITY; s[e] = 0.f; acc[e] = 0.f; cs[e] = 1.f;
                if (a.agg == LT_AGG_CONF_NORM) {
                    cs[e] = 0.f;
                    for (int v = 0; v < a.NV; ++v) cs[e] += a.conf[((long long)b * a.NV + v) * a.C + c0 + e];
                }
            }
            if (a.agg == LT_AGG_SOFTMAX || a.agg == LT_AGG_MAX)
                for (int v = 0; v < a.NV; ++v) {
                    sample_view<T, CH>(feats + (long long)v * a.h * a.w * a.C, P + v * 12, X0, X1, X2, a.h, a.w, a.C, c0, val);
#pragma unroll
                    for (int e = 0; e < CH; ++e) m[e] = fmaxf(m[e], val[e]);
                }
            if (a.agg != LT_AGG_MAX)
                for (int v = 0; v < a.NV; ++v) {
                    sample_view<T, CH>(feats + (long long)v * a.h * a.w * a.C, P + v * 12, X0, X1, X2, a.h, a.w, a.C, c0, val);
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        if (a.agg == LT_AGG_SOFTMAX) { const float ex = exp_<FAST>(val[e] - m[e]); s[e] += ex; acc[e] += val[e] * ex; }
                        else if (a.agg == LT_AGG_CONF || a.agg == LT_AGG_CONF_NORM)
                            acc[e] += val[e] * __fdiv_rn(a.conf[((long long)b * a.NV + v) * a.C + c0 + e], cs[e]);
                        else acc[e] += val[e];
                    }
                }
#pragma unroll
            for (int e = 0; e < CH; ++e)
                res[e] = a.agg == LT_AGG_MAX ? m[e] : (a.agg == LT_AGG_SOFTMAX ? __fdiv_rn(acc[e], s[e]) : acc[e]);
        }
        ChVec<T, CH>::st(out + vox * a.C + c0, res);
    }
}

// (unproject_lds_kernel, an LDS-staged variant of the gather -- opt-in with LT_UNPROJ_LDS=1 -- measured 0.90 ms against 0.57 ms for the
// gathering kernel at the BASELINE shape and was removed in round 2: the 37 MB of feature maps are L2-resident and the gather keeps 16
// independent loads in flight per lane, while the staged copy exposes one load round trip per workgroup between its two barriers.)

// ---- quad kernel: bf16, C = 32, 4 or 8 views, softmax aggregation, bricked volumes (BASELINE configurations 2 and 4) ------------------
// The generic kernel above is VALU-bound (~700 instructions per lane-item at 4 views; PMC: 40 % of the cycles waiting on instruction
// issue): every one of the four lanes of a voxel (one per 8-channel vector) redoes the projection of the voxel into all views.
// Here the four lanes of a voxel's quad split the VIEWS for the projection (lane j projects into views j, j+4, ... and keeps those
// 3x4 matrices in registers for the whole kernel), fold the zero-padding rules into the four bilinear weights (an invalid corner
// gets weight 0 and a clamped, readable address), and hand the four tap offsets + four weights of their views to the other three
// lanes with DPP quad broadcasts; the bilinear blend and the view softmax run on packed fp32 pairs with EXPLICIT fused
// multiply-adds (the library is built with -ffp-contract=off for the bit-exact coordinate grid, which left this kernel with 109
// v_pk_mul + 102 v_pk_add per voxel where 125 v_pk_fma do).  NVL = views per lane (1: 4 views, 2: 8 views).  GRID: the voxel centre
// is computed in registers from the cuboid description (bit-identical to coord_volumes_kernel) instead of read from the 12-byte-
// per-voxel coordinate volume, and lanes 0-2 of the quad write its three components to the returned tensor on the way.
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int Q>
__device__ __forceinline__ int quad_bcast_i(int v) {   // value of lane Q of this lane's quad
    return __builtin_amdgcn_mov_dpp(v, Q | (Q << 2) | (Q << 4) | (Q << 6), 0xf, 0xf, true);
}
template <int Q>
__device__ __forceinline__ float quad_bcast_f(float v) { return __int_as_float(quad_bcast_i<Q>(__float_as_int(v))); }
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }

// (round 6, measured and not kept: the 8-view instantiation takes 152 VGPRs = 3 waves per SIMD; capped at 128 = 4 waves with __launch_bounds__(256, 4) --
//  4 registers of loop invariants in scratch -- it ran 3.72-3.75 ms against 3.62-3.67 ms at BASELINE config 4, and with MORE memory traffic (3.09 vs 2.80 GB):
//  the kernel is bound by VALU issue + the gather path together (valu_issue_frac 0.44-0.45), and a fourth wave adds neither.)
template <int NVL, bool GRID>
__global__ __launch_bounds__(256) void unproject_qn_kernel(const UnprojArgs a) {
    typedef bf16_t T;
    constexpr int C = 32, NV = 4 * NVL;
    int b, chunk;
    if (a.xcd_pin) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        b = xcd + 8 * (j / a.chunks);
        chunk = j % a.chunks;
    } else {
        b = blockIdx.x / a.chunks;
        chunk = blockIdx.x % a.chunks;
    }
    const long long nvox = (long long)a.v0 * a.v1 * a.v2;
    const T* feats = (const T*)a.feats + (long long)b * NV * a.h * a.w * C;
    const float* coords = GRID ? nullptr : a.coords + (long long)b * nvox * 3;
    float* coords_out = (GRID && a.coords_out) ? a.coords_out + (long long)b * nvox * 3 : nullptr;
    T* out = (T*)a.out + (long long)b * nvox * C;
    int bi, bj, bk;
    brick_of(a, chunk, bi, bj, bk);
    const int t = threadIdx.x, qv = t & 3;               // qv: the views this lane projects into (qv, qv + 4) AND its 8-channel vector
    const int h = a.h, w = a.w;
    float P[NVL][12];
#pragma unroll
    for (int s = 0; s < NVL; ++s)
#pragma unroll
        for (int i = 0; i < 12; ++i) P[s][i] = a.proj[((long long)b * NV + qv + 4 * s) * 12 + i];
    float gp[3], gc[3], gR[9];
    if (GRID) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { gp[i] = a.g_pos[3 * b + i]; gc[i] = a.g_center[3 * b + i]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) gR[i] = a.g_rot[9 * b + i];
    }
    const float inv_h = __builtin_amdgcn_rcpf((float)h), inv_w = __builtin_amdgcn_rcpf((float)w);

#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        const int vb = it * 64 + (t >> 2);
        const int vi = bi + (vb >> 6), vj = bj + ((vb >> 4) & 3), vk = bk + (vb & 15);
        const long long vox = ((long long)vi * a.v1 + vj) * a.v2 + vk;
        float X0, X1, X2;
        if (GRID) {
            voxel_coord(gp, gc, gR, a.g_step, vi, vj, vk, a.v0, a.g_cmu, X0, X1, X2);
            if (coords_out && qv < 3) coords_out[vox * 3 + qv] = qv == 0 ? X0 : (qv == 1 ? X1 : X2);
        } else {
            X0 = coords[vox * 3]; X1 = coords[vox * 3 + 1]; X2 = coords[vox * 3 + 2];
        }
        int o00[NVL], o01[NVL], o10[NVL], o11[NVL];
        float k00[NVL], k01[NVL], k10[NVL], k11[NVL];
#pragma unroll
        for (int s = 0; s < NVL; ++s) {
            // ---- projection of the voxel into view qv + 4 s (the arithmetic of sample_view<bf16>) ----
            const float* Ps = P[s];
            const float px = __fadd_rn(fmaf(X2, Ps[2], fmaf(X1, Ps[1], __fmul_rn(X0, Ps[0]))), Ps[3]);
            const float py = __fadd_rn(fmaf(X2, Ps[6], fmaf(X1, Ps[5], __fmul_rn(X0, Ps[4]))), Ps[7]);
            float pz = __fadd_rn(fmaf(X2, Ps[10], fmaf(X1, Ps[9], __fmul_rn(X0, Ps[8]))), Ps[11]);
            const bool invalid = pz <= 0.0f;
            if (pz == 0.0f) pz = 1.0f;
            const float rz = __builtin_amdgcn_rcpf(pz);
            const float u = px * rz, vv = py * rz;
            const float gx = __fmul_rn(2.0f, __fsub_rn(u * inv_h, 0.5f));      // op.py:128-129: x by h, y by w
            const float gy = __fmul_rn(2.0f, __fsub_rn(vv * inv_w, 0.5f));
            const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), 0.5f * (float)(w - 1));
            const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), 0.5f * (float)(h - 1));
            const float xw = floorf(ix), yn = floorf(iy);
            const float we = __fsub_rn(ix, xw), ww = __fsub_rn(1.0f, we);
            const float ws = __fsub_rn(iy, yn), wn = __fsub_rn(1.0f, ws);
            const bool xw_ok = xw >= 0.f && xw <= (float)(w - 1), xe_ok = xw >= -1.f && xw <= (float)(w - 2);
            const bool yn_ok = yn >= 0.f && yn <= (float)(h - 1), ys_ok = yn >= -1.f && yn <= (float)(h - 2);
            const bool act = !invalid && (xw_ok || xe_ok) && (yn_ok || ys_ok);
            const int x0 = act ? (int)xw : 0, y0 = act ? (int)yn : 0;
            const int xwc = min(max(x0, 0), w - 1), xec = min(max(x0 + 1, 0), w - 1);
            const int rn = min(max(y0, 0), h - 1) * w, rs = min(max(y0 + 1, 0), h - 1) * w;
            // this view's four corners: element offsets (clamped into the map) and weights (0 where the corner is padding)
            const int vbase = (qv + 4 * s) * h * w * C;
            o00[s] = vbase + (rn + xwc) * C; o01[s] = vbase + (rn + xec) * C; o10[s] = vbase + (rs + xwc) * C; o11[s] = vbase + (rs + xec) * C;
            k00[s] = (act && yn_ok && xw_ok) ? __fmul_rn(wn, ww) : 0.f; k01[s] = (act && yn_ok && xe_ok) ? __fmul_rn(wn, we) : 0.f;
            k10[s] = (act && ys_ok && xw_ok) ? __fmul_rn(ws, ww) : 0.f; k11[s] = (act && ys_ok && xe_ok) ? __fmul_rn(ws, we) : 0.f;
        }

        f32x2_t val[NV][4];                               // [view][channel pair] of this lane's 8 channels
        auto view = [&](auto vc) {
            constexpr int V = decltype(vc)::value, Q = V & 3, S = V >> 2;
            const int p00 = quad_bcast_i<Q>(o00[S]), p01 = quad_bcast_i<Q>(o01[S]), p10 = quad_bcast_i<Q>(o10[S]), p11 = quad_bcast_i<Q>(o11[S]);
            const float w00 = quad_bcast_f<Q>(k00[S]), w01 = quad_bcast_f<Q>(k01[S]), w10 = quad_bcast_f<Q>(k10[S]), w11 = quad_bcast_f<Q>(k11[S]);
            const uint4 t00 = *(const uint4*)(feats + p00 + qv * 8), t01 = *(const uint4*)(feats + p01 + qv * 8);
            const uint4 t10 = *(const uint4*)(feats + p10 + qv * 8), t11 = *(const uint4*)(feats + p11 + qv * 8);
            const unsigned u00[4] = {t00.x, t00.y, t00.z, t00.w}, u01[4] = {t01.x, t01.y, t01.z, t01.w};
            const unsigned u10[4] = {t10.x, t10.y, t10.z, t10.w}, u11[4] = {t11.x, t11.y, t11.z, t11.w};
            const f32x2_t W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const f32x2_t f00 = {__uint_as_float(u00[d] << 16), __uint_as_float(u00[d] & 0xffff0000u)};
                const f32x2_t f01 = {__uint_as_float(u01[d] << 16), __uint_as_float(u01[d] & 0xffff0000u)};
                const f32x2_t f10 = {__uint_as_float(u10[d] << 16), __uint_as_float(u10[d] & 0xffff0000u)};
                const f32x2_t f11 = {__uint_as_float(u11[d] << 16), __uint_as_float(u11[d] & 0xffff0000u)};
                val[V][d] = pk_fma(f11, W11, pk_fma(f10, W10, pk_fma(f01, W01, f00 * W00)));
            }
        };
        static_for_views<0, NV>(view);

        // ---- softmax over the views per channel: sum_v x_v softmax_v(x) = (sum_v x_v e_v) / (sum_v e_v) ----
        unsigned o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f32x2_t m = val[0][d];
#pragma unroll
            for (int v = 1; v < NV; ++v) { m[0] = fmaxf(m[0], val[v][d][0]); m[1] = fmaxf(m[1], val[v][d][1]); }
            f32x2_t s = {0.f, 0.f}, tt = {0.f, 0.f};
            const f32x2_t L2E = {1.4426950408889634f, 1.4426950408889634f}, mL = m * L2E;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const f32x2_t dl = pk_fma(val[v][d], L2E, -mL);                 // (x - m) log2(e)
                const f32x2_t ex = {__builtin_amdgcn_exp2f(dl[0]), __builtin_amdgcn_exp2f(dl[1])};
                s += ex;
                tt = pk_fma(val[v][d], ex, tt);
            }
            o[d] = pack_bf16x2(tt[0] * __builtin_amdgcn_rcpf(s[0]), tt[1] * __builtin_amdgcn_rcpf(s[1]));
        }
        *(uint4*)(out + vox * C + qv * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

template <typename T, int CH>
int launch_unproject(const UnprojArgs& a, hipStream_t st) {
    const unsigned grid = (unsigned)((long long)a.B * a.chunks);
    if (a.NV <= 8) hipLaunchKernelGGL((unproject_kernel<T, CH, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((unproject_kernel<T, CH, false>), dim3(grid), dim3(256), 0, st, a);
    LT_CHECK_LAUNCH("lt_unproject_fwd");
    return LT_OK;
}

}  // namespace

extern "C" int lt_coord_volumes(const float* pos, const float* center, const float* rot, float step, int32_t B, int32_t V,
                                int32_t cmu_transfer, float* coords, void* stream) {
    LT_REQUIRE(pos && center && rot && coords && B >= 1 && V >= 2, LT_ERR_INVALID, "lt_coord_volumes: bad argument");
    const long long total = (long long)B * V * V * V;
    const long long blocks = cdiv(total, 256);
    hipLaunchKernelGGL(coord_volumes_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, pos,
                       center, rot, step, B, V, cmu_transfer, coords);
    LT_CHECK_LAUNCH("lt_coord_volumes");
    return LT_OK;
}

namespace {
// fills the launch geometry and picks the kernel; a.coords (read) or the a.g_* grid description must be set by the caller
int unproject_dispatch(UnprojArgs& a, int dtype, bool grid, hipStream_t st) {
    const long long nvox = (long long)a.v0 * a.v1 * a.v2;
    a.bricked = (a.v0 % 4 == 0 && a.v1 % 4 == 0 && a.v2 % 16 == 0) ? 1 : 0;
    a.blocked = (a.bricked && a.v0 % 32 == 0 && a.v1 % 32 == 0 && a.v2 % 32 == 0 && !getenv("LT_UNPROJ_RASTER")) ? 1 : 0;          // LT_UNPROJ_RASTER=1: the old order (A/B)
    a.chunks = (int)cdiv(nvox, 256);
    a.xcd_pin = (a.B % 8 == 0) ? 1 : 0;
    LT_REQUIRE((long long)a.B * a.chunks < (1ll << 31), LT_ERR_UNSUPPORTED, "lt_unproject_fwd: grid too large");
    const unsigned nblk = (unsigned)((long long)a.B * a.chunks);
    // quad kernel: bf16, 32 channels, 4 or 8 views, view softmax, bricked volume (BASELINE configurations 2 and 4)
    const char* no_q4 = getenv("LT_UNPROJ_NO_Q4");       // A/B, read per call
    const bool quad = dtype == LT_BF16 && a.C == 32 && (a.NV == 4 || a.NV == 8) && a.bricked && a.agg == LT_AGG_SOFTMAX && !no_q4 &&
                      (long long)a.NV * a.h * a.w * a.C < (1ll << 30);
    if (quad) {
        if (a.NV == 4) {
            if (grid) hipLaunchKernelGGL((unproject_qn_kernel<1, true>), dim3(nblk), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((unproject_qn_kernel<1, false>), dim3(nblk), dim3(256), 0, st, a);
        } else {
            if (grid) hipLaunchKernelGGL((unproject_qn_kernel<2, true>), dim3(nblk), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((unproject_qn_kernel<2, false>), dim3(nblk), dim3(256), 0, st, a);
        }
        LT_CHECK_LAUNCH("lt_unproject_fwd(quad)");
        return LT_OK;
    }
    if (grid) {      // no fused kernel for this configuration: materialise the grid (it is a returned tensor anyway), then gather
        const long long total = (long long)a.B * nvox;
        const long long blocks = cdiv(total, 256);
        hipLaunchKernelGGL(coord_volumes_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, a.g_pos, a.g_center, a.g_rot,
                           a.g_step, a.B, a.v0, a.g_cmu, a.coords_out);
        LT_CHECK_LAUNCH("lt_unproject_grid_fwd(coords)");
        a.coords = a.coords_out;
    }
    if (dtype == LT_F32) {
        if (a.C % 4 == 0) return launch_unproject<float, 4>(a, st);
        return launch_unproject<float, 1>(a, st);
    }
    if (a.C % 8 == 0) return launch_unproject<bf16_t, 8>(a, st);
    return launch_unproject<bf16_t, 1>(a, st);
}
}  // namespace

extern "C" int lt_unproject_fwd(int32_t dtype, const void* feats, const float* proj, const float* coords, const float* conf,
                                void* out, int32_t B, int32_t NV, int32_t C, int32_t h, int32_t w, int32_t v0, int32_t v1,
                                int32_t v2, int32_t agg, void* stream) {
    LT_REQUIRE(feats && proj && coords && out, LT_ERR_INVALID, "lt_unproject_fwd: null argument");
    LT_REQUIRE(dtype == LT_F32 || dtype == LT_BF16, LT_ERR_INVALID, "lt_unproject_fwd: bad dtype %d", dtype);
    LT_REQUIRE(agg >= LT_AGG_SUM && agg <= LT_AGG_CONF_NORM, LT_ERR_INVALID, "lt_unproject_fwd: unknown aggregation %d", agg);
    LT_REQUIRE((agg != LT_AGG_CONF && agg != LT_AGG_CONF_NORM) || conf, LT_ERR_INVALID, "lt_unproject_fwd: LT_AGG_CONF* needs confidences");
    LT_REQUIRE(B >= 1 && NV >= 1 && C >= 1 && h >= 2 && w >= 2 && v0 >= 1 && v1 >= 1 && v2 >= 1, LT_ERR_INVALID, "lt_unproject_fwd: bad shape");
    LT_REQUIRE((long long)NV * h * w * C < (1ll << 31), LT_ERR_UNSUPPORTED, "lt_unproject_fwd: feature maps too large");
    UnprojArgs a = {};
    a.feats = feats; a.proj = proj; a.coords = coords; a.conf = conf; a.out = out;
    a.B = B; a.NV = NV; a.C = C; a.h = h; a.w = w; a.v0 = v0; a.v1 = v1; a.v2 = v2; a.agg = agg;
    return unproject_dispatch(a, dtype, false, (hipStream_t)stream);
}

extern "C" int lt_unproject_grid_fwd(int32_t dtype, const void* feats, const float* proj, const float* pos, const float* center, const float* rot,
                                     float step, int32_t cmu_transfer, float* coords_out, const float* conf, void* out, int32_t B, int32_t NV,
                                     int32_t C, int32_t h, int32_t w, int32_t V, int32_t agg, void* stream) {
    LT_REQUIRE(feats && proj && pos && center && rot && coords_out && out, LT_ERR_INVALID, "lt_unproject_grid_fwd: null argument");
    LT_REQUIRE(dtype == LT_F32 || dtype == LT_BF16, LT_ERR_INVALID, "lt_unproject_grid_fwd: bad dtype %d", dtype);
    LT_REQUIRE(agg >= LT_AGG_SUM && agg <= LT_AGG_CONF_NORM, LT_ERR_INVALID, "lt_unproject_grid_fwd: unknown aggregation %d", agg);
    LT_REQUIRE((agg != LT_AGG_CONF && agg != LT_AGG_CONF_NORM) || conf, LT_ERR_INVALID, "lt_unproject_grid_fwd: LT_AGG_CONF* needs confidences");
    LT_REQUIRE(B >= 1 && NV >= 1 && C >= 1 && h >= 2 && w >= 2 && V >= 2, LT_ERR_INVALID, "lt_unproject_grid_fwd: bad shape");
    LT_REQUIRE((long long)NV * h * w * C < (1ll << 31), LT_ERR_UNSUPPORTED, "lt_unproject_grid_fwd: feature maps too large");
    UnprojArgs a = {};
    a.feats = feats; a.proj = proj; a.coords = nullptr; a.conf = conf; a.out = out;
    a.B = B; a.NV = NV; a.C = C; a.h = h; a.w = w; a.v0 = V; a.v1 = V; a.v2 = V; a.agg = agg;
    a.g_pos = pos; a.g_center = center; a.g_rot = rot; a.g_step = step; a.g_cmu = cmu_transfer; a.coords_out = coords_out;
    return unproject_dispatch(a, dtype, true, (hipStream_t)stream);
}

namespace {
__global__ void rotate_points_kernel(const float* __restrict__ x, const float* __restrict__ R, float* __restrict__ y, long long n) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const float a = x[g * 3], b = x[g * 3 + 1], c = x[g * 3 + 2];
        y[g * 3] = fmaf(R[2], c, fmaf(R[1], b, __fmul_rn(R[0], a)));
        y[g * 3 + 1] = fmaf(R[5], c, fmaf(R[4], b, __fmul_rn(R[3], a)));
        y[g * 3 + 2] = fmaf(R[8], c, fmaf(R[7], b, __fmul_rn(R[6], a)));
    }
}
}  // namespace

extern "C" int lt_rotate_points(const float* x, const float* rot, float* y, int64_t n, void* stream) {
    LT_REQUIRE(x && rot && y && n >= 1, LT_ERR_INVALID, "lt_rotate_points: bad argument");
    const long long blocks = cdiv(n, 256);
    hipLaunchKernelGGL(rotate_points_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, rot, y, (long long)n);
    LT_CHECK_LAUNCH("lt_rotate_points");
    return LT_OK;
}
