// Backward of the unprojection (what torch.autograd derives for the reference from op.unproject_heatmaps, mvn/utils/op.py:99-166):
//   no gradient flows to the grid / projection / coordinates (none requires grad there); grid_sample's backward spreads wt_tap * g over
//   the four taps of feat[b, v, c]; a sample with depth <= 0 is zeroed IN PLACE after sampling (op.py:141), so it passes no gradient
//   back but its 0 still enters the view softmax; softmax aggregation out = sum_v w_v x_v, w = softmax_v(x):
//   d out / d x_v = w_v (1 + x_v - out); 'conf': d out / d x_v = c_v, d out / d c_v = x_v (summed over the voxels).
//
// Round 3: a GATHER, deterministic, no atomics at all (the round-2 scatter issued NV x 4 taps x C global float atomics per voxel: 134 M
// per sample at config 2, 1.8 ms per sample, and was not bitwise repeatable).  Three launches over a workspace:
//   K0  unproj_taps_kernel   one wave per 4x4x4 voxel brick: every voxel projected into every view ONCE: its tap record (the 2x2 patch's
//                            origin pixel and the four bilinear weights, 0 = padding / depth <= 0) -> txy / tw[b][v][voxel], and the
//                            pixel bounding box of the brick's weighted taps -> bbox[b][v][brick] (empty bricks: x0 > x1)
//   K1  unproj_dx_kernel     one lane per (voxel, 4-channel vector), the forward's sampling redone: d out / d x_v times the upstream
//                            gradient -> dxs[b][v][voxel][C] fp32 (the per-view gradient of the SAMPLED value); the confidence gradient
//                            as per-workgroup partial sums, added in a fixed order by unproj_gconf_finalize_kernel
//   K2  unproj_gather_kernel one workgroup per (16 x 16 pixel tile, view, sample), the tile's gradient in LDS: walks the bricks whose
//                            bbox meets the tile in index order (next brick's tap records and dx values in flight), and adds every tap
//                            that falls INSIDE the tile in (brick, voxel) order -- each wave owns a quarter of the channels, each
//                            half-wave half of the tile's pixel columns, lane = pixel parity class x channel (the four taps of a
//                            voxel have four different parities, and a cell is only ever updated by the lane of its parity); plain
//                            LDS read-add-write in rounds of four hits, the sums forwarded in registers when hits of a round meet in
//                            one cell.  The tile is
//                            stored once with plain stores (no pre-zeroed output: every pixel of the map belongs to exactly one tile).
// A tap is processed by exactly one workgroup (the one whose tile holds its pixel), every cell's additions have a fixed order: results
// are bitwise repeatable.  Measured on the way (B = 8, config-2 shape, profiles/r03_unproject_bwd.md): the same gather with ds_add_f32
// (LDS float atomics) took 9.0 ms, 3.1 ms with the atomic replaced by a plain store -- LDS float atomics run at a small fraction of the
// LDS rate -- and 0.8 ms without the hit loop.  The round-2 scatter kernel stays reachable (LT_UNPROJ_BWD_ATOMICS=1) as the A/B
// reference and for configurations the gather does not take (C > 64 or not a power of two).
#include <stdlib.h>

#include "lt_common.h"

using namespace lt;

namespace {

struct UnprojBwdArgs {
    const void* feats;       // (B, NV, h, w, C) T
    const float* proj;       // (B, NV, 3, 4)
    const float* coords;     // (B, nvox, 3)
    const float* conf;       // (B, NV, C) or null
    const float* gout;       // (B, nvox, C) fp32: dL/d volume, channels-last
    float* gfeats;           // (B, NV, h, w, C) fp32
    float* gconf;            // (B, NV, C) fp32, or null
    float* dxs;              // workspace: (B, NV, nvox, C) fp32
    int* bbox;               // workspace: (B, NV, nbricks, 4) = x0, x1, y0, y1 of the taps with weight (x0 > x1: none)
    int* txy;                // workspace: (B, NV, nvox) origin pixel of the 2x2 tap patch, (x0 + 1) | (y0 + 1) << 16  (x0, y0 >= -1)
    float4* tw;              // workspace: (B, NV, nvox) weights of the taps (x0,y0) (x0+1,y0) (x0,y0+1) (x0+1,y0+1); 0 = padding / masked
    double* gcpart;          // workspace: (B, nblk1, NV, C) partial confidence gradients
    int B, NV, C, h, w, agg;
    int v0, v1, v2, nb0, nb1, nb2;
    long long nvox;
};

struct Tap {                 // one view's projection of one voxel: four tap pixels (x, y; clamped into the map) and weights; weight 0 = padding / masked
    int x[4], y[4];
    float k[4];
    int x0, y0;              // unclamped origin of the 2x2 patch (>= -1; 0 for an inactive voxel, whose weights are all 0)
};

__device__ __forceinline__ Tap project_taps(const float* __restrict__ P, float X0, float X1, float X2, int h, int w) {
    // the arithmetic of sample_view<float> in unproject.hip (IEEE divisions: this is the fp32 path)
    const float px = __fadd_rn(fmaf(X2, P[2], fmaf(X1, P[1], __fmul_rn(X0, P[0]))), P[3]);
    const float py = __fadd_rn(fmaf(X2, P[6], fmaf(X1, P[5], __fmul_rn(X0, P[4]))), P[7]);
    float pz = __fadd_rn(fmaf(X2, P[10], fmaf(X1, P[9], __fmul_rn(X0, P[8]))), P[11]);
    const bool invalid = pz <= 0.0f;
    if (pz == 0.0f) pz = 1.0f;
    const float u = __fdiv_rn(px, pz), v = __fdiv_rn(py, pz);
    const float gx = __fmul_rn(2.0f, __fsub_rn(__fdiv_rn(u, (float)h), 0.5f));
    const float gy = __fmul_rn(2.0f, __fsub_rn(__fdiv_rn(v, (float)w), 0.5f));
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
    const int ync = min(max(y0, 0), h - 1), ysc = min(max(y0 + 1, 0), h - 1);
    Tap t;
    t.x0 = x0; t.y0 = y0;
    t.x[0] = xwc; t.x[1] = xec; t.x[2] = xwc; t.x[3] = xec;
    t.y[0] = ync; t.y[1] = ync; t.y[2] = ysc; t.y[3] = ysc;
    t.k[0] = (act && yn_ok && xw_ok) ? __fmul_rn(wn, ww) : 0.f;
    t.k[1] = (act && yn_ok && xe_ok) ? __fmul_rn(wn, we) : 0.f;
    t.k[2] = (act && ys_ok && xw_ok) ? __fmul_rn(ws, ww) : 0.f;
    t.k[3] = (act && ys_ok && xe_ok) ? __fmul_rn(ws, we) : 0.f;
    return t;
}

template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&f)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&f)[4]) {
    const float4 v = *(const float4*)p;
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&f)[4]) {
    const uint2 v = *(const uint2*)p;
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}

constexpr int UB_MAXV = 8;   // views kept in registers

// sampled values x[v][4] of one voxel / channel vector in every view (the forward's bilinear sampling with its padding / depth rules)
template <typename T, int MV>
__device__ __forceinline__ void sample_views(const UnprojBwdArgs& a, const T* feats, const float* P, float X0, float X1, float X2, int c0,
                                             Tap (&tp)[MV], float (&x)[MV][4]) {
    const long long hw = (long long)a.h * a.w;
#pragma unroll
    for (int v = 0; v < MV; ++v) {
        if (v < a.NV) {
            tp[v] = project_taps(P + v * 12, X0, X1, X2, a.h, a.w);
            const T* fm = feats + v * hw * a.C + c0;
            float t0[4], t1[4], t2[4], t3[4];
            ld4<T>(fm + (long long)(tp[v].y[0] * a.w + tp[v].x[0]) * a.C, t0); ld4<T>(fm + (long long)(tp[v].y[1] * a.w + tp[v].x[1]) * a.C, t1);
            ld4<T>(fm + (long long)(tp[v].y[2] * a.w + tp[v].x[2]) * a.C, t2); ld4<T>(fm + (long long)(tp[v].y[3] * a.w + tp[v].x[3]) * a.C, t3);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[v][e] = t0[e] * tp[v].k[0] + t1[e] * tp[v].k[1] + t2[e] * tp[v].k[2] + t3[e] * tp[v].k[3];
        }
    }
}

// d out / d x_v times the upstream gradient, per channel of the vector; gc[v][e] = g * x_v (the confidence gradient's summand)
template <int MV>
__device__ __forceinline__ void view_gradients(const UnprojBwdArgs& a, int b, int c0, const float (&g)[4], const float (&x)[MV][4],
                                               float (&dx)[MV][4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (a.agg == LT_AGG_SOFTMAX) {
            float m = x[0][e];
#pragma unroll
            for (int v = 1; v < MV; ++v) if (v < a.NV) m = fmaxf(m, x[v][e]);
            float s = 0.f, tt = 0.f, ex[MV];
#pragma unroll
            for (int v = 0; v < MV; ++v) if (v < a.NV) { ex[v] = expf(x[v][e] - m); s += ex[v]; tt += x[v][e] * ex[v]; }
            const float out = __fdiv_rn(tt, s);
#pragma unroll
            for (int v = 0; v < MV; ++v) if (v < a.NV) dx[v][e] = g[e] * __fdiv_rn(ex[v], s) * (1.0f + x[v][e] - out);
        } else if (a.agg == LT_AGG_MAX) {               // torch.max(dim): gradient to the FIRST maximal view
            int am = 0;
            float best = x[0][e];
#pragma unroll
            for (int v = 1; v < MV; ++v) if (v < a.NV && x[v][e] > best) { best = x[v][e]; am = v; }
#pragma unroll
            for (int v = 0; v < MV; ++v) if (v < a.NV) dx[v][e] = v == am ? g[e] : 0.f;
        } else if (a.agg == LT_AGG_CONF || a.agg == LT_AGG_CONF_NORM) {
            float cs = 1.f;
            if (a.agg == LT_AGG_CONF_NORM) {            // the forward normalises the confidences over the views (triangulation.py:268-269)
                cs = 0.f;
#pragma unroll
                for (int v = 0; v < MV; ++v) if (v < a.NV) cs += a.conf[((long long)b * a.NV + v) * a.C + c0 + e];
            }
#pragma unroll
            for (int v = 0; v < MV; ++v)
                if (v < a.NV) {
                    const float cv = a.conf[((long long)b * a.NV + v) * a.C + c0 + e];
                    dx[v][e] = g[e] * (a.agg == LT_AGG_CONF_NORM ? __fdiv_rn(cv, cs) : cv);
                }
        } else {
#pragma unroll
            for (int v = 0; v < MV; ++v) if (v < a.NV) dx[v][e] = g[e];
        }
    }
}

// ---- round-2 scatter (LT_UNPROJ_BWD_ATOMICS=1; gfeats / gconf zeroed by the host entry) ------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void unproject_bwd_kernel(const UnprojBwdArgs a) {
    const int tpv = a.C >> 2;                              // lanes per voxel (4 channels each)
    const long long items = a.nvox * tpv;
    const int b = blockIdx.y;
    const T* feats = (const T*)a.feats + (long long)b * a.NV * a.h * a.w * a.C;
    float* gfeats = a.gfeats + (long long)b * a.NV * a.h * a.w * a.C;
    const float* P = a.proj + (long long)b * a.NV * 12;
    const float* coords = a.coords + (long long)b * a.nvox * 3;
    const float* gout = a.gout + (long long)b * a.nvox * a.C;
    const long long hw = (long long)a.h * a.w;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
        const long long vox = q / tpv;
        const int c0 = (int)(q - vox * tpv) * 4;
        const float X0 = coords[vox * 3], X1 = coords[vox * 3 + 1], X2 = coords[vox * 3 + 2];
        float g[4];
        ld4<float>(gout + vox * a.C + c0, g);
        Tap tp[UB_MAXV];
        float x[UB_MAXV][4], dx[UB_MAXV][4];
        sample_views<T, UB_MAXV>(a, feats, P, X0, X1, X2, c0, tp, x);
        view_gradients<UB_MAXV>(a, b, c0, g, x, dx);
        if (a.gconf && a.agg == LT_AGG_CONF) {
#pragma unroll
            for (int v = 0; v < UB_MAXV; ++v)
                if (v < a.NV)
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(a.gconf + ((long long)b * a.NV + v) * a.C + c0 + e, g[e] * x[v][e]);
        }
#pragma unroll
        for (int v = 0; v < UB_MAXV; ++v) {
            if (v < a.NV) {
                float* gm = gfeats + v * hw * a.C + c0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float wk = tp[v].k[k];
                    if (wk != 0.f) {
                        float* dst = gm + (long long)(tp[v].y[k] * a.w + tp[v].x[k]) * a.C;
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(dst + e, wk * dx[v][e]);
                    }
                }
            }
        }
    }
}

// ---- gather: bricks ---------------------------------------------------------------------------------------------------------------------
// voxel of (brick, lane): bricks are 4 x 4 x 4 voxels of the (v0, v1, v2) grid; -1 past the grid's end
__device__ __forceinline__ long long brick_voxel(const UnprojBwdArgs& a, int brick, int lane) {
    const int b2 = brick % a.nb2, b1 = (brick / a.nb2) % a.nb1, b0 = brick / (a.nb2 * a.nb1);
    const int i = b0 * 4 + (lane >> 4), j = b1 * 4 + ((lane >> 2) & 3), k = b2 * 4 + (lane & 3);
    if (i >= a.v0 || j >= a.v1 || k >= a.v2) return -1;
    return ((long long)i * a.v1 + j) * a.v2 + k;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// K0: grid (ceil(nbricks / 4), B), 4 waves, one brick per wave
__global__ __launch_bounds__(256) void unproj_taps_kernel(const UnprojBwdArgs a) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int nbricks = a.nb0 * a.nb1 * a.nb2;
    const int brick = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (brick >= nbricks) return;
    const float* P = a.proj + (long long)b * a.NV * 12;
    const float* coords = a.coords + (long long)b * a.nvox * 3;
    const long long vox = brick_voxel(a, brick, lane);
    float X0 = 0.f, X1 = 0.f, X2 = 0.f;
    if (vox >= 0) { X0 = coords[vox * 3]; X1 = coords[vox * 3 + 1]; X2 = coords[vox * 3 + 2]; }
    for (int v = 0; v < a.NV; ++v) {
        int x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
        if (vox >= 0) {
            const Tap t = project_taps(P + v * 12, X0, X1, X2, a.h, a.w);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t.k[k] != 0.f) { x0 = min(x0, t.x[k]); x1 = max(x1, t.x[k]); y0 = min(y0, t.y[k]); y1 = max(y1, t.y[k]); }
            const long long ti = ((long long)b * a.NV + v) * a.nvox + vox;
            a.txy[ti] = (t.x0 + 1) | ((t.y0 + 1) << 16);
            a.tw[ti] = make_float4(t.k[0], t.k[1], t.k[2], t.k[3]);
        }
        x0 = wave_min(x0); x1 = wave_max(x1); y0 = wave_min(y0); y1 = wave_max(y1);
        if (lane == 0) *(int4*)(a.bbox + (((long long)b * a.NV + v) * nbricks + brick) * 4) = make_int4(x0, x1, y0, y1);
    }
}

// K1: grid (nblk1, B): one lane per (voxel, 4-channel vector), grid-stride (stride a multiple of C/4: a lane keeps its channel vector).
// MV = view capacity of the register arrays (4 or 8): at 4 views the kernel fits 128 VGPRs and two waves share a SIMD
template <typename T, int MV>
__global__ __launch_bounds__(256) void unproj_dx_kernel(const UnprojBwdArgs a) {
    __shared__ float red[256][4];
    const int tpv = a.C >> 2;
    const long long items = a.nvox * tpv;
    const int b = blockIdx.y;
    const T* feats = (const T*)a.feats + (long long)b * a.NV * a.h * a.w * a.C;
    const float* P = a.proj + (long long)b * a.NV * 12;
    const float* coords = a.coords + (long long)b * a.nvox * 3;
    const float* gout = a.gout + (long long)b * a.nvox * a.C;
    float* dxs = a.dxs + (long long)b * a.NV * a.nvox * a.C;
    const bool want_gc = a.gconf != nullptr;
    float gc[MV][4];
#pragma unroll
    for (int v = 0; v < MV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) gc[v][e] = 0.f;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
        const long long vox = q / tpv;
        const int c0 = (int)(q - vox * tpv) * 4;
        const float X0 = coords[vox * 3], X1 = coords[vox * 3 + 1], X2 = coords[vox * 3 + 2];
        float g[4];
        ld4<float>(gout + vox * a.C + c0, g);
        float x[MV][4], dx[MV][4];
        {
            Tap tp[MV];
            sample_views<T, MV>(a, feats, P, X0, X1, X2, c0, tp, x);
        }
        view_gradients<MV>(a, b, c0, g, x, dx);
#pragma unroll
        for (int v = 0; v < MV; ++v)
            if (v < a.NV) {
                *(float4*)(dxs + ((long long)v * a.nvox + vox) * a.C + c0) = make_float4(dx[v][0], dx[v][1], dx[v][2], dx[v][3]);
                if (want_gc)
#pragma unroll
                    for (int e = 0; e < 4; ++e) gc[v][e] += g[e] * x[v][e];
            }
    }
    if (!want_gc) return;
    // per-workgroup partial of the confidence gradient: the 256 / tpv threads that share a channel vector, added in thread order
    const int cl = threadIdx.x % tpv;            // the stride (gridDim.x * 256) and 256 are multiples of tpv (C / 4 a power of two <= 64)
    for (int v = 0; v < a.NV; ++v) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = gc[v][e];
        __syncthreads();
        if (threadIdx.x < tpv) {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            for (int t = threadIdx.x; t < 256; t += tpv)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += (double)red[t][e];
            double* dst = a.gcpart + (((long long)b * gridDim.x + blockIdx.x) * a.NV + v) * a.C + cl * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = s[e];
        }
    }
}

// one thread per (b, c): the partials of every workgroup in index order; conf_norm: chain rule through c_v / sum_u c_u
__global__ void unproj_gconf_finalize_kernel(const UnprojBwdArgs a, int nblk1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.C) return;
    const int b = i / a.C, c = i - b * a.C;
    double s[UB_MAXV];
    for (int v = 0; v < a.NV; ++v) {
        double t = 0.0;
        for (int k = 0; k < nblk1; ++k) t += a.gcpart[(((long long)b * nblk1 + k) * a.NV + v) * a.C + c];
        s[v] = t;
    }
    if (a.agg == LT_AGG_CONF_NORM) {      // s[v] is the gradient of the NORMALISED confidence n_v = c_v / S: d/dc_v = (s_v - sum_u s_u n_u) / S
        double S = 0.0, dot = 0.0;
        for (int v = 0; v < a.NV; ++v) S += (double)a.conf[((long long)b * a.NV + v) * a.C + c];
        for (int v = 0; v < a.NV; ++v) dot += s[v] * (double)a.conf[((long long)b * a.NV + v) * a.C + c] / S;
        for (int v = 0; v < a.NV; ++v) s[v] = (s[v] - dot) / S;
    }
    for (int v = 0; v < a.NV; ++v) a.gconf[((long long)b * a.NV + v) * a.C + c] = (float)s[v];
}

// K2: grid (tiles_x * tiles_y, NV, B), 4 waves; wave wv owns channels [wv * CW, (wv + 1) * CW) of the tile, CW = C / 4.
// LDS tile: HALVES half tiles of G_TW / HALVES pixel columns, [half][row][column][C + 8 floats] (+16 per row, +32 per half: the four taps
// of a voxel land in different banks).  With 4 * CW <= 32 lanes per hit (C <= 32) the two half-waves work on DIFFERENT hits at the same
// time -- half-wave 0 adds only taps in pixel columns 0-7, half-wave 1 only columns 8-15, each walking its own compact list of the
// brick's voxels that touch its half -- so no cell is ever addressed by both, and the per-cell order stays (brick, voxel).
constexpr int G_TW = 16, G_TH = 16;
constexpr int G_LIST = 256;      // candidate bricks per refill of the list

template <int CW>
struct GatherCfg {
    static constexpr int C = 4 * CW;
    static constexpr int HALVES = (4 * CW <= 32) ? 2 : 1;
    static constexpr int HPX = G_TW / HALVES;            // pixel columns per half
    static constexpr int PS = C + 8;                     // floats per pixel
    static constexpr int RSH = HPX * PS + 16;            // floats per row of a half
    static constexpr int HALF_SZ = G_TH * RSH + 32;
    static constexpr int TILE = HALVES * HALF_SZ;        // + 4 * 64 dummy cells (one per thread: where the adds of absent taps go)
    static constexpr int DW = CW < 4 ? 4 : CW;           // floats of dx per record (16-byte granules)
    static constexpr int REC = 4 + 4 + DW;               // ints per hit record: 4 cell byte offsets, 4 weights, the wave's dx channels
    static constexpr int WAVE_INTS = HALVES * 68 * REC;  // per half 64 records + one round of padding
    static constexpr size_t LDS = (size_t)(TILE + 256 + 4 * WAVE_INTS + G_LIST + 4) * 4;
};

template <int CW>
__global__ __launch_bounds__(256) void unproj_gather_kernel(const UnprojBwdArgs a) {
    typedef GatherCfg<CW> G;
    constexpr int C = G::C, HALVES = G::HALVES, REC = G::REC;
    extern __shared__ float smem[];
    float* tile = smem;
    float* dummy = smem + G::TILE;                             // [256]
    int* wbase = (int*)(dummy + 256);
    int* list = wbase + 4 * G::WAVE_INTS;                      // [G_LIST] candidate bricks of this tile, ascending
    int* meta = list + G_LIST;                                 // count, next brick group
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int* myR = wbase + wv * G::WAVE_INTS;                      // [HALVES][68] hit records
    const int tiles_x = (a.w + G_TW - 1) / G_TW;
    const int tx0 = (blockIdx.x % tiles_x) * G_TW, ty0 = (blockIdx.x / tiles_x) * G_TH;
    const int v = blockIdx.y, b = blockIdx.z;
    const int nbricks = a.nb0 * a.nb1 * a.nb2;
    const long long vbase = ((long long)b * a.NV + v) * a.nvox;
    const int* txy = a.txy + vbase;
    const float4* tw = a.tw + vbase;
    const float* dxs = a.dxs + vbase * C + wv * CW;
    const int* bbox = a.bbox + ((long long)b * a.NV + v) * nbricks * 4;
    for (int i = threadIdx.x; i < G::TILE; i += 256) tile[i] = 0.f;
    // this lane's role in the hit loop: half-wave `half`, pixel-parity class t_of (see the records), channel c_of (lanes past 4 * CW of a
    // half idle: they work on a dummy cell)
    const int half = HALVES == 2 ? lane >> 5 : 0, hl_lane = HALVES == 2 ? lane & 31 : lane;
    const int t_of = hl_lane / CW, c_of = hl_lane - t_of * CW;
    const bool adder = hl_lane < 4 * CW;
    // byte addresses (LDS) this lane works with: its channel inside a pixel's cell run, its dummy cell, its fields of record 0 of its half
    char* const cell_base = (char*)(tile + wv * CW + c_of);
    const int dummy_off = (int)((char*)(dummy + threadIdx.x) - cell_base);
    const char* const rec_c = (const char*)(myR + half * 68 * REC + (adder ? t_of : 0));
    const char* const rec_w = rec_c + 16;
    const char* const rec_d = (const char*)(myR + half * 68 * REC + 8 + (adder ? c_of : 0));
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    struct Req { long long vox; int xy; float4 w; float4 d[G::DW / 4]; };
    auto request = [&](int brick) {                            // this lane's voxel of `brick`: its tap record and this wave's dx channels
        Req r;
        r.vox = brick_voxel(a, brick, lane);
        const long long vx = r.vox >= 0 ? r.vox : 0;           // a lane past the grid's end reads voxel 0 and is masked later
        r.xy = txy[vx];
        r.w = tw[vx];
        const float* src = dxs + vx * C;
        if (CW >= 4) {
#pragma unroll
            for (int e = 0; e < CW / 4; ++e) r.d[e] = *(const float4*)(src + e * 4);
        } else {
            r.d[0] = make_float4(src[0], CW > 1 ? src[CW > 1 ? 1 : 0] : 0.f, 0.f, 0.f);
        }
        return r;
    };
    int g_next = 0;
    while (g_next < nbricks) {
        // ---- wave 0 lists the next (up to G_LIST) bricks whose tap bounding box meets the tile, in brick order
        if (wv == 0) {
            int cnt = 0, g0 = g_next;
            while (g0 < nbricks && cnt <= G_LIST - 64) {
                bool cand = false;
                if (g0 + lane < nbricks) {
                    const int4 bb = *(const int4*)(bbox + (long long)(g0 + lane) * 4);
                    cand = bb.x <= bb.y && bb.y >= tx0 && bb.x < tx0 + G_TW && bb.w >= ty0 && bb.z < ty0 + G_TH;
                }
                const unsigned long long m = __ballot(cand);
                if (cand) list[cnt + __popcll(m & lt_mask)] = g0 + lane;
                cnt += __popcll(m);
                g0 += 64;
            }
            if (lane == 0) { meta[0] = cnt; meta[1] = g0; }
        }
        __syncthreads();                                       // (also orders the tile's zero fill in front of the first additions)
        const int cnt = meta[0];
        g_next = meta[1];
        // ---- every wave walks the list (its own channels): brick i + 1's tap records and dx values are in flight while brick i is added
        Req nxt;
        if (cnt > 0) nxt = request(list[0]);
        for (int i = 0; i < cnt; ++i) {
            const Req cur = nxt;
            if (i + 1 < cnt) nxt = request(list[i + 1]);
            // the voxel's taps that fall into this tile: byte offset of the pixel's cell run (relative to the tile), per half
            int L[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}};
            bool hit[2] = {false, false};
            const float Wt[4] = {cur.w.x, cur.w.y, cur.w.z, cur.w.w};
            if (cur.vox >= 0) {
                const int lx0 = (cur.xy & 0xffff) - 1 - tx0, ly0 = (cur.xy >> 16) - 1 - ty0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int lx = lx0 + (k & 1), ly = ly0 + (k >> 1);
                    if (Wt[k] != 0.f && lx >= 0 && lx < G_TW && ly >= 0 && ly < G_TH) {
                        const int hh = HALVES == 2 ? lx >> 3 : 0;
                        const int off = (hh * G::HALF_SZ + ly * G::RSH + (lx - hh * G::HPX) * G::PS) * 4;
                        if (hh == 0) { L[0][k] = off; hit[0] = true; } else { L[1][k] = off; hit[1] = true; }
                    }
                }
            }
            const unsigned long long m0 = __ballot(hit[0]), m1 = HALVES == 2 ? __ballot(hit[1]) : 0ull;
            if (!(m0 | m1)) continue;
            const int n0 = __popcll(m0), n1 = __popcll(m1);
            // record slots are PIXEL PARITY classes, not tap numbers: slot s holds the tap whose pixel has (x & 1) + 2 (y & 1) == s, i.e.
            // tap k = s ^ parity(origin).  The four taps of a voxel always have four different parities, and lane group s of the hit loop
            // then owns every cell of parity s -- a pixel reached as tap 0 of one voxel and tap 1 of the next is updated by the SAME lane,
            // so the read-add-write below needs no atomics (the tile origin is a multiple of 16: tile-relative parity = map parity)
            float Ws[4];
            {
                const int lx0 = (cur.xy & 0xffff) - 1, ly0 = (cur.xy >> 16) - 1;
                const bool p1 = lx0 & 1, p2 = ly0 & 1;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int t0 = p1 ? L[hh][1] : L[hh][0], t1 = p1 ? L[hh][0] : L[hh][1], t2 = p1 ? L[hh][3] : L[hh][2], t3 = p1 ? L[hh][2] : L[hh][3];
                    L[hh][0] = p2 ? t2 : t0; L[hh][1] = p2 ? t3 : t1; L[hh][2] = p2 ? t0 : t2; L[hh][3] = p2 ? t1 : t3;
                }
                const float t0 = p1 ? Wt[1] : Wt[0], t1 = p1 ? Wt[0] : Wt[1], t2 = p1 ? Wt[3] : Wt[2], t3 = p1 ? Wt[2] : Wt[3];
                Ws[0] = p2 ? t2 : t0; Ws[1] = p2 ? t3 : t1; Ws[2] = p2 ? t0 : t2; Ws[3] = p2 ? t1 : t3;
            }
            // compact hit records per half, in voxel order: [4 cell offsets | 4 weights | dx of this wave's channels].  Same-wave LDS traffic: a wave's LDS queue is in order, the fences stop the compiler
#pragma unroll
            for (int hh = 0; hh < HALVES; ++hh) {
                const unsigned long long m = hh == 0 ? m0 : m1;
                int* rb = myR + hh * 68 * REC;
                if (hit[hh]) {
                    int* r = rb + __popcll(m & lt_mask) * REC;
                    *(int4*)r = make_int4(L[hh][0], L[hh][1], L[hh][2], L[hh][3]);
                    *(float4*)(r + 4) = make_float4(Ws[0], Ws[1], Ws[2], Ws[3]);
#pragma unroll
                    for (int e = 0; e < G::DW / 4; ++e) *(float4*)(r + 8 + e * 4) = cur.d[e];
                }
            }
            const int nmax = HALVES == 2 ? max(n0, n1) : n0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // rounds of four hits: twelve independent record reads, four cell reads, the sums (forwarded in registers where two hits of
            // the round meet in one cell), four cell writes in hit order.  A half with fewer hits re-reads stale / empty records into its
            // dummy cell (records past its count are never trusted: the count test below)
            const int my_n = half == 0 ? n0 : n1;
            for (int r0 = 0; r0 < nmax; r0 += 4) {
                int off[4];
                float x[4], cv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ro = (r0 + j) * REC * 4;
                    const int cell = *(const int*)(rec_c + ro);
                    const float wt = *(const float*)(rec_w + ro), dv = *(const float*)(rec_d + ro);
                    const bool ok = adder && cell >= 0 && r0 + j < my_n;
                    off[j] = ok ? cell : dummy_off;
                    x[j] = wt * dv;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) cv[j] = *(const float*)(cell_base + off[j]);
                const float a0 = cv[0] + x[0];
                const float a1 = (off[1] == off[0] ? a0 : cv[1]) + x[1];
                const float a2 = (off[2] == off[1] ? a1 : (off[2] == off[0] ? a0 : cv[2])) + x[2];
                const float a3 = (off[3] == off[2] ? a2 : (off[3] == off[1] ? a1 : (off[3] == off[0] ? a0 : cv[3]))) + x[3];
                *(float*)(cell_base + off[0]) = a0;
                *(float*)(cell_base + off[1]) = a1;
                *(float*)(cell_base + off[2]) = a2;
                *(float*)(cell_base + off[3]) = a3;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();                                       // the list is free to be rewritten; after the last round: the tile is complete
    }
    // the tile's pixels inside the map, every channel: plain stores, each cell of gfeats written exactly once
    float* gm = a.gfeats + ((long long)b * a.NV + v) * a.h * a.w * C;
    constexpr int c4 = C / 4;
    for (int i = threadIdx.x; i < G_TH * G_TW * c4; i += 256) {
        const int cv = i % c4, px = (i / c4) % G_TW, py = i / (c4 * G_TW);
        if (tx0 + px < a.w && ty0 + py < a.h) {
            const int hh = px / G::HPX;
            *(float4*)(gm + ((long long)(ty0 + py) * a.w + tx0 + px) * C + cv * 4) =
                *(const float4*)(tile + hh * G::HALF_SZ + py * G::RSH + (px - hh * G::HPX) * G::PS + cv * 4);
        }
    }
}

template <int CW>
int launch_gather(const UnprojBwdArgs& c, int tiles, int NV, int nb, hipStream_t st) {
    LT_OPT_IN_LDS(unproj_gather_kernel<CW>, 160 * 1024);
    hipLaunchKernelGGL(unproj_gather_kernel<CW>, dim3((unsigned)tiles, (unsigned)NV, (unsigned)nb), dim3(256), GatherCfg<CW>::LDS, st, c);
    LT_CHECK_LAUNCH("lt_unproject_bwd(gather)");
    return LT_OK;
}

int blocks_k1(long long nvox, int C) {
    const long long blocks = cdiv(nvox * (C / 4), 256);
    return (int)(blocks < 2048 ? blocks : 2048);
}

bool gather_takes(int C) { return C >= 4 && C <= 64 && (C & (C - 1)) == 0 && getenv("LT_UNPROJ_BWD_ATOMICS") == nullptr; }

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

// bytes for ALL B samples at once; lt_unproject_bwd walks the batch in chunks when it is handed less (at least one sample's worth)
extern "C" size_t lt_unproject_bwd_workspace(int32_t B, int32_t NV, int32_t C, int32_t v0, int32_t v1, int32_t v2) {
    if (B < 1 || NV < 1 || C < 4 || v0 < 1 || v1 < 1 || v2 < 1 || !gather_takes(C)) return 0;
    const long long nvox = (long long)v0 * v1 * v2;
    const long long nbricks = cdiv(v0, 4) * cdiv(v1, 4) * cdiv(v2, 4);
    const size_t per_sample = align256((size_t)NV * nvox * C * 4) + align256((size_t)NV * nbricks * 16) + align256((size_t)blocks_k1(nvox, C) * NV * C * 8) +
                              align256((size_t)NV * nvox * 4) + align256((size_t)NV * nvox * 16);
    return per_sample * (size_t)B;
}

extern "C" int lt_unproject_bwd(int32_t dtype, const void* feats, const float* proj, const float* coords, const float* conf, const float* grad_out,
                                float* grad_feats, float* grad_conf, int32_t B, int32_t NV, int32_t C, int32_t h, int32_t w, int32_t v0, int32_t v1,
                                int32_t v2, int32_t agg, void* workspace, size_t workspace_bytes, void* stream) {
    LT_REQUIRE(feats && proj && coords && grad_out && grad_feats, LT_ERR_INVALID, "lt_unproject_bwd: null argument");
    LT_REQUIRE(dtype == LT_F32 || dtype == LT_BF16, LT_ERR_INVALID, "lt_unproject_bwd: bad dtype %d", dtype);
    LT_REQUIRE(agg == LT_AGG_SUM || agg == LT_AGG_MAX || agg == LT_AGG_SOFTMAX || agg == LT_AGG_CONF || agg == LT_AGG_CONF_NORM, LT_ERR_UNSUPPORTED,
               "lt_unproject_bwd: unknown aggregation %d", agg);
    const bool is_conf = agg == LT_AGG_CONF || agg == LT_AGG_CONF_NORM;
    LT_REQUIRE(!is_conf || conf, LT_ERR_INVALID, "lt_unproject_bwd: the conf aggregations need confidences");
    LT_REQUIRE(B >= 1 && NV >= 1 && NV <= UB_MAXV && C >= 4 && C % 4 == 0 && h >= 2 && w >= 2 && v0 >= 1 && v1 >= 1 && v2 >= 1, LT_ERR_UNSUPPORTED,
               "lt_unproject_bwd: needs 1 <= NV <= %d and C %% 4 == 0 (got NV=%d C=%d)", UB_MAXV, NV, C);
    LT_REQUIRE((long long)NV * h * w * C < (1ll << 31), LT_ERR_UNSUPPORTED, "lt_unproject_bwd: feature maps too large");
    hipStream_t st = (hipStream_t)stream;
    UnprojBwdArgs a;
    a.feats = feats; a.proj = proj; a.coords = coords; a.conf = conf; a.gout = grad_out; a.gfeats = grad_feats; a.gconf = is_conf ? grad_conf : nullptr;
    a.dxs = nullptr; a.bbox = nullptr; a.gcpart = nullptr; a.txy = nullptr; a.tw = nullptr;
    a.B = B; a.NV = NV; a.C = C; a.h = h; a.w = w; a.agg = agg;
    a.v0 = v0; a.v1 = v1; a.v2 = v2; a.nb0 = (int)cdiv(v0, 4); a.nb1 = (int)cdiv(v1, 4); a.nb2 = (int)cdiv(v2, 4);
    a.nvox = (long long)v0 * v1 * v2;
    const size_t fbytes = (size_t)NV * h * w * C * 4;
    if (!gather_takes(C)) {
        // round-2 scatter: global float atomics into zeroed buffers (not bitwise repeatable)
        LT_REQUIRE(agg != LT_AGG_CONF_NORM, LT_ERR_UNSUPPORTED, "lt_unproject_bwd: conf_norm needs the gather path (C a power of two <= 64)");
        LT_REQUIRE(hipMemsetAsync(grad_feats, 0, fbytes * B, st) == hipSuccess, LT_ERR_LAUNCH, "lt_unproject_bwd: hipMemsetAsync failed");
        if (a.gconf) LT_REQUIRE(hipMemsetAsync(a.gconf, 0, (size_t)B * NV * C * 4, st) == hipSuccess, LT_ERR_LAUNCH, "lt_unproject_bwd: hipMemsetAsync failed");
        const long long blocks = cdiv(a.nvox * (C / 4), 256);
        dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536), (unsigned)B);
        if (dtype == LT_F32) hipLaunchKernelGGL(unproject_bwd_kernel<float>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(unproject_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, a);
        LT_CHECK_LAUNCH("lt_unproject_bwd(scatter)");
        return LT_OK;
    }
    const int nbricks = a.nb0 * a.nb1 * a.nb2;
    const int nblk1 = blocks_k1(a.nvox, C);
    const size_t s_dx = align256((size_t)NV * a.nvox * C * 4), s_bb = align256((size_t)NV * nbricks * 16), s_gc = align256((size_t)nblk1 * NV * C * 8);
    const size_t s_xy = align256((size_t)NV * a.nvox * 4), s_tw = align256((size_t)NV * a.nvox * 16);
    const size_t per_sample = s_dx + s_bb + s_gc + s_xy + s_tw;
    LT_REQUIRE(workspace && workspace_bytes >= per_sample && ((size_t)workspace % 16) == 0, LT_ERR_INVALID,
               "lt_unproject_bwd: workspace of at least %zu bytes (one sample; lt_unproject_bwd_workspace for the whole batch), 16-byte aligned", per_sample);
    const int chunk = (int)(workspace_bytes / per_sample < (size_t)B ? workspace_bytes / per_sample : (size_t)B);
    const int tiles = (int)(cdiv(w, G_TW) * cdiv(h, G_TH));
    const size_t in_elt = dtype == LT_F32 ? 4 : 2;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = B - b0 < chunk ? B - b0 : chunk;
        UnprojBwdArgs c = a;
        c.B = nb;
        c.feats = (const char*)feats + (size_t)b0 * NV * h * w * C * in_elt;
        c.proj = proj + (size_t)b0 * NV * 12;
        c.coords = coords + (size_t)b0 * a.nvox * 3;
        c.conf = conf ? conf + (size_t)b0 * NV * C : nullptr;
        c.gout = grad_out + (size_t)b0 * a.nvox * C;
        c.gfeats = grad_feats + (size_t)b0 * NV * h * w * C;
        c.gconf = a.gconf ? a.gconf + (size_t)b0 * NV * C : nullptr;
        char* wsp = (char*)workspace;
        c.dxs = (float*)wsp; wsp += s_dx * nb;
        c.bbox = (int*)wsp; wsp += s_bb * nb;
        c.gcpart = (double*)wsp; wsp += s_gc * nb;
        c.tw = (float4*)wsp; wsp += s_tw * nb;
        c.txy = (int*)wsp;
        hipLaunchKernelGGL(unproj_taps_kernel, dim3((unsigned)cdiv(nbricks, 4), (unsigned)nb), dim3(256), 0, st, c);
        LT_CHECK_LAUNCH("lt_unproject_bwd(taps)");
        const dim3 g1((unsigned)nblk1, (unsigned)nb);
        if (dtype == LT_F32) {
            if (NV <= 4) hipLaunchKernelGGL((unproj_dx_kernel<float, 4>), g1, dim3(256), 0, st, c);
            else hipLaunchKernelGGL((unproj_dx_kernel<float, UB_MAXV>), g1, dim3(256), 0, st, c);
        } else {
            if (NV <= 4) hipLaunchKernelGGL((unproj_dx_kernel<bf16_t, 4>), g1, dim3(256), 0, st, c);
            else hipLaunchKernelGGL((unproj_dx_kernel<bf16_t, UB_MAXV>), g1, dim3(256), 0, st, c);
        }
        LT_CHECK_LAUNCH("lt_unproject_bwd(dx)");
        if (c.gconf) {
            hipLaunchKernelGGL(unproj_gconf_finalize_kernel, dim3((unsigned)cdiv((long long)nb * C, 64)), dim3(64), 0, st, c, nblk1);
            LT_CHECK_LAUNCH("lt_unproject_bwd(gconf)");
        }
        int rc;
        switch (C / 4) {
            case 1: rc = launch_gather<1>(c, tiles, NV, nb, st); break;
            case 2: rc = launch_gather<2>(c, tiles, NV, nb, st); break;
            case 4: rc = launch_gather<4>(c, tiles, NV, nb, st); break;
            case 8: rc = launch_gather<8>(c, tiles, NV, nb, st); break;
            default: rc = launch_gather<16>(c, tiles, NV, nb, st); break;
        }
        if (rc != LT_OK) return rc;
    }
    return LT_OK;
}
