// Soft-argmax reductions (3D: op.integrate_tensor_3d_with_coordinates, 2D: op.integrate_tensor_2d) and
// the confidence-weighted DLT (multiview.triangulate_batch_of_points).
//
// 3D soft-argmax is HBM-bound: read J x V^3 logits (twice: statistics, then normalise) and write the
// J x V^3 probabilities the API returns.  One lane owns one voxel and walks the J joints with an online
// (max, sum, sum*coord) state per joint in registers, so the voxel's coordinates are read once for all
// joints; channels-last logits are staged through an LDS tile with an odd row stride, which turns the
// strided per-joint walk into conflict-free LDS reads while global loads stay fully coalesced.
// Planar logits ((B, J, V^3), what lt_pwchain_fwd writes for the bf16 V2V tail) need no staging: a lane owns FOUR consecutive
// voxels, every joint is one 16-byte load per lane (all J of them in flight at once) and the probabilities are written by a
// plain vectorised elementwise pass.
#include "lt_common.h"

using namespace lt;

namespace {

constexpr int SA_ITERS = 8;                  // voxels per lane
constexpr int SA_CHUNK = 256 * SA_ITERS;     // voxels per workgroup
constexpr int SA_REC = 5;                    // partial record: m, s, sx, sy, sz
typedef float sa_f32x4 __attribute__((ext_vector_type(4)));

struct SA3Args {
    const float* logits;
    const float* coords;
    float* partial;   // [B][J][nchunks][5]
    float* stats;     // [B][J][2] = max, sum
    float* kp;        // [B][J][3]
    float* probs;     // [B][J][nvox] or null
    float mult;
    int softmax, J, ld, nchunks;
    long long nvox;
};

// stage a [nv][J] tile of channels-last logits into LDS (row stride J|1 -> odd)
__device__ __forceinline__ void stage_cl(const SA3Args& a, const float* src_b, long long vox0, int nv, float* tile, int ts) {
    if (a.ld == a.J) {
        const float* src = src_b + vox0 * a.J;
        for (int i = threadIdx.x; i < nv * a.J; i += 256) {
            const int v = i / a.J;
            tile[v * ts + (i - v * a.J)] = src[i];
        }
    } else {
        for (int i = threadIdx.x; i < nv * a.J; i += 256) {
            const int v = i / a.J, j = i - v * a.J;
            tile[v * ts + j] = src_b[(vox0 + v) * a.ld + j];
        }
    }
}

template <int JP, bool CL>
__global__ __launch_bounds__(256, (JP <= 17 ? 3 : 2)) void sa3_partial_kernel(const SA3Args a) {   // 3 workgroups per SIMD-quad fit at <= 17 joints (152 VGPRs)
    extern __shared__ float smem[];
    const int ts = a.J | 1;
    float* tile = smem;                       // CL only: [256][ts]
    float* red = smem + (CL ? 256 * ts : 0);  // [4][JP][5]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float* lg = a.logits + (long long)b * a.J * a.nvox;
    const float* cd = a.coords + (long long)b * a.nvox * 3;
    float m[JP], s[JP], sx[JP], sy[JP], sz[JP];
#pragma unroll
    for (int j = 0; j < JP; ++j) { m[j] = -INFINITY; s[j] = sx[j] = sy[j] = sz[j] = 0.f; }
    for (int it = 0; it < SA_ITERS; ++it) {
        const long long vox0 = (long long)chunk * SA_CHUNK + it * 256;
        if (vox0 >= a.nvox) break;
        const int nv = (int)min((long long)256, a.nvox - vox0);
        if (CL) {
            __syncthreads();
            stage_cl(a, lg, vox0, nv, tile, ts);
            __syncthreads();
        }
        const int t = threadIdx.x;
        if (t < nv) {
            const long long vox = vox0 + t;
            const float cx = cd[vox * 3], cy = cd[vox * 3 + 1], cz = cd[vox * 3 + 2];
#pragma unroll
            for (int j = 0; j < JP; ++j) {
                if (j < a.J) {
                    const float x = a.mult * (CL ? tile[t * ts + j] : lg[(long long)j * a.nvox + vox]);
                    if (a.softmax) {
                        const float mn = fmaxf(m[j], x);
                        const float sc = expf(m[j] - mn), e = expf(x - mn);  // m = -inf -> sc = 0
                        s[j] = s[j] * sc + e;
                        sx[j] = sx[j] * sc + e * cx;
                        sy[j] = sy[j] * sc + e * cy;
                        sz[j] = sz[j] * sc + e * cz;
                        m[j] = mn;
                    } else {
                        const float p = fmaxf(x, 0.f);
                        s[j] += p; sx[j] += p * cx; sy[j] += p * cy; sz[j] += p * cz;
                        m[j] = 0.f;
                    }
                }
            }
        }
    }
    // wave reduction (64 lanes), then the 4 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        if (j < a.J) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float m2 = __shfl_xor(m[j], off), s2 = __shfl_xor(s[j], off);
                const float x2 = __shfl_xor(sx[j], off), y2 = __shfl_xor(sy[j], off), z2 = __shfl_xor(sz[j], off);
                const float mn = fmaxf(m[j], m2);
                const float c1 = (m[j] == -INFINITY) ? 0.f : expf(m[j] - mn), c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
                s[j] = s[j] * c1 + s2 * c2; sx[j] = sx[j] * c1 + x2 * c2; sy[j] = sy[j] * c1 + y2 * c2; sz[j] = sz[j] * c1 + z2 * c2;
                m[j] = mn;
            }
            if (lane == 0) {
                float* r = red + (wave * JP + j) * SA_REC;
                r[0] = m[j]; r[1] = s[j]; r[2] = sx[j]; r[3] = sy[j]; r[4] = sz[j];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < a.J) {
        const int j = threadIdx.x;
        float M = -INFINITY;
        for (int w = 0; w < 4; ++w) M = fmaxf(M, red[(w * JP + j) * SA_REC]);
        float S = 0.f, X = 0.f, Y = 0.f, Z = 0.f;
        for (int w = 0; w < 4; ++w) {
            const float* r = red + (w * JP + j) * SA_REC;
            const float c = (r[0] == -INFINITY) ? 0.f : expf(r[0] - M);
            S += r[1] * c; X += r[2] * c; Y += r[3] * c; Z += r[4] * c;
        }
        float* o = a.partial + (((long long)b * a.J + j) * a.nchunks + chunk) * SA_REC;
        o[0] = M; o[1] = S; o[2] = X; o[3] = Y; o[4] = Z;
    }
}

// Planar logits, nvox % 4 == 0: a lane owns SAP_V groups of four consecutive voxels (their coordinates stay in registers for
// all joints), the workgroup walks the joints: per joint one 16-byte load per group (the next joint's loads are in flight
// under this joint's arithmetic), an exact two-step softmax over the lane's 16 values, then one wave reduction
// (max -> rescale -> sums).  Register state is independent of J, so this kernel runs at 3+ waves per SIMD.
constexpr int SAP_V = 4;
constexpr int SAP_CHUNK = 256 * 4 * SAP_V;   // voxels per workgroup

__global__ __launch_bounds__(256, 3) void sa3_partial_planar_kernel(const SA3Args a) {
    __shared__ float red[32 * 4 * SA_REC];    // [J][wave][5]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float* lg = a.logits + (long long)b * a.J * a.nvox;
    const float* cd = a.coords + (long long)b * a.nvox * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long vox[SAP_V];
    float wgt[SAP_V];
    sa_f32x4 ca[SAP_V], cb[SAP_V], cc[SAP_V];   // x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3
#pragma unroll
    for (int k = 0; k < SAP_V; ++k) {
        const long long v = (long long)chunk * SAP_CHUNK + k * 1024 + threadIdx.x * 4;
        const bool ok = v < a.nvox;          // nvox % 4 == 0: a group is inside or outside as a whole
        wgt[k] = ok ? 1.f : 0.f;             // outside: re-read group 0 (finite values of the same volume) with weight 0
        vox[k] = ok ? v : 0;
        const sa_f32x4* c4 = (const sa_f32x4*)(cd + vox[k] * 3);
        ca[k] = c4[0]; cb[k] = c4[1]; cc[k] = c4[2];
    }
    sa_f32x4 cur[SAP_V], nxt[SAP_V];
#pragma unroll
    for (int k = 0; k < SAP_V; ++k) cur[k] = *(const sa_f32x4*)(lg + vox[k]);
#pragma unroll 1
    for (int j = 0; j < a.J; ++j) {
        const float* nl = lg + (long long)min(j + 1, a.J - 1) * a.nvox;
#pragma unroll
        for (int k = 0; k < SAP_V; ++k) nxt[k] = *(const sa_f32x4*)(nl + vox[k]);
        float m = 0.f;
        if (a.softmax) {
            m = -INFINITY;
#pragma unroll
            for (int k = 0; k < SAP_V; ++k) {
                cur[k] *= a.mult;
                m = fmaxf(fmaxf(m, fmaxf(cur[k][0], cur[k][1])), fmaxf(cur[k][2], cur[k][3]));
            }
        }
        float s = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int k = 0; k < SAP_V; ++k) {
            float e0, e1, e2, e3;
            if (a.softmax) {
                e0 = expf(cur[k][0] - m); e1 = expf(cur[k][1] - m); e2 = expf(cur[k][2] - m); e3 = expf(cur[k][3] - m);
            } else {
                e0 = fmaxf(a.mult * cur[k][0], 0.f); e1 = fmaxf(a.mult * cur[k][1], 0.f);
                e2 = fmaxf(a.mult * cur[k][2], 0.f); e3 = fmaxf(a.mult * cur[k][3], 0.f);
            }
            e0 *= wgt[k]; e1 *= wgt[k]; e2 *= wgt[k]; e3 *= wgt[k];
            s += (e0 + e1) + (e2 + e3);
            sx += (e0 * ca[k][0] + e1 * ca[k][3]) + (e2 * cb[k][2] + e3 * cc[k][1]);
            sy += (e0 * ca[k][1] + e1 * cb[k][0]) + (e2 * cb[k][3] + e3 * cc[k][2]);
            sz += (e0 * ca[k][2] + e1 * cb[k][1]) + (e2 * cc[k][0] + e3 * cc[k][3]);
        }
        float M = m;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
        if (a.softmax) {
            const float c = expf(m - M);
            s *= c; sx *= c; sy *= c; sz *= c;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s += __shfl_xor(s, off); sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off);
        }
        if (lane == 0) {
            float* r = red + (j * 4 + wave) * SA_REC;
            r[0] = M; r[1] = s; r[2] = sx; r[3] = sy; r[4] = sz;
        }
#pragma unroll
        for (int k = 0; k < SAP_V; ++k) cur[k] = nxt[k];
    }
    __syncthreads();
    if (threadIdx.x < a.J) {
        const int j = threadIdx.x;
        float M = -INFINITY;
        for (int w = 0; w < 4; ++w) M = fmaxf(M, red[(j * 4 + w) * SA_REC]);
        float S = 0.f, X = 0.f, Y = 0.f, Z = 0.f;
        for (int w = 0; w < 4; ++w) {
            const float* r = red + (j * 4 + w) * SA_REC;
            const float c = expf(r[0] - M);
            S += r[1] * c; X += r[2] * c; Y += r[3] * c; Z += r[4] * c;
        }
        float* o = a.partial + (((long long)b * a.J + j) * a.nchunks + chunk) * SA_REC;
        o[0] = M; o[1] = S; o[2] = X; o[3] = Y; o[4] = Z;
    }
}

// planar logits -> planar probabilities: elementwise with per-(b, joint) statistics, four 16-byte vectors per lane in flight
__global__ __launch_bounds__(256) void sa3_probs_planar_kernel(const SA3Args a) {
    const long long plane = blockIdx.y;       // b * J + joint
    const float M = a.stats[plane * 2], S = a.stats[plane * 2 + 1];
    const float4* src = (const float4*)(a.logits + plane * a.nvox);
    float4* dst = (float4*)(a.probs + plane * a.nvox);
    const long long n4 = a.nvox / 4, i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k * 256 < n4) v[k] = src[i0 + k * 256];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k * 256 < n4) {
            float4 p;
            if (a.softmax) {
                p.x = __fdiv_rn(expf(a.mult * v[k].x - M), S); p.y = __fdiv_rn(expf(a.mult * v[k].y - M), S);
                p.z = __fdiv_rn(expf(a.mult * v[k].z - M), S); p.w = __fdiv_rn(expf(a.mult * v[k].w - M), S);
            } else {
                p.x = fmaxf(a.mult * v[k].x, 0.f); p.y = fmaxf(a.mult * v[k].y, 0.f);
                p.z = fmaxf(a.mult * v[k].z, 0.f); p.w = fmaxf(a.mult * v[k].w, 0.f);
            }
            dst[i0 + k * 256] = p;
        }
    }
}

// one wave per (b, joint): combine the chunk partials in fp64
__global__ void sa3_finalize_kernel(const SA3Args a, int B) {
    const int bj = blockIdx.x;
    const float* p = a.partial + (long long)bj * a.nchunks * SA_REC;
    const int lane = threadIdx.x;
    float M = -INFINITY;
    for (int c = lane; c < a.nchunks; c += 64) M = fmaxf(M, p[c * SA_REC]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
    double S = 0, X = 0, Y = 0, Z = 0;
    for (int c = lane; c < a.nchunks; c += 64) {
        const float* r = p + c * SA_REC;
        const double w = (r[0] == -INFINITY) ? 0.0 : (double)expf(r[0] - M);
        S += r[1] * w; X += r[2] * w; Y += r[3] * w; Z += r[4] * w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        S += __shfl_xor(S, off); X += __shfl_xor(X, off); Y += __shfl_xor(Y, off); Z += __shfl_xor(Z, off);
    }
    if (lane == 0) {
        a.stats[bj * 2] = M;
        a.stats[bj * 2 + 1] = (float)S;
        const double d = a.softmax ? S : 1.0;  // op.py:90-94: the ReLU variant is NOT normalised
        a.kp[bj * 3] = (float)(X / d); a.kp[bj * 3 + 1] = (float)(Y / d); a.kp[bj * 3 + 2] = (float)(Z / d);
    }
}

template <bool CL>
__global__ __launch_bounds__(256) void sa3_probs_kernel(const SA3Args a) {
    extern __shared__ float smem[];
    const int ts = a.J | 1;
    float* tile = smem;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float* lg = a.logits + (long long)b * a.J * a.nvox;
    const float* st = a.stats + (long long)b * a.J * 2;
    float* pr = a.probs + (long long)b * a.J * a.nvox;
    for (int it = 0; it < SA_ITERS; ++it) {
        const long long vox0 = (long long)chunk * SA_CHUNK + it * 256;
        if (vox0 >= a.nvox) break;
        const int nv = (int)min((long long)256, a.nvox - vox0);
        if (CL) {
            __syncthreads();
            stage_cl(a, lg, vox0, nv, tile, ts);
            __syncthreads();
        }
        const int t = threadIdx.x;
        if (t < nv) {
            const long long vox = vox0 + t;
            for (int j = 0; j < a.J; ++j) {
                const float x = a.mult * (CL ? tile[t * ts + j] : lg[(long long)j * a.nvox + vox]);
                const float p = a.softmax ? __fdiv_rn(expf(x - st[j * 2]), st[j * 2 + 1]) : fmaxf(x, 0.f);
                pr[(long long)j * a.nvox + vox] = p;
            }
        }
    }
}

// planar logits whose volumes are float4-addressable take the vectorised kernels
bool sa3_planar_vec(const SA3Args& a) {
    static const bool off = getenv("LT_SA3_NO_VEC") != nullptr;
    return !off && a.nvox % 4 == 0 && ((uintptr_t)a.logits & 15) == 0 && ((uintptr_t)a.coords & 15) == 0 &&
           (!a.probs || ((uintptr_t)a.probs & 15) == 0);
}

template <bool CL>
int sa3_launch_partial(const SA3Args& a, int B, hipStream_t st) {
    const int J = a.J;
    // 17 joints (COCO / Human3.6M skeletons) get an exact instantiation: 35 fewer state registers than the 24-wide one
    const int JP = J == 17 ? 17 : (J <= 8 ? 8 : (J <= 16 ? 16 : (J <= 24 ? 24 : 32)));
    const size_t lds = ((CL ? 256 * (J | 1) : 0) + 4 * JP * SA_REC) * sizeof(float);
    const dim3 grid(a.nchunks, B), blk(256);
    switch (JP) {
        case 8: hipLaunchKernelGGL((sa3_partial_kernel<8, CL>), grid, blk, lds, st, a); break;
        case 16: hipLaunchKernelGGL((sa3_partial_kernel<16, CL>), grid, blk, lds, st, a); break;
        case 17: hipLaunchKernelGGL((sa3_partial_kernel<17, CL>), grid, blk, lds, st, a); break;
        case 24: hipLaunchKernelGGL((sa3_partial_kernel<24, CL>), grid, blk, lds, st, a); break;
        default: hipLaunchKernelGGL((sa3_partial_kernel<32, CL>), grid, blk, lds, st, a); break;
    }
    LT_CHECK_LAUNCH("lt_softargmax3d_fwd(partial)");
    return LT_OK;
}

// ---- 2D soft-argmax: one workgroup per heatmap ------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(v, off);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

__global__ __launch_bounds__(256) void sa2_kernel(const float* __restrict__ hm, float mult, int softmax, float* __restrict__ coords,
                                                   float* __restrict__ probs, int h, int w) {
    __shared__ float red[4];
    const long long base = (long long)blockIdx.x * h * w;
    const int n = h * w;
    float M = 0.f;
    if (softmax) {
        float m = -INFINITY;
        for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, mult * hm[base + i]);
        M = block_reduce(m, red, true);
    }
    float s = 0.f, sx = 0.f, sy = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float x = mult * hm[base + i];
        const float e = softmax ? expf(x - M) : fmaxf(x, 0.f);
        const int yy = i / w, xx = i - yy * w;
        s += e; sx += e * (float)xx; sy += e * (float)yy;
    }
    const float S = block_reduce(s, red, false);
    const float X = block_reduce(sx, red, false);
    const float Y = block_reduce(sy, red, false);
    if (threadIdx.x == 0) {
        // softmax: sum p*x with p = e/S; relu: (sum e*x) / (sum e)  (op.py:39-44) -- the same quotient
        coords[blockIdx.x * 2] = X / S;
        coords[blockIdx.x * 2 + 1] = Y / S;
    }
    if (probs)
        for (int i = threadIdx.x; i < n; i += 256) {
            const float x = mult * hm[base + i];
            probs[base + i] = softmax ? __fdiv_rn(expf(x - M), S) : fmaxf(x, 0.f);
        }
}

// ---- DLT: one lane per (sample, joint) ----------------------------------------------------------
__global__ void dlt_kernel(const float* __restrict__ proj, const float* __restrict__ pts, const float* __restrict__ conf,
                           float* __restrict__ out, int B, int NV, int J) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B * J) return;
    const int b = g / J, j = g - b * J;
    double Mx[4][4] = {};
    for (int v = 0; v < NV; ++v) {  // rows of A, multiview.py:159-161, accumulated into A^T A
        const float* P = proj + ((long long)b * NV + v) * 12;
        const float* p = pts + (((long long)b * NV + v) * J + j) * 2;
        const float c = conf ? conf[((long long)b * NV + v) * J + j] : 1.f;
        for (int r = 0; r < 2; ++r) {
            float arow[4];
            for (int k = 0; k < 4; ++k) arow[k] = (P[8 + k] * p[r] - P[4 * r + k]) * c;  // fp32 like the reference
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 4; ++k) Mx[i][k] += (double)arow[i] * (double)arow[k];
        }
    }
    // cyclic Jacobi on the symmetric 4x4; eigenvector of the smallest eigenvalue = last right singular vector
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) off += Mx[p][q] * Mx[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (Mx[p][q] == 0.0) continue;
                const double theta = (Mx[q][q] - Mx[p][p]) / (2.0 * Mx[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < 4; ++k) {
                    const double akp = Mx[k][p], akq = Mx[k][q];
                    Mx[k][p] = cs * akp - sn * akq; Mx[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = Mx[p][k], aqk = Mx[q][k];
                    Mx[p][k] = cs * apk - sn * aqk; Mx[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (Mx[i][i] < Mx[best][best]) best = i;
    const double wv = V[3][best];
    float* o = out + (long long)g * 3;
    o[0] = (float)(V[0][best] / wv); o[1] = (float)(V[1][best] / wv); o[2] = (float)(V[2][best] / wv);
}

// ---- backward of the 2D soft-argmax and of the DLT (training of the algebraic model, train.py:189-236) ---------------------------------------
// integrate_tensor_2d (op.py:11-47), softmax mode: p = softmax_i(mult * h_i), (X, Y) = sum_i p_i (x_i, y_i):
//   d L / d h_i = mult * p_i * ((x_i - X) g_x + (y_i - Y) g_y).  One elementwise pass over the returned heatmaps p.
// ReLU mode (heatmap_softmax: false): e_i = relu(mult h_i) (the returned heatmaps), (X, Y) = sum_i e_i (x_i, y_i) / S, S = sum_i e_i:
//   d L / d h_i = mult [e_i > 0] ((x_i - X) g_x + (y_i - Y) g_y) / S   (every workgroup re-adds S: h x w is a few thousand values).
__global__ __launch_bounds__(256) void sa2_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ coords, const float* __restrict__ gcoords,
                                                       float mult, int softmax, float* __restrict__ ghm, int h, int w) {
    __shared__ float red[4];
    const long long base = (long long)blockIdx.y * h * w;
    const float X = coords[blockIdx.y * 2], Y = coords[blockIdx.y * 2 + 1];
    const float gx = gcoords[blockIdx.y * 2], gy = gcoords[blockIdx.y * 2 + 1];
    const int n = h * w;
    float invS = 1.f;
    if (!softmax) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += probs[base + i];
        invS = 1.f / block_reduce(s, red, false);
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int yy = i / w, xx = i - yy * w;
        const float pi = probs[base + i];
        const float wgt = softmax ? pi : (pi > 0.f ? invS : 0.f);
        ghm[base + i] = mult * wgt * (((float)xx - X) * gx + ((float)yy - Y) * gy);
    }
}

// triangulate_point_from_multiple_views_linear_torch (multiview.py:141-168): X = v[:3] / v[3], v = right singular vector of A for its smallest
// singular value = eigenvector of M = A^T A for its smallest eigenvalue lambda; rows of A: c_v (x_{v,r} P_v[2,:] - P_v[r,:]).  What autograd derives
// through torch.svd there, written out:  g_v = (g_X / v3, -(g_X . v[:3]) / v3^2) (orthogonal to v: X does not depend on |v|),
//   w = (lambda I - M)^+ g_v = sum_{i != min} e_i (e_i . g_v) / (lambda - lambda_i),   dL/dM = (w v^T + v w^T) / 2,   dL/dA = A (w v^T + v w^T),
//   dL/dc_v = sum_r dL/dA[v,r,:] . (x P2 - Pr),   dL/dx_{v,r} = c_v dL/dA[v,r,:] . P_v[2,:].   fp64, one lane per (sample, joint).
__global__ void dlt_bwd_kernel(const float* __restrict__ proj, const float* __restrict__ pts, const float* __restrict__ conf,
                               const float* __restrict__ gout, float* __restrict__ gpts, float* __restrict__ gconf, int B, int NV, int J) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B * J) return;
    const int b = g / J, j = g - b * J;
    double Mx[4][4] = {};
    for (int v = 0; v < NV; ++v) {
        const float* P = proj + ((long long)b * NV + v) * 12;
        const float* p = pts + (((long long)b * NV + v) * J + j) * 2;
        const float c = conf ? conf[((long long)b * NV + v) * J + j] : 1.f;
        for (int r = 0; r < 2; ++r) {
            float arow[4];
            for (int k = 0; k < 4; ++k) arow[k] = (P[8 + k] * p[r] - P[4 * r + k]) * c;
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 4; ++k) Mx[i][k] += (double)arow[i] * (double)arow[k];
        }
    }
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 16; ++sweep) {          // the forward's cyclic Jacobi
        double off = 0;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) off += Mx[p][q] * Mx[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (Mx[p][q] == 0.0) continue;
                const double theta = (Mx[q][q] - Mx[p][p]) / (2.0 * Mx[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < 4; ++k) { const double akp = Mx[k][p], akq = Mx[k][q]; Mx[k][p] = cs * akp - sn * akq; Mx[k][q] = sn * akp + cs * akq; }
                for (int k = 0; k < 4; ++k) { const double apk = Mx[p][k], aqk = Mx[q][k]; Mx[p][k] = cs * apk - sn * aqk; Mx[q][k] = sn * apk + cs * aqk; }
                for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq; }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (Mx[i][i] < Mx[best][best]) best = i;
    const double lam = Mx[best][best];
    double vv[4], gv[4], wv[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) vv[k] = V[k][best];
    const double gX[3] = {(double)gout[g * 3], (double)gout[g * 3 + 1], (double)gout[g * 3 + 2]};
    gv[0] = gX[0] / vv[3]; gv[1] = gX[1] / vv[3]; gv[2] = gX[2] / vv[3];
    gv[3] = -(gX[0] * vv[0] + gX[1] * vv[1] + gX[2] * vv[2]) / (vv[3] * vv[3]);
    for (int i = 0; i < 4; ++i) {
        if (i == best) continue;
        double dot = 0;
        for (int k = 0; k < 4; ++k) dot += V[k][i] * gv[k];
        const double f = dot / (lam - Mx[i][i]);
        for (int k = 0; k < 4; ++k) wv[k] += V[k][i] * f;
    }
    for (int v = 0; v < NV; ++v) {
        const float* P = proj + ((long long)b * NV + v) * 12;
        const long long pi = ((long long)b * NV + v) * J + j;
        const float* p = pts + pi * 2;
        const double c = conf ? (double)conf[pi] : 1.0;
        double gc = 0;
        for (int r = 0; r < 2; ++r) {
            double raw[4], arow[4], av = 0, aw = 0;
            for (int k = 0; k < 4; ++k) { raw[k] = (double)(P[8 + k] * p[r] - P[4 * r + k]); arow[k] = raw[k] * c; av += arow[k] * vv[k]; aw += arow[k] * wv[k]; }
            double gp = 0;
            for (int k = 0; k < 4; ++k) {
                const double ga = aw * vv[k] + av * wv[k];          // dL/dA[v,r,k] = (A (w v^T + v w^T))[row][k]
                gc += ga * raw[k];
                gp += ga * (double)P[8 + k];
            }
            gpts[pi * 2 + r] = (float)(gp * c);
        }
        if (gconf) gconf[pi] = (float)gc;
    }
}

}  // namespace

extern "C" int lt_softargmax2d_bwd(const float* probs, const float* coords, const float* grad_coords, float mult, int32_t softmax, float* grad_heatmaps,
                                   int32_t NJ, int32_t h, int32_t w, void* stream) {
    LT_REQUIRE(probs && coords && grad_coords && grad_heatmaps && NJ >= 1 && h >= 1 && w >= 1, LT_ERR_INVALID, "lt_softargmax2d_bwd: bad argument");
    const int bx = (int)((h * w + 255) / 256);
    hipLaunchKernelGGL(sa2_bwd_kernel, dim3(bx < 64 ? bx : 64, NJ), dim3(256), 0, (hipStream_t)stream, probs, coords, grad_coords, mult, softmax, grad_heatmaps, h, w);
    LT_CHECK_LAUNCH("lt_softargmax2d_bwd");
    return LT_OK;
}

extern "C" int lt_triangulate_dlt_bwd(const float* proj, const float* points, const float* conf, const float* grad_out, float* grad_points, float* grad_conf,
                                      int32_t B, int32_t NV, int32_t J, void* stream) {
    LT_REQUIRE(proj && points && grad_out && grad_points && B >= 1 && NV >= 2 && J >= 1, LT_ERR_INVALID, "lt_triangulate_dlt_bwd: bad argument");
    hipLaunchKernelGGL(dlt_bwd_kernel, dim3((B * J + 63) / 64), dim3(64), 0, (hipStream_t)stream, proj, points, conf, grad_out, grad_points, grad_conf, B, NV, J);
    LT_CHECK_LAUNCH("lt_triangulate_dlt_bwd");
    return LT_OK;
}

extern "C" size_t lt_softargmax3d_workspace(int32_t B, int32_t J, int64_t nvox) {
    const int64_t nchunks = (nvox + SA_CHUNK - 1) / SA_CHUNK;
    return (size_t)((int64_t)B * J * nchunks * SA_REC + (int64_t)B * J * 2) * sizeof(float);
}

extern "C" int lt_softargmax3d_fwd(const float* logits, const float* coords, float mult, int32_t softmax, int32_t channels_last,
                                   int32_t ld, float* kp, float* probs, int32_t B, int32_t J, int64_t nvox, void* workspace,
                                   void* stream) {
    LT_REQUIRE(logits && coords && kp && workspace, LT_ERR_INVALID, "lt_softargmax3d_fwd: null argument");
    LT_REQUIRE(B >= 1 && J >= 1 && nvox >= 1, LT_ERR_INVALID, "lt_softargmax3d_fwd: bad shape");
    LT_REQUIRE(J <= 32, LT_ERR_UNSUPPORTED, "lt_softargmax3d_fwd: J=%d > 32 (split the joints into groups of <= 32)", J);
    LT_REQUIRE(!channels_last || ld >= J, LT_ERR_INVALID, "lt_softargmax3d_fwd: ld < J");
    LT_REQUIRE(!channels_last || ld == J, LT_ERR_UNSUPPORTED, "lt_softargmax3d_fwd: channels-last logits must be dense (ld == J)");
    SA3Args a;
    a.logits = logits; a.coords = coords; a.kp = kp; a.probs = probs; a.mult = mult; a.softmax = softmax; a.J = J; a.ld = ld; a.nvox = nvox;
    a.nchunks = (int)((nvox + SA_CHUNK - 1) / SA_CHUNK);
    a.partial = (float*)workspace;
    a.stats = a.partial + (long long)B * J * a.nchunks * SA_REC;   // the workspace layout is sized for SA_CHUNK chunks
    hipStream_t st = (hipStream_t)stream;
    const bool vec = !channels_last && sa3_planar_vec(a);
    int rc = LT_OK;
    if (vec) {
        a.nchunks = (int)((nvox + SAP_CHUNK - 1) / SAP_CHUNK);    // fewer, larger chunks: the records fit a fortiori
        hipLaunchKernelGGL(sa3_partial_planar_kernel, dim3(a.nchunks, B), dim3(256), 0, st, a);
        LT_CHECK_LAUNCH("lt_softargmax3d_fwd(partial, planar)");
    } else {
        rc = channels_last ? sa3_launch_partial<true>(a, B, st) : sa3_launch_partial<false>(a, B, st);
    }
    if (rc != LT_OK) return rc;
    hipLaunchKernelGGL(sa3_finalize_kernel, dim3(B * J), dim3(64), 0, st, a, B);
    LT_CHECK_LAUNCH("lt_softargmax3d_fwd(finalize)");
    if (probs) {
        const dim3 grid(a.nchunks, B);
        if (channels_last) hipLaunchKernelGGL(sa3_probs_kernel<true>, grid, dim3(256), 256 * (J | 1) * sizeof(float), st, a);
        else if (vec)
            hipLaunchKernelGGL(sa3_probs_planar_kernel, dim3((unsigned)((nvox / 4 + 1023) / 1024), B * J), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(sa3_probs_kernel<false>, grid, dim3(256), 0, st, a);
        LT_CHECK_LAUNCH("lt_softargmax3d_fwd(probs)");
    }
    return LT_OK;
}

extern "C" int lt_softargmax2d_fwd(const float* heatmaps, float mult, int32_t softmax, float* coords, float* probs, int32_t NJ,
                                   int32_t h, int32_t w, void* stream) {
    LT_REQUIRE(heatmaps && coords && NJ >= 1 && h >= 1 && w >= 1, LT_ERR_INVALID, "lt_softargmax2d_fwd: bad argument");
    hipLaunchKernelGGL(sa2_kernel, dim3(NJ), dim3(256), 0, (hipStream_t)stream, heatmaps, mult, softmax, coords, probs, h, w);
    LT_CHECK_LAUNCH("lt_softargmax2d_fwd");
    return LT_OK;
}

extern "C" int lt_triangulate_dlt(const float* proj, const float* points, const float* conf, float* out, int32_t B, int32_t NV,
                                  int32_t J, void* stream) {
    LT_REQUIRE(proj && points && out && B >= 1 && NV >= 2 && J >= 1, LT_ERR_INVALID, "lt_triangulate_dlt: bad argument");
    hipLaunchKernelGGL(dlt_kernel, dim3((B * J + 63) / 64), dim3(64), 0, (hipStream_t)stream, proj, points, conf, out, B, NV, J);
    LT_CHECK_LAUNCH("lt_triangulate_dlt");
    return LT_OK;
}
