// Column sums over a rows x C fp32 matrix (channels-last activations: BatchNorm statistics, BatchNorm backward sums, bias gradients).
// Two deterministic stages:
//   colsum_partial   grid (nslab, column blocks of 1024 channels): a workgroup owns a slab of rows; its 256 threads are RL row lanes x
//                    CW4 float4 channel lanes (consecutive threads read consecutive 16-byte vectors of one row); every thread keeps
//                    four rows in flight, accumulates in fp64, the row lanes are combined through LDS, one fp64 partial per
//                    (slab, channel, quantity) goes to the workspace;
//   colsum_finalize  64 slab lanes x 4 channels per workgroup add the partials in a fixed order and hand the NQ totals to a functor.
// The slab count keeps >= 8 rows per thread, <= 2048 workgroups and <= 8 MB of partials per quantity pair.
#pragma once
#include "lt_common.h"

namespace lt {

struct ColsumPlan { int nslab, ncb, cw4, rl; };

inline ColsumPlan colsum_plan(long long rows, int C) {
    ColsumPlan p;
    const int c4 = C / 4;
    p.cw4 = c4 < 256 ? c4 : 256;
    p.rl = 256 / p.cw4;
    p.ncb = (c4 + 255) / 256;
    long long n = rows / ((long long)p.rl * 8);
    // (round 4: 512K / C slabs -- 256 workgroups of four waves with four 8-byte loads per lane in flight gave the 18432 x 1024 bf16 layers
    //  1.2 TB/s: the pass is bound by bytes in flight, not by bytes)
    const long long cap_blocks = 2048 / p.ncb, cap_part = (512ll << 10) / C;
    if (n > cap_blocks) n = cap_blocks;
    if (n > cap_part) n = cap_part;
    if (n < 1) n = 1;
    p.nslab = (int)n;
    return p;
}

// the vector path: C a multiple of 4 with C/4 a power of two up to 256, or a multiple of 1024
inline bool colsum_fast(int C) {
    if (C < 4 || C % 4) return false;
    const int c4 = C / 4;
    return c4 <= 256 ? (c4 & (c4 - 1)) == 0 : c4 % 256 == 0;
}

inline size_t colsum_workspace(long long rows, int C, int nq) {
    return colsum_fast(C) ? (size_t)colsum_plan(rows, C).nslab * C * nq * sizeof(double) : 0;
}

// Load: void prepare(int c) -- once per thread, c = the first of its four channels (per-channel constants into registers);
//       typedef Raw; Raw fetch(long long row, int c) -- the MEMORY READS of one row's four channels, raw bits (8 bytes per bf16 tensor, 16 per fp32 one);
//       void eval(const Raw&, int c, float (&q)[NQ][4]) -- the NQ quantities of the four channels from them;  static constexpr int ROWS -- rows in flight.
// fetch / eval are separate so that ROWS rows of raw bits are in flight and evaluated one at a time: holding the EVALUATED quantities of eight
// rows (round 4's first version) cost the BatchNorm-backward reduction 211 VGPRs = two waves per SIMD, and it ran at a third of the bytes per second of
// the apply pass next to it (53 vs 26 us per layer for 6 vs 10 bytes per element).
template <int NQ, class Load>
__device__ __forceinline__ void colsum_partial(long long rows, int C, int nslab, int cw4, int rl_n, double* __restrict__ part, const Load& load_in) {
    __shared__ double red[256][NQ * 4 + 1];
    const int slab = blockIdx.x;
    const long long r0 = rows * slab / nslab, r1 = rows * (slab + 1) / nslab;
    const int rl = threadIdx.x / cw4, cv = threadIdx.x - rl * cw4;
    const int c = (blockIdx.y * 256 + cv) * 4;
    double acc[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = 0.0;
    Load load = load_in;
    if (c < C) {
        load.prepare(c);
        long long r = r0 + rl;
        constexpr int ROWS = Load::ROWS;
        for (; r + (long long)(ROWS - 1) * rl_n < r1; r += (long long)ROWS * rl_n) {          // ROWS rows in flight (four left the pass latency-bound at ~1.2 TB/s)
            typename Load::Raw raw[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; ++u) raw[u] = load.fetch(r + (long long)u * rl_n, c);
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                float v[NQ][4];
                load.eval(raw[u], c, v);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q][e] += (double)v[q][e];
            }
        }
        for (; r < r1; r += rl_n) {
            float v[NQ][4];
            load.eval(load.fetch(r, c), c, v);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q][e] += (double)v[q][e];
        }
    }
    if (rl_n > 1) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[threadIdx.x][q * 4 + e] = acc[q][e];
        __syncthreads();
        if (rl == 0)
            for (int k = 1; k < rl_n; ++k)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q][e] += red[k * cw4 + cv][q * 4 + e];
    }
    if (rl == 0 && c < C)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < NQ; ++q) part[((long long)slab * C + c + e) * NQ + q] = acc[q][e];
}

// Fin: void operator()(int c, const double (&tot)[NQ]).  64 slab lanes x COLSUM_FIN_C channels per workgroup: with 16 slab lanes a thread walked up to 64
// partials one dependent load after the other -- 16 us per launch for a few KB, 6.6 ms of a training step over its ~420 finalize launches.
constexpr int COLSUM_FIN_C = 4;
template <int NQ, class Fin>
__device__ __forceinline__ void colsum_finalize(const double* __restrict__ part, int C, int nslab, const Fin& fin) {
    __shared__ double red[64][COLSUM_FIN_C][NQ];
    const int kl = threadIdx.x / COLSUM_FIN_C, cl = threadIdx.x % COLSUM_FIN_C;
    const int c = blockIdx.x * COLSUM_FIN_C + cl;
    double tot[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) tot[q] = 0.0;
    if (c < C) {
        int k = kl;
        for (; k + 192 < nslab; k += 256) {          // four partials in flight
            double v[4][NQ];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) v[u][q] = part[((long long)(k + 64 * u) * C + c) * NQ + q];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) tot[q] += v[u][q];
        }
        for (; k < nslab; k += 64)
#pragma unroll
            for (int q = 0; q < NQ; ++q) tot[q] += part[((long long)k * C + c) * NQ + q];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) red[kl][cl][q] = tot[q];
    __syncthreads();
    if (kl == 0 && c < C) {
        for (int k = 1; k < 64; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) tot[q] += red[k][cl][q];
        fin(c, tot);
    }
}

}  // namespace lt
