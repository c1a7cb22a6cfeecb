// Implicit-GEMM convolution, third generation: 288 x BN tile, eight waves, three-stage LDS-DMA ring, skewed DMA issue.
//
// Why (measured on MI355X with the shader-clock build of conv_igemm2, tools/trace_kstep.py, ResNet-152 layer3 at 64 images):
//   * a 128x128 tile needs 64 bytes of A/B per CU clock at the MFMA peak = exactly what the vector-memory path can deliver;
//     its K step cost 2130 cycles = 1080 (DMA issue: eight waves of two co-resident workgroups queue on the same TA, the
//     wave is stuck in the issue for that long) + 860 (fragment reads + MFMAs), the two phases in lock step, not overlapped;
//   * M = 36864 rows give 576 tiles of 128x128 for 512 workgroup slots: a second round at 12 % occupancy.
// Here:
//   * BM = 288 = 2 x 144 rows: every ResNet level of the 384x384 input is a multiple of 144 pixels per image (24^2 = 4 x 144,
//     48^2, 96^2, 12^2 = 144), so layer3 at 64 images is 128 x (Cout/128) tiles = one workgroup per CU, rounds are exact;
//   * one workgroup per CU, eight waves = 2 (M) x 4 (N), wave tile 144 x BN/4 on the 16x16x32 MFMA (9 x 2 accumulator tiles);
//     288x128 needs 46 B/clk at the MFMA peak (below the 64 B/clk of the load path);
//   * three stages: a stage is requested two K steps before it is needed, so WHEN inside the step it is requested is free:
//     every wave issues one DMA piece behind every second A fragment (a piece holds the wave in the issue stage for ~100
//     cycles; the other wave of its SIMD feeds the matrix pipe meanwhile);
//   * fragments: hand-issued ds_read_b128 with counted lgkmcnt (hipcc falls back to lgkmcnt(0) beyond one group in flight).
// bf16 only (the fp32 parity mode stays on conv_igemm2), one phase, pointwise or uniform-tap addressing, vector epilogue.
//
// Kernels in this file (dispatch: conv3_try at the end): conv_igemm3_kernel (288 x 128 / 64, 8 or 12 waves),
// conv_igemm5_kernel (288 x 256, 32-element K steps, both operands staged),
// conv_igemm6_kernel (288 x 256 or 144 x 256, weights from global memory in fragment order: the default for Cout % 256 == 0).
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

using namespace lt;

namespace {

__device__ uint4 g_zero_page3[2];

#ifdef LT_TRACE
__device__ long long g_trace3[8 * 1024];
#define LT_CLK3() ((long long)__builtin_amdgcn_s_memtime())
#endif

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16(const void* src, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_base)
        : "memory");
}

__device__ __forceinline__ void wait_vmcnt3(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // conservative
    }
}

template <int IMM>
__device__ __forceinline__ void lds_read16(V16& d, unsigned addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds_read offset field");
    f32x4 t;   // a native vector (HIP's uint4 is a struct, which inline asm can only take indirectly)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(IMM));
    d.f = t;
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}
__device__ __forceinline__ void frag_ready(V16& f) {
    f32x4 t = f.f;
    asm volatile("" : "+v"(t));
    f.f = t;
}
template <int I0, int I1, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

constexpr int BM3 = 288;

#ifdef LT_ABL_NO_MMA
#define LT3_MMA(c_, a_, b_) (void)0
#else
#define LT3_MMA(c_, a_, b_) Mma<T, MF>::run(c_, a_, b_)
#endif

// read stream of one K step (see the kernel): position of A fragment u, and how far the stream must have been issued before
// fragment u is waited for (LOOK entries of lookahead)
constexpr int s3_pos(int u, int SM, int SN) { return (u / SM) * (SM + SN) + SN + u % SM; }
constexpr int s3_target(int u, int SM, int SN, int LOOK, int TOTAL) {
    return u < 0 ? 0 : (s3_pos(u, SM, SN) + 1 + LOOK < TOTAL ? s3_pos(u, SM, SN) + 1 + LOOK : TOTAL);
}

template <int BN, int MODE, int NWM>
__global__ __launch_bounds__(256 * NWM) void conv_igemm3_kernel(const ConvArgs a) {
    typedef bf16_t T;
    constexpr bool PW = MODE == 1;   // pointwise: rows contiguous, no taps; else uniform tap (Cin*2 % 128 == 0)
    // NWM waves along M x 4 along N: 2 -> eight waves of 144 x BN/4, 3 -> twelve waves of 96 x BN/4 (three per SIMD)
    constexpr int BM = BM3, NW = 4 * NWM, WM = BM / NWM, WN = BN / 4, MF = 16, SM = WM / MF, SN = WN / MF, G = 2, NST = 3, VEC = 8, BK = 64;
    constexpr int NPA = BM / 8, NPB = BN / 8;            // 1 KiB DMA pieces per stage (8 rows of 128 B each)
    constexpr int A_IT = (NPA + NW - 1) / NW;            // piece wave + NW i, valid while < NPA (wave-uniform)
    constexpr int B_IT = (NPB + NW - 1) / NW;
    constexpr int NPASS = WM / 48;                       // epilogue passes of 48 rows
    constexpr int STAGE = (BM + BN) * ROW_BYTES;
    constexpr int REGION = NST * STAGE;
    constexpr int EP_ROWS = 48, EP_LD = WN + 4, EP_WAVE = EP_ROWS * EP_LD * 4;   // per-wave fp32 staging, three passes of 48 rows
    static_assert(SN >= 1 && NW * EP_WAVE <= REGION && WM == NPASS * EP_ROWS && (NPASS == 2 || NPASS == 3), "tile shape");
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int4* s_taps = (int4*)(smem + REGION);               // [ntaps] (unused when PW)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;

    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page3;
    asm volatile("" : "+s"(zp_bits));                    // opaque: keeps the address in SGPRs instead of a GOT load per use
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // XCD-aware tile order (see conv_igemm2.hip): every XCD walks one contiguous run of the (tile_m, tile_n) raster
    int lin = blockIdx.x;
    if (!(a.flags & LT_EPI_NO_XCD_REMAP)) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = lin & 7, j = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_n = lin % a.tiles_n;
    const int tile_m = lin / a.tiles_n;
    const PhaseArg ph = a.phase[0];
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ x = (const T*)a.x;
    const T* __restrict__ w = (const T*)ph.w;

    // DMA ownership: thread t fills physical slot t&7 of rows (t>>3) + 8 NW i; the logical K vector it must fetch is
    // v = (t&7) ^ ((row>>1)&7), the same for all its rows (8 NW i = 64 i or 96 i does not change (row>>1)&7)
    const int v = (t & 7) ^ ((t >> 4) & 7);
    // the last round of A / B pieces may be partial: only waves < NP % NW own a piece in it (wave-uniform)
    const bool a_tail = (NPA % NW == 0) || wave < NPA % NW;
    const bool b_tail = (NPB % NW == 0) || wave < NPB % NW;
    const int na = A_IT - (a_tail ? 0 : 1), nbp = B_IT - (b_tail ? 0 : 1);
    // -DLT_ABL_*: timing ablations for profiling builds (lt_build.build_variant); results are WRONG with any of them
#ifdef LT_ABL_NO_A
    constexpr bool ABL_A = true;
#else
    constexpr bool ABL_A = false;
#endif
#ifdef LT_ABL_NO_B
    constexpr bool ABL_B = true;
#else
    constexpr bool ABL_B = false;
#endif
    const int dps = (ABL_A ? 0 : na) + (ABL_B ? 0 : nbp);   // DMA pieces of this wave per stage
    int id0[PW ? 1 : A_IT], ih0[PW ? 1 : A_IT], iw0[PW ? 1 : A_IT], baseC[A_IT];
    if (PW) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + (t >> 3) + 8 * NW * i;
            baseC[i] = (m < a.M && i < na) ? m * a.Cin + v * VEC : -1;
        }
    } else {
        for (int i = t; i < ph.ntaps; i += 64 * NW) s_taps[i] = ph.taps[i];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + (t >> 3) + 8 * NW * i;
            if (m < a.M && i < na) {
                int n, od, oh, ow;
                decode_row(a, m, n, od, oh, ow);
                id0[i] = od * a.sd - a.pd;
                ih0[i] = oh * a.sh - a.ph;
                iw0[i] = ow * a.sw - a.pw;
                baseC[i] = (((n * a.D + id0[i]) * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
            } else {
                id0[i] = -(1 << 24);
                ih0[i] = iw0[i] = baseC[i] = 0;
            }
        }
    }
    int cur[PW ? 1 : A_IT];   // element offset of (row, current tap, this lane's vector) or -1 when the tap is out of the image
    const T* wrow[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) wrow[j] = w + (size_t)(n0 + (t >> 3) + 8 * NW * j) * a.k_pad + v * VEC;
    __syncthreads();

    const int nk = a.k_pad / BK;

    // ---- output rows of this lane in the epilogue: pass p (48 rows of the wave tile), iteration k ----
    constexpr int LPR = WN / 8, RPP = 64 / LPR, ITP = (EP_ROWS + RPP - 1) / RPP, NRES = NPASS * ITP;   // <= 9
    static_assert(NRES <= 9, "named residual registers");
    const int colv = n0 + wn * WN + (lane % LPR) * 8;
    auto out_off = [&](int p, int k) -> long long {   // element offset of the lane's 8-channel vector, or -1
        const int rr = k * RPP + lane / LPR;
        const int m = m0 + wm * WM + p * EP_ROWS + rr;
        if (rr >= EP_ROWS || m >= a.M || colv >= a.Cout) return -1;
        long long pix = m;
        if (!PW) {
            int n, od, oh, ow;
            decode_row(a, m, n, od, oh, ow);
            pix = ((long long)(n * a.OD + od * a.osd + ph.ood) * a.OH + oh * a.osh + ph.ooh) * a.OW + ow * a.osw + ph.oow;
        }
        return pix * a.ldc + colv;
    };
    // residual vectors requested before the K loop, in named registers (an array stayed in scratch memory, see conv_igemm2.hip)
    uint4 rp0, rp1, rp2, rp3, rp4, rp5, rp6, rp7, rp8;
    rp0 = rp1 = rp2 = rp3 = rp4 = rp5 = rp6 = rp7 = rp8 = make_uint4(0, 0, 0, 0);
    const bool has_res = a.res != nullptr;
    if (has_res) {
        auto pf = [&](int idx) -> uint4 {
            const long long o = out_off(idx / ITP, idx % ITP);
            const void* src = o >= 0 ? (const void*)((const T*)a.res + o) : zero_page;
            return *(const uint4*)src;
        };
        if (NRES > 0) rp0 = pf(0);
        if (NRES > 1) rp1 = pf(1);
        if (NRES > 2) rp2 = pf(2);
        if (NRES > 3) rp3 = pf(3);
        if (NRES > 4) rp4 = pf(4);
        if (NRES > 5) rp5 = pf(5);
        if (NRES > 6) rp6 = pf(6);
        if (NRES > 7) rp7 = pf(7);
        if (NRES > 8) rp8 = pf(8);
    }

    // ---- DMA of this wave's pieces of one stage: stage_prep (tap bookkeeping), then pieces 0..NPIECE-1 (A rounds, then B) ----
    constexpr int NPIECE = A_IT + B_IT;
    int c0s = 0;                                         // channel offset of the stage being requested (uniform-tap mode)
    auto stage_prep = [&](int ks) {
        if (!PW) {
            const int k0 = ks * BK;                      // wave-uniform
            c0s = k0 & (a.Cin - 1);
            if (c0s == 0) {                              // the tap (and with it the bounds test) changes every Cin/BK steps
                const int tap = k0 >> a.log2Cin;
                int4 tp = make_int4(-(1 << 24), 0, 0, 0);
                if (tap < ph.ntaps) tp = s_taps[tap];
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    const int id = id0[i] + tp.x, ih = ih0[i] + tp.y, iw = iw0[i] + tp.z;
                    const bool ok = ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                    cur[i] = ok ? baseC[i] + tp.w + v * VEC : -1;
                }
            }
        }
    };
    auto stage_piece = [&](int ks, int buf, auto pc) {
        constexpr int P = decltype(pc)::value;
        const unsigned sA = lds0 + buf * STAGE;
        if constexpr (P < A_IT) {
            if (ABL_A || (P == A_IT - 1 && !a_tail)) return;
            const void* src;
            if (PW) src = baseC[P] >= 0 ? (const void*)(x + (baseC[P] + ks * BK)) : zero_page;
            else src = cur[P] >= 0 ? (const void*)(x + (cur[P] + c0s)) : zero_page;
            dma16(src, sA + (wave + NW * P) * 1024);
        } else {
            constexpr int j = P - A_IT;
            if (ABL_B || (j == B_IT - 1 && !b_tail)) return;
            dma16(wrow[j] + ks * BK, sA + BM * ROW_BYTES + (wave + NW * j) * 1024);
        }
    };
    auto stage = [&](int ks, int buf) {
        stage_prep(ks);
        static_for<0, NPIECE>([&](auto pc) { stage_piece(ks, buf, pc); });
    };

    // ---- fragment addresses: row r15 = lane & 15 of a 16-row MFMA tile, K vector (lane >> 4) + 4 g, slot = vector ^ ((row>>1)&7)
    // ((row>>1)&7 depends on r15 only: 144 wm, 16 i, WN wn and 16 j are all multiples of 16) ----
    const int r15 = lane & 15;
    const int fsw = (r15 >> 1) & 7;
    unsigned aoff[G], boff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const unsigned fo = r15 * ROW_BYTES + ((((lane >> 4) + 4 * g) ^ fsw) << 4);
        aoff[g] = lds0 + wm * WM * ROW_BYTES + fo;
        boff[g] = lds0 + BM * ROW_BYTES + wn * WN * ROW_BYTES + fo;
    }

    acc_t acc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    // prologue: two stages in flight
    if (0 < nk) stage(0, 0);
    if (1 < nk) stage(1, 1);

    // read stream of one K step: per group g the SN B fragments, then the SM A fragments; LOOK entries of lookahead
    constexpr int RPG = SN + SM, TOTAL = G * RPG, LOOK = 6, RA = LOOK + 1;
    static_assert(LOOK + 1 <= 15, "lgkmcnt range");

#ifdef LT_TRACE
    long long tr_vm = 0, tr_bar = 0, tr_iss = 0, tr_cmp = 0, tr_prev = 0;
    const long long tr_begin = LT_CLK3();
    const long long tr_rt0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    for (int ks = 0; ks < nk; ++ks) {
#ifdef LT_TRACE
        const long long tr0 = LT_CLK3();
        if (ks > 0) tr_cmp += tr0 - tr_prev;
#endif
        // stage ks must have landed; stage ks+1 (this wave's dps pieces, if it exists) may stay in flight
        wait_vmcnt3(ks + 1 < nk ? dps : 0);
#ifdef LT_TRACE
        const long long tr1 = LT_CLK3();
        tr_vm += tr1 - tr0;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all DMAs of stage ks landed; stage ks-1 fully consumed
#ifdef LT_TRACE
        const long long tr2 = LT_CLK3();
        tr_bar += tr2 - tr1;
#endif
        const bool more = ks + 2 < nk;
        const int nbuf = (ks + 2) % NST;
        if (more) stage_prep(ks + 2);                    // tap table read here, while no fragment read is in flight
#ifdef LT_TRACE
        const long long tr3 = LT_CLK3();
        tr_iss += tr3 - tr2;
        tr_prev = tr3;
#endif
        const unsigned sbase = (ks % NST) * STAGE;
        const unsigned abase[G] = {aoff[0] + sbase, aoff[1] + sbase};
        const unsigned bbase[G] = {boff[0] + sbase, boff[1] + sbase};
        V16 fa[RA], fb[G][SN];
        auto issue = [&](auto kc) {                      // stream entry K
            constexpr int K = decltype(kc)::value;
            constexpr int g = K / RPG, r = K % RPG;
            if constexpr (r < SN) lds_read16<r * 16 * ROW_BYTES>(fb[g][r], bbase[g]);
            else lds_read16<(r - SN) * 16 * ROW_BYTES>(fa[(g * SM + r - SN) % RA], abase[g]);
        };
        auto step_u = [&](auto uc) {                     // A fragment u = g * SM + i: bring the stream up to date, wait, 2 MFMAs
            constexpr int u = decltype(uc)::value;
            constexpr int g = u / SM, i = u % SM;
            constexpr int pos = s3_pos(u, SM, SN);
            constexpr int prev_target = s3_target(u - 1, SM, SN, LOOK, TOTAL);
            constexpr int target = s3_target(u, SM, SN, LOOK, TOTAL);
            static_for<prev_target, target>([&](auto kc) { issue(kc); });
            lgkm_wait<target - pos - 1>();
            if constexpr (i == 0) {
#pragma unroll
                for (int j = 0; j < SN; ++j) frag_ready(fb[g][j]);
            }
            frag_ready(fa[u % RA]);
#pragma unroll
            for (int j = 0; j < SN; ++j) LT3_MMA(acc[i][j], fa[u % RA], fb[g][j]);
            __builtin_amdgcn_sched_barrier(0);           // keep the MFMAs with their wait (they are not volatile and would sink)
            // one DMA piece of stage ks+2 behind every second fragment: the ~100 cycles a wave spends in a DMA issue are then
            // filled by the MFMAs of the other wave on its SIMD (three stages: the piece has two K steps to land)
            static_assert(2 * NPIECE <= G * SM, "one DMA piece behind every second fragment");
            if constexpr (u % 2 == 1 && u / 2 < NPIECE) {
                if (more) stage_piece(ks + 2, nbuf, std::integral_constant<int, u / 2>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        static_for<0, G * SM>([&](auto uc) { step_u(uc); });
    }
#ifdef LT_TRACE
    {
        const long long tr_end = LT_CLK3();
        tr_cmp += tr_end - tr_prev;
        const long long tr_rt1 = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 1024 && lane == 0) {
            long long* o = g_trace3 + (blockIdx.x >> 3) * 8;
            o[0] = tr_end - tr_begin; o[1] = tr_vm; o[2] = tr_bar; o[3] = tr_iss; o[4] = tr_cmp; o[5] = nk; o[6] = tr_rt1 - tr_rt0; o[7] = blockIdx.x;
        }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring becomes the epilogue staging area
#ifdef LT_ABL_NO_EPI
    if (a.M >= 0) return;
#endif

    // ---- epilogue: three passes of 48 rows through this wave's private fp32 LDS tile -> 16-byte vectors ----
    const EpiFloors fl = epi_floors(a.flags);
    const unsigned no_res = has_res ? 0u : 0x80008000u;   // residual registers are zero without a residual: make them -0.0 (v + -0.0 == v)
    float* ep = (float*)(smem + wave * EP_WAVE);
    float bi[SN], sc[SN], sf[SN];
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int colj = n0 + wn * WN + j * MF + r15;   // < cout_pad: the constant arrays are padded
        bi[j] = a.bias ? a.bias[colj] : 0.f;
        sc[j] = a.scale ? a.scale[colj] : 1.f;
        sf[j] = a.shift ? a.shift[colj] : 0.f;
    }
    auto row_out = [&](int p, int k, uint4 resv) {
        const long long o = out_off(p, k);
        if (o < 0) return;
        const float* src = ep + (k * RPP + lane / LPR) * EP_LD + (lane % LPR) * 8;
        const float4 q0 = *(const float4*)src, q1 = *(const float4*)(src + 4);
        const float vv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const unsigned ru[4] = {resv.x | no_res, resv.y | no_res, resv.z | no_res, resv.w | no_res};
        unsigned ou[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            ou[e] = pack_bf16x2(epi_apply(vv[2 * e], fl, __uint_as_float(ru[e] << 16)), epi_apply(vv[2 * e + 1], fl, __uint_as_float(ru[e] & 0xffff0000u)));
        *(uint4*)((T*)a.y + o) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
    };
#define LT3_PASS(P_, RA_, RB_, RC_)                                                                      \
    {                                                                                                    \
        _Pragma("unroll") for (int ii = 0; ii < 3; ++ii)                                                 \
            _Pragma("unroll") for (int j = 0; j < SN; ++j)                                               \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                            \
                    ep[(ii * MF + (lane >> 4) * 4 + e) * EP_LD + j * MF + r15] =                        \
                        (acc[(3 * P_ + ii) % SM][j][e] + bi[j]) * sc[j] + sf[j];                         \
        if (ITP > 0) row_out(P_, 0, RA_);                                                                \
        if (ITP > 1) row_out(P_, 1, RB_);                                                                \
        if (ITP > 2) row_out(P_, 2, RC_);                                                                \
    }
    if (ITP == 3) {
        LT3_PASS(0, rp0, rp1, rp2)
        LT3_PASS(1, rp3, rp4, rp5)
        if (NPASS > 2) LT3_PASS(2, rp6, rp7, rp8)
    } else {
        LT3_PASS(0, rp0, rp1, rp1)
        LT3_PASS(1, rp2, rp3, rp3)
        if (NPASS > 2) LT3_PASS(2, rp4, rp5, rp5)
    }
#undef LT3_PASS
}

// ---- 288 x 256 tile, 32-element K steps, four stages (Cout % 256 == 0) ---------------------------------------------------------
// The L2->LDS stream is what bounds the kernels above (one 52 KB stage per ~2200 cycles with two stages in flight):
// what helps is fewer staged bytes per FLOP at the same bytes in flight.  A 288 x 256 tile stages (288+256) rows per K step for
// twice the MFMA work of 288 x 128 (135 instead of 88 FLOP per byte); to keep three stages in flight in 160 KB the K step is 32
// elements: 64-byte LDS rows, 34 KB per stage, four stages.  Eight waves = 2 (M) x 4 (N), wave tile 144 x 64 (9 x 4 accumulator
// tiles), every wave issues its share of the DMA pieces sliced between its fragment steps (as in the 12-wave kernel).
// 64-byte rows: a 16-byte slot holds K vector (slot ^ g(row)), g = [0,2,3,1][(row >> 2) & 3] -- with that the four 16-lane groups
// of a ds_read_b128 fragment read (rows r, K vector lane >> 4) each touch all 64 banks once (checked by enumeration, comment in
// DESIGN.md); the DMA writes lane-linearly, so the swizzle is applied to the source address as everywhere else.
__device__ __forceinline__ int swz64(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int MODE>
__global__ __launch_bounds__(512) void conv_igemm5_kernel(const ConvArgs a) {
    typedef bf16_t T;
    constexpr bool PW = MODE == 1;
    constexpr int BM = BM3, BN = 256, NW = 8, WM = 144, WN = 64, MF = 16, SM = WM / MF, SN = WN / MF, NST = 4, VEC = 8, BK = 32, ROWB = 64;
    constexpr int NPA = BM / 16, NPB = BN / 16;          // 1 KiB DMA pieces per stage (16 rows of 64 B each): 18 + 16
    constexpr int A_IT = (NPA + NW - 1) / NW, B_IT = NPB / NW;   // 3 (waves 0,1) / 2, and 2
    static_assert(NPB % NW == 0, "B pieces divide evenly");
    constexpr int STAGE = (BM + BN) * ROWB;              // 34816 B
    constexpr int REGION = NST * STAGE;
    constexpr int EP_ROWS = 48, EP_LD = WN + 4, EP_WAVE = EP_ROWS * EP_LD * 4, NPASS = WM / EP_ROWS;
    static_assert(NW * EP_WAVE <= REGION && NPASS == 3, "epilogue staging");
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int4* s_taps = (int4*)(smem + REGION);               // [ntaps] (unused when PW)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;

    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page3;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int lin = blockIdx.x;
    if (!(a.flags & LT_EPI_NO_XCD_REMAP)) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = lin & 7, j = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_n = lin % a.tiles_n;
    const int tile_m = lin / a.tiles_n;
    const PhaseArg ph = a.phase[0];
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ x = (const T*)a.x;
    const T* __restrict__ w = (const T*)ph.w;

    // DMA ownership: piece p = wave + 8 i covers rows 16 p .. 16 p + 15; lane -> row 16 p + (lane >> 2), physical slot lane & 3,
    // logical K vector kv = slot ^ g(row), and g(row) = g(lane >> 2) because 16 p is a multiple of 16
    const int prow = lane >> 2;
    const int kv = (lane & 3) ^ swz64(prow);
    const bool a_tail = wave < NPA % NW;                 // waves 0, 1 own a third A piece
    const int dps = (A_IT - 1) + (a_tail ? 1 : 0) + B_IT;   // 5 or 4 DMA pieces per wave and stage
    int id0[PW ? 1 : A_IT], ih0[PW ? 1 : A_IT], iw0[PW ? 1 : A_IT], baseC[A_IT], cur[PW ? 1 : A_IT];
    if (!PW)
        for (int i = t; i < ph.ntaps; i += 64 * NW) s_taps[i] = ph.taps[i];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + 16 * (wave + NW * i) + prow;
        const bool own = i < A_IT - 1 || a_tail;
        if (PW) baseC[i] = (own && m < a.M) ? m * a.Cin + kv * VEC : -1;
        else if (own && m < a.M) {
            int n, od, oh, ow;
            decode_row(a, m, n, od, oh, ow);
            id0[i] = od * a.sd - a.pd;
            ih0[i] = oh * a.sh - a.ph;
            iw0[i] = ow * a.sw - a.pw;
            baseC[i] = (((n * a.D + id0[i]) * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
        } else {
            id0[i] = -(1 << 24);
            ih0[i] = iw0[i] = baseC[i] = 0;
        }
    }
    const T* wrow[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) wrow[j] = w + (size_t)(n0 + 16 * (wave + NW * j) + prow) * a.k_pad + kv * VEC;
    __syncthreads();

    const int nk = a.k_pad / BK;
    constexpr int NPIECE = A_IT + B_IT;                   // piece slots per wave and stage (the last A slot may be empty)
    int c0s = 0;
    auto stage_prep = [&](int ks) {
        if (!PW) {
            const int k0 = ks * BK;
            c0s = k0 & (a.Cin - 1);
            if (c0s == 0) {                              // the tap changes every Cin / 32 steps
                const int tap = k0 >> a.log2Cin;
                int4 tp = make_int4(-(1 << 24), 0, 0, 0);
                if (tap < ph.ntaps) tp = s_taps[tap];
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    const int id = id0[i] + tp.x, ih = ih0[i] + tp.y, iw = iw0[i] + tp.z;
                    const bool ok = ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                    cur[i] = ok ? baseC[i] + tp.w + kv * VEC : -1;
                }
            }
        }
    };
    auto stage_piece = [&](int ks, int buf, auto pc) {
        constexpr int P = decltype(pc)::value;
        const unsigned sA = lds0 + buf * STAGE;
        if constexpr (P < A_IT) {
            if (P == A_IT - 1 && !a_tail) return;
            const void* src;
            if (PW) src = baseC[P] >= 0 ? (const void*)(x + (baseC[P] + ks * BK)) : zero_page;
            else src = cur[P] >= 0 ? (const void*)(x + (cur[P] + c0s)) : zero_page;
            dma16(src, sA + (wave + NW * P) * 1024);
        } else {
            constexpr int j = P - A_IT;
            dma16(wrow[j] + ks * BK, sA + BM * ROWB + (wave + NW * j) * 1024);
        }
    };
    auto stage = [&](int ks, int buf) {
        stage_prep(ks);
        static_for<0, NPIECE>([&](auto pc) { stage_piece(ks, buf, pc); });
    };

    // fragment addresses: row r15 = lane & 15 of a 16-row tile, K vector lane >> 4 in slot (lane >> 4) ^ g(r15) (tile bases are multiples of 16)
    const int r15 = lane & 15;
    const unsigned fo = r15 * ROWB + (((lane >> 4) ^ swz64(r15)) << 4);
    const unsigned aoff = lds0 + wm * WM * ROWB + fo;
    const unsigned boff = lds0 + BM * ROWB + wn * WN * ROWB + fo;

    acc_t acc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

#pragma unroll
    for (int sgi = 0; sgi < NST - 1; ++sgi)
        if (sgi < nk) stage(sgi, sgi);

    constexpr int RPG = SN + SM, LOOK = 6, RA = LOOK + 1;   // read stream of a step: SN B fragments, then SM A fragments
    for (int ks = 0; ks < nk; ++ks) {
        // stage ks must have landed; up to NST-2 younger stages of this wave's pieces stay in flight
        int younger = nk - 1 - ks;
        if (younger > NST - 2) younger = NST - 2;
        wait_vmcnt3(younger * dps);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all DMAs of stage ks landed; stage ks-1 fully consumed
        const bool more = ks + NST - 1 < nk;
        const int nbuf = (ks + NST - 1) % NST;
        if (more) stage_prep(ks + NST - 1);
        const unsigned sbase = (ks % NST) * STAGE;
        const unsigned abase = aoff + sbase, bbase = boff + sbase;
        V16 fa[RA], fb[SN];
        auto issue = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (K < SN) lds_read16<K * 16 * ROWB>(fb[K], bbase);
            else lds_read16<(K - SN) * 16 * ROWB>(fa[(K - SN) % RA], abase);
        };
        static_for<0, SM>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int pos = SN + u;
            constexpr int prev_target = u == 0 ? 0 : (pos + LOOK < RPG ? pos + LOOK : RPG);
            constexpr int target = pos + 1 + LOOK < RPG ? pos + 1 + LOOK : RPG;
            static_for<prev_target, target>([&](auto kc) { issue(kc); });
            lgkm_wait<target - pos - 1>();
            if constexpr (u == 0) {
#pragma unroll
                for (int j = 0; j < SN; ++j) frag_ready(fb[j]);
            }
            frag_ready(fa[u % RA]);
#pragma unroll
            for (int j = 0; j < SN; ++j) LT3_MMA(acc[u][j], fa[u % RA], fb[j]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u % 2 == 0 && u / 2 < NPIECE) {   // one DMA piece of stage ks+3 behind every second fragment
                if (more) stage_piece(ks + NST - 1, nbuf, std::integral_constant<int, u / 2>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring becomes the epilogue staging area
#ifdef LT_ABL_NO_EPI
    if (a.M >= 0) return;
#endif
    // ---- epilogue: three passes of 48 rows through this wave's private fp32 LDS tile -> 16-byte vectors ----
    constexpr int LPR = WN / 8, RPP = 64 / LPR, ITP = EP_ROWS / RPP;   // 8 lanes per row, 8 rows per iteration, 6 iterations per pass
    const int colv = n0 + wn * WN + (lane % LPR) * 8;
    auto out_off = [&](int p, int k) -> long long {
        const int m = m0 + wm * WM + p * EP_ROWS + k * RPP + lane / LPR;
        if (m >= a.M || colv >= a.Cout) return -1;
        long long pix = m;
        if (!PW) {
            int n, od, oh, ow;
            decode_row(a, m, n, od, oh, ow);
            pix = ((long long)(n * a.OD + od * a.osd + ph.ood) * a.OH + oh * a.osh + ph.ooh) * a.OW + ow * a.osw + ph.oow;
        }
        return pix * a.ldc + colv;
    };
    const EpiFloors fl = epi_floors(a.flags);
    const bool has_res = a.res != nullptr;
    float* ep = (float*)(smem + wave * EP_WAVE);
    float bi[SN], sc[SN], sf[SN];
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int colj = n0 + wn * WN + j * MF + r15;    // < cout_pad: the constant arrays are padded
        bi[j] = a.bias ? a.bias[colj] : 0.f;
        sc[j] = a.scale ? a.scale[colj] : 1.f;
        sf[j] = a.shift ? a.shift[colj] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        long long off[ITP];
        uint4 rv[ITP];
#pragma unroll
        for (int k = 0; k < ITP; ++k) {                  // this pass's residual vectors first: independent round trips
            off[k] = out_off(p, k);
            rv[k] = (has_res && off[k] >= 0) ? *(const uint4*)((const T*)a.res + off[k]) : make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);
        }
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
#pragma unroll
            for (int j = 0; j < SN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ep[(ii * MF + (lane >> 4) * 4 + e) * EP_LD + j * MF + r15] = (acc[3 * p + ii][j][e] + bi[j]) * sc[j] + sf[j];
#pragma unroll
        for (int k = 0; k < ITP; ++k) {
            if (off[k] < 0) continue;
            const float* src = ep + (k * RPP + lane / LPR) * EP_LD + (lane % LPR) * 8;
            const float4 q0 = *(const float4*)src, q1 = *(const float4*)(src + 4);
            const float vv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const unsigned ru[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
            unsigned ou[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                ou[e] = pack_bf16x2(epi_apply(vv[2 * e], fl, __uint_as_float(ru[e] << 16)), epi_apply(vv[2 * e + 1], fl, __uint_as_float(ru[e] & 0xffff0000u)));
            *(uint4*)((T*)a.y + off[k]) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
        }
    }
}

// ---- 288 x 256 tile with the B operand (weights) read straight from global memory in fragment order ---------------------------
// What bounds conv_igemm5 is the number of LDS-DMA pieces a CU can push per unit of time (each piece = 16 rows of 64 bytes = 16
// separate segments for the address path): 34 per K step, 18 of activations and 16 of weights.  The weights need no gather at
// all: lt_conv_pack_weights stores them once per model in MFMA fragment order ([K / 32][Cout / 16][lane] x 16 bytes), so a wave
// reads each of its four B fragments with ONE fully coalesced 1 KB global load (L1 / L2 resident: every workgroup of the same
// N tile reads the same bytes) into registers, one K step ahead.  The ring then carries activations only: 18 pieces and 18 KB
// per stage, six stages in 108 KB, and half the LDS fragment reads.  Everything else (tile, waves, sliced DMA issue, read
// stream with counted lgkmcnt, epilogue) is conv_igemm5's.  The K loop is unrolled by two because the two B register sets
// alternate (k_pad is a multiple of 64, so the number of 32-element steps is even).
// wave-uniform base in SGPRs + 32-bit lane offset + immediate: no 64-bit address arithmetic in VGPRs
template <int IMM>
__device__ __forceinline__ void gload16(V16& d, const void* sbase, unsigned voff) {
    static_assert(IMM >= 0 && IMM < 4096, "global_load immediate offset");
    f32x4 t;
    const unsigned long long b = (unsigned long long)(size_t)sbase;
    const unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)b);   // uniform by construction; make it provable
    // s_nop: the base may have just been written by v_readfirstlane, and a VALU write of an SGPR needs 5 wait states before a
    // vector-memory instruction reads it -- the hazard recognizer does not look inside inline asm (seen: the load took the
    // stale low dword and faulted)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(t) : "v"(voff), "s"(ub), "n"(IMM) : "memory");
    d.f = t;
}
__device__ __forceinline__ void wait_vmcnt6(int n) {
    switch (n) {
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: wait_vmcnt3(n); break;
    }
}

// NWM = 2: 288-row tile, eight waves, one workgroup per CU; NWM = 1: 144-row tile, four waves, TWO workgroups per CU (55 KB ring each):
// the short-K pointwise layers are mostly epilogue traffic, and a second workgroup's K loop overlaps it.
template <int MODE, int NWM>
__global__ __launch_bounds__(256 * NWM, 2) void conv_igemm6_kernel(const ConvArgs a) {
    typedef bf16_t T;
    constexpr bool PW = MODE == 1;
    constexpr int BM = 144 * NWM, BN = 256, NW = 4 * NWM, WM = 144, WN = 64, MF = 16, SM = WM / MF, SN = WN / MF, VEC = 8, BK = 32, ROWB = 64;
#ifndef LT6_NST
#define LT6_NST 6
#endif
    constexpr int NST = LT6_NST;                          // 6 x 18 KB (8 stages measured no faster: the loop is not waiting for data)
    // Ping-pong of the two waves of a SIMD (waves w and w+4 share one: a workgroup's waves go to the SIMDs cyclically).  Waves 4-7
    // ("Y") take the per-step barrier in the MIDDLE of their MFMA sequence instead of at its top, so they run half a step behind waves
    // 0-3 ("X"): when X leaves the barrier into its step start-up (wait for its B fragments, request the next ones, first A reads) the
    // matrix pipe of the SIMD is fed by Y's second half, and Y's start-up falls into the middle of X's MFMAs.  Before, all eight waves
    // did the same thing at the same time and the pipe idled through every start-up (-DLT_ABL_NO_A -DLT_ABL_NO_B: 75 us for the 3x3
    // 256->256 layer with NO memory instruction in the loop, against 43 us of MFMA issue).  Price: Y may still read stage ks-1 when X
    // requests new pieces after barrier ks, so a piece goes to the slot of stage ks-2: NST-2 stages ahead instead of NST-1.
    // Measured (profiles/r02_ab_igemm6.txt): per layer on dense random data (tools/conv_bench.py, 128 images) 3x3 256->256
    // 106-121 -> 92-96 us, 1x1 1024->256 51 -> 46-48 us; inside the forward (post-ReLU activations, half of them zero, higher
    // clocks: the same 3x3 layer takes 91 us there WITHOUT the ping-pong) 91.1 -> 92.7 us and 1196 vs 1194 samples/s end to end --
    // the in-model kernels run against the power / clock limit, not against issue bubbles.  Default off; -DLT6_PP=1 builds it.
#ifndef LT6_PP
#define LT6_PP 0
#endif
#ifndef LT6_BPF
#define LT6_BPF 1
#endif
    constexpr bool PP = LT6_PP && NWM == 2;
    constexpr int AHEAD = PP ? NST - 2 : NST - 1;         // a piece issued during step ks belongs to stage ks + AHEAD
    constexpr int YBAR = 4;                               // Y's barrier sits behind A fragment YBAR of the 9
    static_assert(!PP || AHEAD >= LT6_BPF + 2, "ping-pong: a Y wave vouches for stage ks+1 at the top of its step ks");
    constexpr int NPA = BM / 16;                          // 18 DMA pieces of 1 KiB per stage (16 rows of 64 B each)
    constexpr int A_IT = (NPA + NW - 1) / NW;             // 3 (waves 0, 1) / 2
    constexpr int STAGE = BM * ROWB;                      // 18432 B: activations only
    constexpr int REGION = NST * STAGE;
    constexpr int EP_ROWS = 48, EP_LD = WN + 4, EP_WAVE = EP_ROWS * EP_LD * 4, NPASS = WM / EP_ROWS;
    static_assert(NW * EP_WAVE <= REGION && NPASS == 3, "epilogue staging");
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int4* s_taps = (int4*)(smem + REGION);               // [ntaps] (unused when PW)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;

    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page3;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
#ifdef LT6_STAGGER
    // A/B: the two co-resident workgroups of the four-wave variant start half a K step apart (they stay in lock step otherwise)
    if (NWM == 1 && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_sleep(LT6_STAGGER);
#endif
    int lin = blockIdx.x;
    if (!(a.flags & LT_EPI_NO_XCD_REMAP)) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = lin & 7, j = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_n = lin % a.tiles_n;
    const int tile_m = lin / a.tiles_n;
    const PhaseArg ph = a.phase[0];
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ x = (const T*)a.x;

    const int prow = lane >> 2;
    const int kv = (lane & 3) ^ swz64(prow);
    const bool a_tail = wave < NPA % NW;                 // waves 0, 1 (wave 0 of the four-wave variant) own a third piece
    const int dps = (A_IT - 1) + (a_tail ? 1 : 0);       // 3 or 2 DMA pieces per wave and stage
    int id0[PW ? 1 : A_IT], ih0[PW ? 1 : A_IT], iw0[PW ? 1 : A_IT], baseC[A_IT], cur[PW ? 1 : A_IT];
    if (!PW)
        for (int i = t; i < ph.ntaps; i += 64 * NW) s_taps[i] = ph.taps[i];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + 16 * (wave + NW * i) + prow;
        const bool own = i < A_IT - 1 || a_tail;
        if (PW) baseC[i] = (own && m < a.M) ? m * a.Cin + kv * VEC : -1;
        else if (own && m < a.M) {
            int n, od, oh, ow;
            decode_row(a, m, n, od, oh, ow);
            id0[i] = od * a.sd - a.pd;
            ih0[i] = oh * a.sh - a.ph;
            iw0[i] = ow * a.sw - a.pw;
            baseC[i] = (((n * a.D + id0[i]) * a.H + ih0[i]) * a.W + iw0[i]) * a.Cin;
        } else {
            id0[i] = -(1 << 24);
            ih0[i] = iw0[i] = baseC[i] = 0;
        }
    }
    // B fragments of this wave: N tiles n0 / 16 + 4 wn + j, K step ks -> ((ks * cout_pad / 16 + tile) * 64 + lane) * 16 bytes
    const size_t wstep = (size_t)a.tiles_n * (BN / 16) * 64 * VEC;   // elements per K step
    const T* wfrag = (const T*)ph.wfrag + (size_t)(n0 / 16 + SN * wn) * 64 * VEC;   // wave-uniform; lane l reads + 16 l bytes
    const unsigned wlane = lane * 16;
    __syncthreads();

    const int nk = a.k_pad / BK;
    int c0s = 0;
    auto stage_prep = [&](int ks) {
        if (!PW) {
            const int k0 = ks * BK;
            c0s = k0 & (a.Cin - 1);
            if (c0s == 0) {                              // the tap changes every Cin / 32 steps
                const int tap = k0 >> a.log2Cin;
                int4 tp = make_int4(-(1 << 24), 0, 0, 0);
                if (tap < ph.ntaps) tp = s_taps[tap];
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    const int id = id0[i] + tp.x, ih = ih0[i] + tp.y, iw = iw0[i] + tp.z;
                    const bool ok = ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                    cur[i] = ok ? baseC[i] + tp.w + kv * VEC : -1;
                }
            }
        }
    };
    auto stage_piece = [&](int ks, unsigned sbuf, auto pc) {
        constexpr int P = decltype(pc)::value;
        if (P == A_IT - 1 && !a_tail) return;
        const void* src;
        if (PW) src = baseC[P] >= 0 ? (const void*)(x + (baseC[P] + ks * BK)) : zero_page;
        else src = cur[P] >= 0 ? (const void*)(x + (cur[P] + c0s)) : zero_page;
        dma16(src, lds0 + sbuf + (wave + NW * P) * 1024);
    };

    const int r15 = lane & 15;
    const unsigned fo = r15 * ROWB + (((lane >> 4) ^ swz64(r15)) << 4);
    const unsigned aoff = lds0 + wm * WM * ROWB + fo;

    acc_t acc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    // The four-wave pointwise variant (the 1x1 expand layers: 8 K steps, then a tile of residual + output traffic) is HBM-bound and its
    // epilogue used to request each 48-row pass's residual vectors at the top of that pass, one exposed round trip per pass.  Now the
    // first pass's vectors are requested HERE, before the first DMA piece (oldest in the in-order vmcnt queue: every counted wait of
    // the K loop still holds), and inside the epilogue pass p + 1's are requested before pass p touches LDS.
    constexpr bool RES_EARLY = PW && NWM == 1;
    constexpr int E_LPR = WN / 8, E_RPP = 64 / E_LPR, E_ITP = 48 / E_RPP;
    const uint4 neg0 = make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);
    uint4 rv_next[RES_EARLY ? E_ITP : 1];
    auto res_load = [&](int p, uint4* rv) {
        const int colv_ = n0 + wn * WN + (lane % E_LPR) * 8;
        if (a.res != nullptr) {
#pragma unroll
            for (int k = 0; k < E_ITP; ++k) {          // unconditional loads (a lane outside the tensor reads element 0 and never uses it): a select
                const int m = m0 + wm * WM + p * 48 + k * E_RPP + lane / E_LPR;          // behind a load would make the compiler wait at the load
                const bool ok = (m < a.M) & (colv_ < a.Cout);
                {          // non-temporal: the residual (the previous block's output) is dead after this read -- it should not push the A panel and the
                    // weights out of L2 / the memory-side cache (+1.2 % end to end, in-session A/B with the plain load; a non-temporal
                    // STORE of the output gave half of that back: the next layer reads it)
                    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                    const u32x4_ t_ = __builtin_nontemporal_load((const u32x4_*)((const T*)a.res + (ok ? (long long)m * a.ldc + colv_ : 0ll)));
                    rv[k] = make_uint4(t_[0], t_[1], t_[2], t_[3]);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < E_ITP; ++k) rv[k] = neg0;
        }
    };
    if constexpr (RES_EARLY) res_load(0, rv_next);

    // B fragments in BPF + 1 register sets: set ks % (BPF + 1) holds step ks, requested BPF steps ahead.  BPF = 2 (-DLT6_BPF=2) was
    // built to test whether the in-order return of vector-memory loads (a B load is only seen once every older LDS-DMA piece of the
    // wave has landed) explains the wait in front of the barrier: per layer 110 / 104 us (BPF 1) vs 112 / 109 and 104 / 107 us (BPF 2,
    // two A-fragment lookaheads) in one session -- it does not; kept as a switch (profiles/r02_ab_igemm6.txt).
#ifndef LT6_BPF
#define LT6_BPF 1
#endif
    constexpr int BPF = LT6_BPF, NBS = BPF + 1;
    static_assert(BPF == 1 || BPF == 2, "B prefetch distance");
#ifdef LT_ABL_NO_A
    constexpr bool ABL_A = true;        // timing ablations (profiling builds, results WRONG): no DMA pieces / no B loads inside the K loop
#else
    constexpr bool ABL_A = false;
#endif
#ifdef LT_ABL_NO_B
    constexpr bool ABL_B = true;
#else
    constexpr bool ABL_B = false;
#endif
    V16 fb[NBS][SN];
    static_for<0, BPF>([&](auto pbc) {
        constexpr int pb = decltype(pbc)::value;
        const T* wpb = wfrag + (size_t)(pb < nk ? pb : nk - 1) * wstep;
        static_for<0, SN>([&](auto jc) { gload16<decltype(jc)::value * 1024>(fb[pb][decltype(jc)::value], wpb, wlane); });
    });
#pragma unroll
    for (int sgi = 0; sgi < AHEAD; ++sgi)
        if (sgi < nk) {
            stage_prep(sgi);
            static_for<0, A_IT>([&](auto pc) { stage_piece(sgi, sgi * STAGE, pc); });
        }

#ifndef LT6_LOOK
#define LT6_LOOK (LT6_BPF == 2 ? 2 : 3)
#endif
    constexpr int LOOK = LT6_LOOK, RA = LOOK + 1;        // read stream of a step: the SM A fragments (one fewer in flight next to three B sets: registers)
    unsigned rbuf = 0, wbuf = AHEAD * STAGE;              // ring offsets of the stage being read / being requested
#ifdef LT_TRACE
    long long tr_vm = 0, tr_bar = 0, tr_iss = 0, tr_cmp = 0, tr_prev = 0;
    const long long tr_begin = LT_CLK3();
    const long long tr_rt0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    const bool YW = PP && wm == 1;                        // a "Y" wave of the ping-pong (wave-uniform: scalar branches around the barriers)
    auto step = [&](int ks, auto rc) {
        constexpr int R = decltype(rc)::value;           // ks % NBS: which B register set this step multiplies with
#ifdef LT_TRACE
        const long long tr0 = LT_CLK3();
        if (ks > 0) tr_cmp += tr0 - tr_prev;
#endif
        // needed now: B(ks) and, older, stage ks.  The queue, oldest first: prologue [B(0 .. BPF-1) | stages 0 .. P-1], then per step j
        // [B(j+BPF): SN loads | pieces of stage j+NST-1: dps or none].  Everything younger than the youngest needed load may stay in flight.
        const int lp = ABL_A ? 0 : dps, lb = ABL_B ? 0 : SN;
        auto pieces_of = [&](int j) { return j + AHEAD < nk ? lp : 0; };
        int after = 0;
        if (ks >= BPF) {                                   // youngest needed: B(ks), requested at the top of step ks-BPF
            for (int j = ks - BPF; j < ks; ++j) after += pieces_of(j);
            after += (BPF - 1) * lb;
        } else {                                           // first steps: B(ks) leads the prologue; youngest needed: prologue stage ks
            const int P = nk < AHEAD ? nk : AHEAD;
            after = (P - 1 - ks - (YW ? 1 : 0)) * dps;     // a Y wave vouches for stage ks+1 here (its barrier ks+1 comes mid-step)
            if (after < 0) after = 0;
            for (int j = 0; j < ks; ++j) after += lb + pieces_of(j);
        }
        wait_vmcnt6(after);                                // (> 12: waits for everything; only possible in the first steps)
#ifdef LT_TRACE
        const long long tr1 = LT_CLK3();
        tr_vm += tr1 - tr0;
#endif
        // X (and every wave without ping-pong): stage ks landed for every wave (each waited for its own pieces above; a wave's wait
        // at the top of step j covers its pieces up to stage j + AHEAD - BPF, so Y's wait at the top of ITS step ks-1 covered stage ks)
        if (!YW) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef LT_TRACE
        const long long tr2 = LT_CLK3();
        tr_bar += tr2 - tr1;
#endif
#pragma unroll
        for (int j = 0; j < SN; ++j) frag_ready(fb[R][j]);
        if (!ABL_B) {   // B(ks+BPF): the last steps re-read the last fragments (a valid address; never used)
            const T* wn1 = wfrag + (size_t)(ks + BPF < nk ? ks + BPF : nk - 1) * wstep;
            static_for<0, SN>([&](auto jc) { gload16<decltype(jc)::value * 1024>(fb[(R + BPF) % NBS][decltype(jc)::value], wn1, wlane); });
        }
        const bool more = !ABL_A && ks + AHEAD < nk;
        if (more) stage_prep(ks + AHEAD);
#ifdef LT_TRACE
        const long long tr3 = LT_CLK3();                  // B loads + tap bookkeeping
        tr_iss += tr3 - tr2;
        tr_prev = tr3;
#endif
        const unsigned abase = aoff + rbuf;
        V16 fa[RA];
        auto issue = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            lds_read16<K * 16 * ROWB>(fa[K % RA], abase);
        };
        static_for<0, SM>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int prev_target = u == 0 ? 0 : (u + LOOK < SM ? u + LOOK : SM);
            constexpr int target = u + 1 + LOOK < SM ? u + 1 + LOOK : SM;
            static_for<prev_target, target>([&](auto kc) { issue(kc); });
            lgkm_wait<target - u - 1>();
            frag_ready(fa[u % RA]);
#pragma unroll
            for (int j = 0; j < SN; ++j) LT3_MMA(acc[u][j], fa[u % RA], fb[R][j]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u % 2 == 0 && u / 2 < A_IT) {   // one DMA piece of stage ks+AHEAD behind every second fragment
                if (more) stage_piece(ks + AHEAD, wbuf, std::integral_constant<int, u / 2>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (PP && u == YBAR) {              // Y's barrier ks+1 (X takes it at the top of its step ks+1); none in the last step
                if (YW && ks + 1 < nk) asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        rbuf = rbuf + STAGE == REGION ? 0 : rbuf + STAGE;
        wbuf = wbuf + STAGE == REGION ? 0 : wbuf + STAGE;
    };
    if (YW) {   // Y's barrier 0 (X: top of step 0): vouch for the own pieces of stage 0 first, like X does at the top of its step 0
        wait_vmcnt6(((nk < AHEAD ? nk : AHEAD) - 1) * dps);
        asm volatile("s_barrier" ::: "memory");
    }
    if constexpr (NBS == 2) {
        for (int ks = 0; ks < nk; ks += 2) {             // nk is even (k_pad % 64 == 0)
            step(ks, std::integral_constant<int, 0>{});
            step(ks + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (int ks = 0; ks < nk; ks += 6) {             // the set index is ks % 3: six steps per trip, nk even
            step(ks, std::integral_constant<int, 0>{});
            step(ks + 1, std::integral_constant<int, 1>{});
            if (ks + 2 < nk) {
                step(ks + 2, std::integral_constant<int, 2>{});
                step(ks + 3, std::integral_constant<int, 0>{});
            }
            if (ks + 4 < nk) {
                step(ks + 4, std::integral_constant<int, 1>{});
                step(ks + 5, std::integral_constant<int, 2>{});
            }
        }
    }
#ifdef LT_TRACE
    {   // same record as conv_igemm3_kernel: total, vmcnt wait, barrier, issue (B loads), fragment reads + MFMAs + DMA issue, steps, 100 MHz ticks
        const long long tr_end = LT_CLK3();
        tr_cmp += tr_end - tr_prev;
        const long long tr_rt1 = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 1024 && lane == 0) {
            long long* o = g_trace3 + (blockIdx.x >> 3) * 8;
            o[0] = tr_end - tr_begin; o[1] = tr_vm; o[2] = tr_bar; o[3] = tr_iss; o[4] = tr_cmp; o[5] = nk; o[6] = tr_rt1 - tr_rt0; o[7] = blockIdx.x;
        }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring becomes the epilogue staging area
#ifdef LT_ABL_NO_EPI
    if (a.M >= 0) return;
#endif
    // ---- epilogue: three passes of 48 rows through this wave's private fp32 LDS tile -> 16-byte vectors (as conv_igemm5) ----
    constexpr int LPR = WN / 8, RPP = 64 / LPR, ITP = EP_ROWS / RPP;
    const int colv = n0 + wn * WN + (lane % LPR) * 8;
    auto out_off = [&](int p, int k) -> long long {
        const int m = m0 + wm * WM + p * EP_ROWS + k * RPP + lane / LPR;
        if (m >= a.M || colv >= a.Cout) return -1;
        long long pix = m;
        if (!PW) {
            int n, od, oh, ow;
            decode_row(a, m, n, od, oh, ow);
            pix = ((long long)(n * a.OD + od * a.osd + ph.ood) * a.OH + oh * a.osh + ph.ooh) * a.OW + ow * a.osw + ph.oow;
        }
        return pix * a.ldc + colv;
    };
    const EpiFloors fl = epi_floors(a.flags);
    const bool has_res = a.res != nullptr;
    float* ep = (float*)(smem + wave * EP_WAVE);
    float bi[SN], sc[SN], sf[SN];
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int colj = n0 + wn * WN + j * MF + r15;    // < cout_pad: the constant arrays are padded
        bi[j] = a.bias ? a.bias[colj] : 0.f;
        sc[j] = a.scale ? a.scale[colj] : 1.f;
        sf[j] = a.shift ? a.shift[colj] : 0.f;
    }
    static_assert(ITP == E_ITP && EP_ROWS == 48, "residual prefetch geometry");
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        long long off[ITP];
        uint4 rv[ITP];
#pragma unroll
        for (int k = 0; k < ITP; ++k) {                  // this pass's residual vectors first: independent round trips
            off[k] = out_off(p, k);
            if constexpr (RES_EARLY) rv[k] = rv_next[k];
            else rv[k] = (has_res && off[k] >= 0) ? *(const uint4*)((const T*)a.res + off[k]) : make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);
        }
        if constexpr (RES_EARLY) {
            if (p == 0) {          // the compiler cannot see the asm vmcnt(0) behind the K loop: let it place its wait for the early loads HERE, not
#pragma unroll                     // behind the next pass's requests (where it would wait for those too)
                for (int k = 0; k < ITP; ++k) asm volatile("" ::"v"(rv[k].x));
            }
            if (p + 1 < NPASS) res_load(p + 1, rv_next);          // the next pass's round trip runs under this pass's LDS staging and stores
        }
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
#pragma unroll
            for (int j = 0; j < SN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ep[(ii * MF + (lane >> 4) * 4 + e) * EP_LD + j * MF + r15] = (acc[3 * p + ii][j][e] + bi[j]) * sc[j] + sf[j];
#pragma unroll
        for (int k = 0; k < ITP; ++k) {
            if (off[k] < 0) continue;
            const float* src = ep + (k * RPP + lane / LPR) * EP_LD + (lane % LPR) * 8;
            const float4 q0 = *(const float4*)src, q1 = *(const float4*)(src + 4);
            const float vv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const unsigned ru[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
            unsigned ou[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                ou[e] = pack_bf16x2(epi_apply(vv[2 * e], fl, __uint_as_float(ru[e] << 16)), epi_apply(vv[2 * e + 1], fl, __uint_as_float(ru[e] & 0xffff0000u)));
            *(uint4*)((T*)a.y + off[k]) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
        }
    }
}

// lt_conv_fwd packing [cout_pad][k_pad] -> MFMA B-fragment order [k_pad / 32][cout_pad / 16][64 lanes][8]: lane l of a fragment
// holds column 16 tile + (l & 15), K elements 32 step + 8 (l >> 4) .. + 7
__global__ void conv_pack_b_kernel(const bf16_t* __restrict__ w, int cout_pad, int k_pad, bf16_t* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (step * cout_pad / 16 + tile) * 64 + lane
    const long long total = (long long)(k_pad / 32) * (cout_pad / 16) * 64;
    if (g >= total) return;
    const int l = (int)(g & 63);
    const long long ft = g >> 6;
    const int tile = (int)(ft % (cout_pad / 16)), stepk = (int)(ft / (cout_pad / 16));
    *(uint4*)(out + g * 8) = *(const uint4*)(w + (size_t)(16 * tile + (l & 15)) * k_pad + 32 * stepk + 8 * (l >> 4));
}

template <int MODE, int NWM>
int launch6(ConvArgs a, int cout_pad, int max_taps, hipStream_t s) {
    a.tiles_n = cout_pad / 256;
    const long long nblk = cdiv(a.M, 144 * NWM) * a.tiles_n;
    LT_REQUIRE(nblk < (1ll << 31), LT_ERR_INVALID, "lt_conv_fwd: grid too large");
    const size_t lds = LT6_NST * (size_t)(144 * NWM) * 64 + (size_t)max_taps * sizeof(int4);
    LT_REQUIRE(lds <= 160 * 1024, LT_ERR_UNSUPPORTED, "lt_conv_fwd: 288x256 tile needs %zu B of LDS", lds);
    auto kern = conv_igemm6_kernel<MODE, NWM>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256 * NWM), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(v6)");
    return LT_OK;
}

template <int MODE>
int launch5(ConvArgs a, int cout_pad, int max_taps, hipStream_t s) {
    a.tiles_n = cout_pad / 256;
    const long long nblk = cdiv(a.M, BM3) * a.tiles_n;
    LT_REQUIRE(nblk < (1ll << 31), LT_ERR_INVALID, "lt_conv_fwd: grid too large");
    const size_t lds = 4 * (size_t)(BM3 + 256) * 64 + (size_t)max_taps * sizeof(int4);
    LT_REQUIRE(lds <= 160 * 1024, LT_ERR_UNSUPPORTED, "lt_conv_fwd: 288x256 tile needs %zu B of LDS", lds);
    auto kern = conv_igemm5_kernel<MODE>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(v5)");
    return LT_OK;
}

template <int BN, int MODE, int NWM>
int launch3(ConvArgs a, int cout_pad, int max_taps, hipStream_t s) {
    a.tiles_n = cout_pad / BN;
    const long long nblk = cdiv(a.M, BM3) * a.tiles_n;
    LT_REQUIRE(nblk < (1ll << 31), LT_ERR_INVALID, "lt_conv_fwd: grid too large");
    const size_t lds = 3 * (size_t)(BM3 + BN) * ROW_BYTES + (size_t)max_taps * sizeof(int4);
    LT_REQUIRE(lds <= 160 * 1024, LT_ERR_UNSUPPORTED, "lt_conv_fwd: 288-row tile needs %zu B of LDS", lds);
    auto kern = conv_igemm3_kernel<BN, MODE, NWM>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256 * NWM), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(v3)");
    return LT_OK;
}

}  // namespace

namespace lt {

// 1 = launched, 0 = not applicable (caller falls back to conv_igemm2), < 0 = error
int conv3_try(int dtype, const ConvArgs& a, int cout_pad, int nphase, int max_taps, bool forced, hipStream_t s) {
    if (dtype != LT_BF16) return 0;
    if (nphase > 1) {
        // a stride-2 transposed convolution = one convolution per output parity over the same iteration space: the phases differ
        // only in taps, weights and output offset, so the first one decides for all (the 4x4 deconvolutions of the backbone head)
        for (int p = 0; p < nphase; ++p) {
            ConvArgs b = a;
            b.phase[0] = a.phase[p];
            const int rc = conv3_try(dtype, b, cout_pad, 1, max_taps, forced, s);
            if (rc < 0) return rc;
            if (rc == 0) {
                if (p == 0) return 0;
                set_error("lt_conv_fwd: phase %d of a transposed convolution is not supported by the kernel that took phase 0", p);
                return LT_ERR_UNSUPPORTED;
            }
        }
        return 1;
    }
    if (a.flags & (LT_EPI_STORE_F32 | LT_EPI_SIGMOID)) return 0;
    if ((a.Cout % 8) || (a.ldc % 8) || (a.k_pad % 64)) return 0;
    if ((a.Cin * 2) % ROW_BYTES || (a.Cin & (a.Cin - 1))) return 0;   // a 128-byte K step must lie inside one tap; Cin = 2^k
    if (max_taps > 64) return 0;                         // tap table beside the 156 KB ring
    // BN = 64 is only reachable when forced (tests): its 96x16 wave tiles re-read A four times from LDS and measured
    // 1.5x slower than the 128x64 tile of conv_igemm2 on the 64-channel level
    const int BN = cout_pad % 128 == 0 ? 128 : ((cout_pad == 64 && forced) ? 64 : 0);
    if (!BN) return 0;
    // 288 x 256 tile (v5): Cout a multiple of 256 and enough 288-row tiles to give every CU one (LT_CONV_V5=0/1 forces it off/on)
    {
        const char* v5e = getenv("LT_CONV_V5");
        const long long tiles_m5 = cdiv(a.M, BM3), nblk5 = tiles_m5 * (cout_pad / 256);
        const bool fits5 = cout_pad % 256 == 0 && a.k_pad % 32 == 0 && max_taps <= 64;
        const PhaseArg& q0 = a.phase[0];
        const bool pw5 = q0.ntaps == 1 && a.sd == 1 && a.sh == 1 && a.sw == 1 && a.pd == 0 && a.ph == 0 && a.pw == 0 && a.osd == 1 &&
                         a.osh == 1 && a.osw == 1 && q0.ood == 0 && q0.ooh == 0 && q0.oow == 0 && a.OD == a.Do && a.OH == a.Ho &&
                         a.OW == a.Wo && a.D == a.Do && a.H == a.Ho && a.W == a.Wo && a.k_pad == a.Cin;
        const char* no6 = getenv("LT_CONV_NO_V6");   // A/B, read per call
        // small batches (the reference trains at 5 samples = 20 images): the short-K expand layers still fill the chip with the 144-row variant of
        // conv_igemm6 (two workgroups per CU) when the 288-row count says no -- 256 -> 1024 at 20 images: 320 tiles of 144 x 256 instead of
        // 1440 L2-stream-bound 128 x 64 tiles of the generic kernel (LT_CONV_NO_SMALL144=1: off)
        const bool small144 = pw5 && q0.wfrag && !no6 && a.k_pad % 64 == 0 && a.k_pad <= 256 && a.M % 144 == 0 && (a.M / 144) * (cout_pad / 256) >= 200 &&
                              getenv("LT_CONV_NO_SMALL144") == nullptr;
        const bool want5 = v5e ? v5e[0] == '1' : ((nblk5 >= 200 && tiles_m5 * BM3 - a.M <= a.M / 16 && a.k_pad >= 64) || small144);
        if (fits5 && want5) {
            if (q0.wfrag32 && a.k_pad % 64 == 0) {       // weights packed for the 32x32x16 MFMA (plan built with LT_CONV_V7=1): conv_igemm7
                const int rc7 = conv7_try(a, cout_pad, max_taps, pw5, s);
                if (rc7 != 0) return rc7;
            }
            if (q0.wfrag && !no6 && a.k_pad % 64 == 0) {  // weights also available in fragment order: B operand from registers
                // short-K pointwise layers (the 1x1 expands): 144-row tiles, two workgroups per CU (LT_CONV_V6_BM144=0/1 forces it off/on)
                const char* e144 = getenv("LT_CONV_V6_BM144");
                const bool bm144 = e144 ? e144[0] == '1' : (pw5 && a.k_pad <= 256 && a.M % 144 == 0);
                int rc6;
                if (bm144) rc6 = pw5 ? launch6<1, 1>(a, cout_pad, max_taps, s) : launch6<2, 1>(a, cout_pad, max_taps, s);
                else rc6 = pw5 ? launch6<1, 2>(a, cout_pad, max_taps, s) : launch6<2, 2>(a, cout_pad, max_taps, s);
                return rc6 == LT_OK ? 1 : rc6;
            }
            const int rc5 = pw5 ? launch5<1>(a, cout_pad, max_taps, s) : launch5<2>(a, cout_pad, max_taps, s);
            return rc5 == LT_OK ? 1 : rc5;
        }
    }
    const PhaseArg& p0 = a.phase[0];
    const bool pw = p0.ntaps == 1 && a.sd == 1 && a.sh == 1 && a.sw == 1 && a.pd == 0 && a.ph == 0 && a.pw == 0 && a.osd == 1 &&
                    a.osh == 1 && a.osw == 1 && p0.ood == 0 && p0.ooh == 0 && p0.oow == 0 && a.OD == a.Do && a.OH == a.Ho && a.OW == a.Wo &&
                    a.D == a.Do && a.H == a.Ho && a.W == a.Wo && a.k_pad == a.Cin;
    if (!forced) {
        // one workgroup per CU: worth it from about one round of the chip (below that the 64x64 / 128x64 tiles spread better),
        // and only when the 288-row tiles leave little of the last one empty
        const long long tiles_m = cdiv(a.M, BM3), nblk = tiles_m * (cout_pad / BN);
        if (nblk < 200) return 0;
        if (tiles_m * BM3 - a.M > a.M / 16) return 0;
        // short-K layers (the 1x1 expand convs) are bound by their epilogue traffic: many small tiles overlap one workgroup's
        // stores with the next one's loads better than one big tile per CU (measured: 128x64 beats this kernel below K = 512)
        if (a.k_pad < 512) return 0;
    }
    static const bool w8 = getenv("LT_CONV_V3_W8") != nullptr;   // A/B: eight waves (2 per SIMD) instead of twelve
    // (a role-specialised variant -- four compute waves + four loader waves, conv_igemm4 -- was 10-15 % faster per layer on dense data
    // and a wash inside the forward: removed in round 2)
    int rc;
    if (w8) {
        if (BN == 128) rc = pw ? launch3<128, 1, 2>(a, cout_pad, max_taps, s) : launch3<128, 2, 2>(a, cout_pad, max_taps, s);
        else rc = pw ? launch3<64, 1, 2>(a, cout_pad, max_taps, s) : launch3<64, 2, 2>(a, cout_pad, max_taps, s);
    } else {
        if (BN == 128) rc = pw ? launch3<128, 1, 3>(a, cout_pad, max_taps, s) : launch3<128, 2, 3>(a, cout_pad, max_taps, s);
        else rc = pw ? launch3<64, 1, 3>(a, cout_pad, max_taps, s) : launch3<64, 2, 3>(a, cout_pad, max_taps, s);
    }
    return rc == LT_OK ? 1 : rc;
}

}  // namespace lt

extern "C" int lt_conv_pack_weights(const void* weight, int32_t cout_pad, int32_t k_pad, void* packed, void* stream) {
    LT_REQUIRE(weight && packed, LT_ERR_INVALID, "lt_conv_pack_weights: null argument");
    LT_REQUIRE(cout_pad >= 16 && cout_pad % 16 == 0 && k_pad >= 32 && k_pad % 32 == 0, LT_ERR_INVALID,
               "lt_conv_pack_weights: cout_pad %d / k_pad %d (multiples of 16 / 32; bf16 weights [cout_pad][k_pad])", cout_pad, k_pad);
    const long long total = (long long)(k_pad / 32) * (cout_pad / 16) * 64;
    hipLaunchKernelGGL(conv_pack_b_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)weight,
                       cout_pad, k_pad, (bf16_t*)packed);
    LT_CHECK_LAUNCH("lt_conv_pack_weights");
    return LT_OK;
}

#ifdef LT_TRACE
extern "C" int lt_trace_read3(long long* dst, int n, int clear) {
    if (n > 8 * 1024) n = 8 * 1024;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace3), (size_t)n * sizeof(long long)) != hipSuccess) return -2;
    if (clear) {
        static long long zeros[8 * 1024];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace3), zeros, sizeof(zeros)) != hipSuccess) return -3;
    }
    return n;
}
#endif
