// Implicit-GEMM convolution, second generation (the default path of lt_conv_fwd).
//
// Same GEMM view, tile shapes, swizzled 128-byte LDS rows and MFMA fragment scheme as conv_igemm.hip;
// what changes is how bytes move:
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write pass.  The DMA
//     writes wave-linearly (M0 base + lane*16), so the XOR swizzle is applied to the SOURCE: lane L of a
//     wave-instruction owns physical slot L&7 of row L>>3 and fetches the logical 16-byte K vector
//     (L&7) ^ ((row>>1)&7) of that row.  Out-of-image taps (zero padding) fetch from a 16-byte zero page.
//     The DMA is issued from inline asm: hipcc drains vmcnt(0) before the next ds_read after a builtin
//     LDS-DMA (measured: the whole pipeline serialised), so the waits are counted by hand.
//   * NST-stage ring (2 or 3): per K step  { s_waitcnt vmcnt(in-flight stages) ; s_barrier ; issue stage
//     ks+NST-1 ; MFMAs on stage ks }  -- one barrier per step, NST-1 tiles of DMA in flight under the MFMAs.
//   * epilogue: (acc + bias)*scale + shift on the accumulator (fp64 in fp32 mode), then accumulators ->
//     per-wave LDS sub-tile -> every lane owns 16 contiguous output bytes of one pixel: 16-byte residual
//     loads, 16-byte stores (a 64-channel row segment = one contiguous 128/256-byte run).
//   * fp32 mode flushes the MFMA accumulators into fp64 registers every 2 K steps (64 products), which
//     removes the long fp32 accumulation chain from the parity path (K up to 10976 in the 7^3 layer).
//   * PW = pointwise fast path (1x1, stride 1, dense output): rows are contiguous, no tap/bounds logic.
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

// fp32 (parity) kernels: the MFMA accumulators are added into fp64 registers every LT_ACC64_MASK + 1 K steps of 32 products
#ifndef LT_ACC64_MASK
#define LT_ACC64_MASK 3
#endif

using namespace lt;

namespace {

__device__ uint4 g_zero_page[2];  // source of out-of-image taps

#ifdef LT_TRACE
// -DLT_TRACE (profiling build, lt_build.build_variant): shader-clock accounting of the K loop phases, summed per wave and
// written by wave 0 of every 32nd workgroup: [total, wait_vmcnt, barrier, dma_issue, compute, nk, realtime_100MHz, blockIdx]
__device__ long long g_trace[8 * 1024];
#define LT_CLK() ((long long)__builtin_amdgcn_s_memtime())
#endif

typedef __attribute__((address_space(3))) void* lptr_t;

// one LDS-DMA wave-instruction: 64 lanes x 16 B -> lds_base .. lds_base + 1 KiB (wave-uniform base)
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

// s_waitcnt vmcnt(n) for a wave-uniform n (the immediate must be a literal)
__device__ __forceinline__ void wait_vmcnt(int n) {
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
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // conservative
    }
}

__device__ __forceinline__ void block_barrier() {
    // LDS reads of this wave are consumed (lgkmcnt(0)) before it releases the buffer to the next DMA
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int VECO> struct OutVec;  // VECO output elements <-> 16 bytes
template <> struct OutVec<4> {      // fp32 out
    static __device__ __forceinline__ void ld_res(const float* p, float (&f)[4]) {
        const float4 v = *(const float4*)p;
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&f)[4]) { *(float4*)p = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct OutVec<8> {      // bf16 out
    static __device__ __forceinline__ void ld_res(const bf16_t* p, float (&f)[8]) {
        const uint4 v = *(const uint4*)p;
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(u[i] << 16); f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&f)[8]) {
        *(uint4*)p = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
};

__device__ __forceinline__ float epi_act(float v, const EpiFloors& fl, float r, bool sigm) {
    v = epi_apply(v, fl, r);
    if (sigm) v = 1.f / (1.f + expf(-v));   // wave-uniform, only the confidence heads
    return v;
}

// one exact-fp32 MFMA on one operand value per lane (32x32x2 / 16x16x4); only instantiated for T = float
template <int MF, typename ACC>
__device__ __forceinline__ ACC mfma_f32_k2(float a, float b, ACC c) {
    if constexpr (MF == 32) return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <typename T, int BM, int BN, int WM, int WN, int MF, int NST, int MODE>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(const ConvArgs a) {
    constexpr bool PW = MODE == 1;   // pointwise: rows contiguous, no taps
    constexpr bool UT = MODE == 2;   // uniform tap: a 128-byte K step never straddles two taps (Cin*sizeof(T) % 128 == 0)
    constexpr bool ACC64 = sizeof(T) == 4;  // fp32 = parity mode
    constexpr int VEC = elt<T>::vec;
    constexpr int BK = 8 * VEC;
    constexpr int A_IT = BM / 32;                  // DMA instructions per wave for the A tile
    constexpr int B_VECS = BN * 8;
    constexpr int B_IT = (B_VECS + 255) / 256;
    constexpr int SM = WM / MF, SN = WN / MF;
    constexpr int WAVES_N = BN / WN;
    constexpr int G = (MF == 32) ? 4 : 2;
    constexpr int NACC = (MF == 32) ? 16 : 4;
    constexpr int STAGE = (BM + BN) * ROW_BYTES;   // bytes per pipeline stage
    constexpr int EP_LD = WN + 4;                  // padded fp32 row of the per-wave epilogue tile
    constexpr int EP_WAVE = WM * EP_LD * 4;        // bytes
    constexpr int REGION = (NST * STAGE > 4 * EP_WAVE) ? NST * STAGE : 4 * EP_WAVE;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves");
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_rowpix = (int*)(smem + REGION);         // [BM]   (unused when PW)
    int4* s_taps = (int4*)(s_rowpix + BM);         // [ntaps]
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;   // LDS byte address of the dynamic region

    // Address of the zero page, made opaque once: otherwise every use re-loads it from the GOT (s_load + s_waitcnt lgkmcnt(0)),
    // which inside the K loop also drains the fragment ds_reads in flight
    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs (each with its own L2),
    // so workgroup b runs on XCD b%8. Give every XCD one contiguous run of the (tile_m, tile_n) raster instead of every
    // 8th tile: the tiles_n workgroups that re-read one A row panel, and the neighbouring row panels whose 3x3 taps
    // overlap, then meet in the same L2 rather than each pulling the panel from HBM / Infinity Cache.
    int lin = blockIdx.x;
    if (!(a.flags & LT_EPI_NO_XCD_REMAP)) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = lin & 7, j = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_n = lin % a.tiles_n;
    const int tile_m = lin / a.tiles_n;
    const PhaseArg ph = a.phase[blockIdx.y];
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ x = (const T*)a.x;
    const T* __restrict__ w = (const T*)ph.w;

    // DMA ownership: thread t fills physical slot t&7 of rows (t>>3) + 32*i; the logical K vector it must fetch is
    // v = (t&7) ^ ((row>>1)&7), the same for all its rows (32*i does not change (row>>1)&7)
    const int v = (t & 7) ^ ((t >> 4) & 7);
    int id0[PW ? 1 : A_IT], ih0[PW ? 1 : A_IT], iw0[PW ? 1 : A_IT], baseC[A_IT];
    if (PW) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + (t >> 3) + 32 * i;
            baseC[i] = (m < a.M) ? m * a.Cin + v * VEC : -1;
        }
    } else {
        for (int i = t; i < ph.ntaps; i += 256) s_taps[i] = ph.taps[i];
        for (int r = t; r < BM; r += 256) {
            int m = m0 + r, pix = -1;
            if (m < a.M) {
                int n, od, oh, ow;
                decode_row(a, m, n, od, oh, ow);
                pix = ((n * a.OD + od * a.osd + ph.ood) * a.OH + oh * a.osh + ph.ooh) * a.OW + ow * a.osw + ph.oow;
            }
            s_rowpix[r] = pix;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int m = m0 + (t >> 3) + 32 * i;
            if (m < a.M) {
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
    int cur[UT ? A_IT : 1];   // UT: element offset of (row, current tap, this lane's vector) or -1 when the tap is out of the image
    const T* wrow[B_IT];
    int nb_wave = 0;  // B-tile DMA instructions this wave issues per stage
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        wrow[j] = w + (size_t)(n0 + ((t + 256 * j) >> 3)) * a.k_pad + v * VEC;
        if (B_VECS >= 256 * (j + 1) || 8 * wave + 32 * j < BN) ++nb_wave;
    }
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
    const int dps = (ABL_A ? 0 : A_IT) + (ABL_B ? 0 : nb_wave);  // DMA instructions per stage per wave (wave-uniform)
    __syncthreads();

    const int nk = a.k_pad / BK;

    // ---- residual prefetch (bf16 vector epilogue only): the lane's 16-byte residual vectors are requested BEFORE the K loop.
    // The whole forward is HBM-traffic-bound in bf16 (35.6 GB of activation traffic per B=16 step vs 9.6 TFLOP); this adds
    // WM*WN*2 bytes per wave of loads in flight under the main loop and takes the residual round trip out of the epilogue.
    constexpr int PF_LPR = WN / 8, PF_RPP = 64 / PF_LPR, PF_NIT = WM / PF_RPP;
    static_assert(PF_NIT <= 8, "at most 8 residual vectors per lane");
    // eight NAMED registers (an array -- even fully unrolled -- was kept in scratch memory by hipcc across the asm K loop)
    uint4 rp0, rp1, rp2, rp3, rp4, rp5, rp6, rp7;
    rp0 = rp1 = rp2 = rp3 = rp4 = rp5 = rp6 = rp7 = make_uint4(0, 0, 0, 0);
    const bool pre_res = sizeof(T) == 2 && a.res != nullptr && !(a.flags & (LT_EPI_STORE_F32 | LT_EPI_NO_RES_PREFETCH)) &&
                         (a.Cout % 8 == 0) && (a.ldc % 8 == 0);
    if constexpr (sizeof(T) == 2) {
        if (pre_res) {
            const int colp = n0 + wn * WN + (lane % PF_LPR) * 8;
            auto pf = [&](int it) -> uint4 {
                const int r = wm * WM + lane / PF_LPR + it * PF_RPP;
                int pix;
                if (PW) { const int m = m0 + r; pix = m < a.M ? m : -1; }
                else pix = s_rowpix[r];
                // rows beyond M / channels beyond Cout read the zero page; their values are never used
                const void* src = (pix >= 0 && colp < a.Cout) ? (const void*)((const bf16_t*)a.res + (size_t)pix * a.ldc + colp)
                                                             : zero_page;
                return *(const uint4*)src;
            };
            if (PF_NIT > 0) rp0 = pf(0);
            if (PF_NIT > 1) rp1 = pf(1);
            if (PF_NIT > 2) rp2 = pf(2);
            if (PF_NIT > 3) rp3 = pf(3);
            if (PF_NIT > 4) rp4 = pf(4);
            if (PF_NIT > 5) rp5 = pf(5);
            if (PF_NIT > 6) rp6 = pf(6);
            if (PF_NIT > 7) rp7 = pf(7);
        }
    }

    // DMA pieces of one stage: A pieces 0..A_IT-1, then B pieces A_IT..A_IT+B_IT-1; [p0, p1) selects a slice (compile-time
    // after unrolling) so that the K loop can issue them between MFMA groups
    constexpr int NPIECE = A_IT + B_IT;
    auto stage = [&](int ks, int buf, int p0, int p1) {
        const unsigned sA = lds0 + buf * STAGE;
        const unsigned sB = sA + BM * ROW_BYTES;
        if (ABL_A) p0 = p0 < A_IT ? A_IT : p0;
        if (ABL_B) p1 = p1 > A_IT ? A_IT : p1;
        if (PW) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                if (i < p0 || i >= p1) continue;
                const void* src = baseC[i] >= 0 ? (const void*)(x + (baseC[i] + ks * BK)) : zero_page;
                dma16(src, sA + (8 * wave + 32 * i) * ROW_BYTES);
            }
        } else if (UT) {
            // the tap (and with it the bounds test) changes only every Cin/BK steps: PMC on the generic path showed 15-40 VALU
            // instructions of address arithmetic per MFMA on narrow tiles
            const int k0 = ks * BK;                      // wave-uniform
            const int c0 = k0 & (a.Cin - 1);
            if (c0 == 0 && p0 == 0) {
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
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                if (i < p0 || i >= p1) continue;
                const void* src = cur[i] >= 0 ? (const void*)(x + (cur[i] + c0)) : zero_page;
                dma16(src, sA + (8 * wave + 32 * i) * ROW_BYTES);
            }
        } else {
            const int kel = ks * BK + v * VEC;
            const int tap = kel >> a.log2Cin;
            const int c = kel & (a.Cin - 1);
            int4 tp = make_int4(-(1 << 24), 0, 0, 0);
            if (tap < ph.ntaps) tp = s_taps[tap];
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                if (i < p0 || i >= p1) continue;
                const int id = id0[i] + tp.x, ih = ih0[i] + tp.y, iw = iw0[i] + tp.z;
                const bool ok = ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                const void* src = ok ? (const void*)(x + (baseC[i] + tp.w + c)) : zero_page;
                dma16(src, sA + (8 * wave + 32 * i) * ROW_BYTES);
            }
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            if (A_IT + j < p0 || A_IT + j >= p1) continue;
            if (B_VECS >= 256 * (j + 1) || 8 * wave + 32 * j < BN)   // wave-uniform: whole 8-row groups
                dma16(wrow[j] + ks * BK, sB + (8 * wave + 32 * j) * ROW_BYTES);
        }
    };

    const int frow = lane & (MF - 1);
    const int fsw = (frow >> 1) & 7;
    int foff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int vec = (MF == 32) ? (2 * g + (lane >> 5)) : ((lane >> 4) + 4 * g);
        foff[g] = frow * ROW_BYTES + ((vec ^ fsw) << 4);
    }
    const int a_base = (wm * WM) * ROW_BYTES;
    const int b_base = BM * ROW_BYTES + (wn * WN) * ROW_BYTES;

    acc_t acc[SM][SN];
    double dacc[ACC64 ? SM : 1][ACC64 ? SN : 1][ACC64 ? NACC : 1];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                acc[i][j][e] = 0.f;
                if (ACC64) dacc[i][j][e] = 0.0;
            }

    // prologue: NST-1 stages in flight
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s, 0, NPIECE);

    // One K step. MORE_ (compile time): stage ks+NST-1 is still to be requested, right after the barrier, so that it has the
    // whole K step to land.  (-DLT_DMA_SLICED issues the pieces in slices behind the MFMAs of each fragment group instead:
    // measured 5 % SLOWER end to end with the 2-stage ring -- the last slice then has only a quarter of a K step to land.)
    // The main loop and the NST-1 drain steps are separate loops so that the K step has no control flow.
#ifdef LT_DMA_SLICED
    constexpr bool UPFRONT = false;
#else
    constexpr bool UPFRONT = true;
#endif
#ifdef LT_FP32_NO_PIPE
    constexpr bool PIPE32 = false;
#else
    constexpr bool PIPE32 = ACC64 && UPFRONT;
#endif
#ifdef LT_TRACE
#define LT_TR0 const long long tr0 = LT_CLK(); if (ks > 0) tr_cmp += tr0 - tr_prev;
#define LT_TR1 const long long tr1 = LT_CLK(); tr_vm += tr1 - tr0;
#define LT_TR2 const long long tr2 = LT_CLK(); tr_bar += tr2 - tr1;
#define LT_TR3 const long long tr3 = LT_CLK(); tr_iss += tr3 - tr2; tr_prev = tr3;
#else
#define LT_TR0
#define LT_TR1
#define LT_TR2
#define LT_TR3
#endif
#ifdef LT_ABL_NO_MMA
#define LT_MMA_RUN(c_, a_, b_) (void)0
#else
#define LT_MMA_RUN(c_, a_, b_) Mma<T, MF>::run(c_, a_, b_)
#endif
    /* fp32 (round 6): the 128 x 128 tile holds 64 accumulators + 64 fp64 sums per lane = 320 registers, i.e. ONE wave per SIMD, and an exact-fp32 MFMA   \
       occupies the matrix pipe for 64 cycles: everything the wave issues BETWEEN two MFMAs is free, everything it issues before the first or after the    \
       last one of a K step is exposed (shader-clock trace, 256 images, 3x3 256 -> 256: 5490 cycles per K step for 4096 of MFMA -- 620 DMA issue, ~700      \
       around the fragment groups and the fp64 flush, 130 wait + barrier).  PIPE32: the next stage's DMA pieces go out one by one behind the MFMA blocks   \
       of the first half of the step, and a flush step folds block b - 1 into its fp64 sums behind the MFMAs of block b.  -DLT_FP32_NO_PIPE: the old order. */ \
#define LT_KSTEP(MORE_)                                                                                              \
    {                                                                                                                \
        LT_TR0                                                                                                       \
        wait_vmcnt((MORE_ ? NST - 2 : (nk - 1 - ks < NST - 2 ? nk - 1 - ks : NST - 2)) * dps);                       \
        LT_TR1                                                                                                       \
        block_barrier(); /* everybody's stage-ks DMAs landed, everybody is done reading stage ks-1 */                \
        LT_TR2                                                                                                       \
        if (MORE_ && UPFRONT && !PIPE32) stage(ks + NST - 1, (ks + NST - 1) % NST, 0, NPIECE);                       \
        LT_TR3                                                                                                       \
        const int buf = ks % NST;                                                                                    \
        const unsigned char* pa = smem + buf * STAGE + a_base;                                                       \
        const unsigned char* pb = smem + buf * STAGE + b_base;                                                       \
        /* fragment group g+1 is requested before the MFMAs of group g issue (register double buffer, order pinned) */ \
        V16 fa[2][SM], fb[2][SN];                                                                                    \
        auto load_group = [&](int g, int slot) {                                                                     \
            _Pragma("unroll") for (int i = 0; i < SM; ++i) fa[slot][i].u = *(const uint4*)(pa + i * MF * ROW_BYTES + foff[g]); \
            _Pragma("unroll") for (int j = 0; j < SN; ++j) fb[slot][j].u = *(const uint4*)(pb + j * MF * ROW_BYTES + foff[g]); \
        };                                                                                                           \
        load_group(0, 0);                                                                                            \
        if constexpr (PIPE32) {                                                                                      \
            const bool flush_ = (ks & LT_ACC64_MASK) == LT_ACC64_MASK || ks + 1 == nk;                               \
            constexpr int NU_ = G * 4, NUH_ = (NU_ / 2 >= NPIECE) ? NU_ / 2 : NU_;                                    \
            _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                          \
                if (g + 1 < G) load_group(g + 1, (g + 1) & 1);                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
                /* K pair e of the fragment OUTERMOST: consecutive MFMAs write different accumulator blocks (a block is revisited every NB_ MFMAs), so no    \
                   MFMA waits for the one in front of it -- four dependent 32x32x2 MFMAs in a row ran ~10 % below the issue rate (trace: 4740 cycles for     \
                   64 MFMAs of 64) -- and the DMA pieces sit between rounds of independent MFMAs */                                                           \
                _Pragma("unroll") for (int e4 = 0; e4 < 4; ++e4) {                                                   \
                    const int unit = g * 4 + e4;                                                                     \
                    _Pragma("unroll") for (int i = 0; i < SM; ++i)                                                   \
                        _Pragma("unroll") for (int j = 0; j < SN; ++j)                                               \
                            acc[i][j] = mfma_f32_k2<MF>(fa[g & 1][i].f[e4], fb[g & 1][j].f[e4], acc[i][j]);          \
                    if (MORE_ && unit < NUH_) stage(ks + NST - 1, (ks + NST - 1) % NST, unit * NPIECE / NUH_, (unit + 1) * NPIECE / NUH_); \
                    __builtin_amdgcn_sched_barrier(0);                                                               \
                }                                                                                                    \
            }                                                                                                        \
            if (flush_) {                                                                                            \
                _Pragma("unroll") for (int i = 0; i < SM; ++i)                                                       \
                    _Pragma("unroll") for (int j = 0; j < SN; ++j)                                                   \
                        _Pragma("unroll") for (int e = 0; e < NACC; ++e) { dacc[i][j][e] += (double)acc[i][j][e]; acc[i][j][e] = 0.f; } \
            }                                                                                                        \
        } else {                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                              \
            if (g + 1 < G) load_group(g + 1, (g + 1) & 1);                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            _Pragma("unroll") for (int i = 0; i < SM; ++i)                                                           \
                _Pragma("unroll") for (int j = 0; j < SN; ++j) LT_MMA_RUN(acc[i][j], fa[g & 1][i], fb[g & 1][j]);      \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            if (MORE_ && !UPFRONT) {                                                                                 \
                stage(ks + NST - 1, (ks + NST - 1) % NST, g * NPIECE / G, (g + 1) * NPIECE / G);                     \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
            }                                                                                                        \
        }                                                                                                            \
        if (ACC64 && ((ks & LT_ACC64_MASK) == LT_ACC64_MASK || ks + 1 == nk)) {                                                              \
            _Pragma("unroll") for (int i = 0; i < SM; ++i)                                                           \
                _Pragma("unroll") for (int j = 0; j < SN; ++j)                                                       \
                    _Pragma("unroll") for (int e = 0; e < NACC; ++e) {                                               \
                        dacc[i][j][e] += (double)acc[i][j][e];                                                       \
                        acc[i][j][e] = 0.f;                                                                          \
                    }                                                                                                \
        }                                                                                                            \
        }                                                                                                            \
    }
#ifdef LT_TRACE
    long long tr_vm = 0, tr_bar = 0, tr_iss = 0, tr_cmp = 0, tr_prev = 0;
    const long long tr_begin = LT_CLK();
    const long long tr_rt0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    int ks = 0;
    for (; ks + NST - 1 < nk; ++ks) LT_KSTEP(true)
    for (; ks < nk; ++ks) LT_KSTEP(false)
#undef LT_KSTEP
#undef LT_MMA_RUN
#ifdef LT_TRACE
    {
        const long long tr_end = LT_CLK();
        tr_cmp += tr_end - tr_prev;
        const long long tr_rt1 = (long long)__builtin_amdgcn_s_memrealtime();
        if (wave == 0 && (blockIdx.x & 31) == 0 && blockIdx.y == 0 && (blockIdx.x >> 5) < 1024 && lane == 0) {
            long long* o = g_trace + (blockIdx.x >> 5) * 8;
            o[0] = tr_end - tr_begin; o[1] = tr_vm; o[2] = tr_bar; o[3] = tr_iss; o[4] = tr_cmp; o[5] = nk; o[6] = tr_rt1 - tr_rt0; o[7] = blockIdx.x;
        }
    }
#endif
#ifdef LT_ABL_NO_EPI
    if (a.M >= 0) return;
#endif
    block_barrier();   // all waves done with the last stage: the region is reused by the epilogue tiles

    // ---- epilogue: (acc + bias)*scale + shift, then this wave's LDS sub-tile (fp32, padded rows) -> 16-byte vectors ---
    float* ep = (float*)(smem + wave * EP_WAVE);
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int colj = n0 + wn * WN + j * MF + (lane & (MF - 1));   // < cout_pad: the constant arrays are padded
        const float bi = a.bias ? a.bias[colj] : 0.f;
        const float sc = a.scale ? a.scale[colj] : 1.f;
        const float sf = a.shift ? a.shift[colj] : 0.f;
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                const int r = i * MF + ((MF == 32) ? ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) : ((lane >> 4) * 4 + e));
                const int cc = j * MF + (lane & (MF - 1));
                float val;
                if (ACC64) val = (float)((dacc[i][j][e] + (double)bi) * (double)sc + (double)sf);
                else val = (acc[i][j][e] + bi) * sc + sf;
                ep[r * EP_LD + cc] = val;
            }
    }
    __syncthreads();

    const EpiFloors fl = epi_floors(a.flags);
    const bool sigm = a.flags & LT_EPI_SIGMOID;
    const bool store_f32 = (a.flags & LT_EPI_STORE_F32) != 0 || sizeof(T) == 4;
    const bool has_res = a.res != nullptr;
    const bool res_f32 = (a.flags & LT_EPI_RES_F32) != 0 && sizeof(T) <= 2;      // fp32 residual of a bf16 / fp8 convolution that stores fp32
    const int col0 = n0 + wn * WN;                 // first output channel of this wave's sub-tile
    const int veco = store_f32 ? 4 : 8;
    const bool vec_ok = (a.Cout % veco == 0) && (a.ldc % veco == 0);
    auto row_pix = [&](int r) -> int {             // output pixel index of tile row r (or -1)
        if (PW) { const int m = m0 + r; return m < a.M ? m : -1; }
        return s_rowpix[r];
    };
    if (vec_ok) {
        // All residual loads of the lane are issued first (they are independent), then consumed: the first version
        // loaded, waited and stored row by row, i.e. WM/rows-per-pass serialized HBM round trips per workgroup.
        // (a macro, not a lambda: capturing the prefetched-residual array by reference kept it in scratch memory)
#define LT_EPILOGUE_ROWS(VECO_, OUT_F32_, PRE_)                                                                           \
        {                                                                                                              \
            constexpr int VECO = VECO_;                                                                                \
            constexpr bool OUT_F32 = OUT_F32_;                                                                         \
            constexpr int LPR = WN / VECO, RPP = 64 / LPR, NIT = WM / RPP;                                             \
            const int cq = (lane % LPR) * VECO;                                                                        \
            const int col = col0 + cq;                                                                                 \
            if (col < a.Cout) {                                                                                        \
                int pix[NIT];                                                                                          \
                float rr[NIT][VECO];                                                                                   \
                _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                   \
                    pix[it] = row_pix(wm * WM + lane / LPR + it * RPP);                                                \
                    _Pragma("unroll") for (int e = 0; e < VECO; ++e) rr[it][e] = -0.0f; /* v + -0.0 == v */              \
                    if (has_res && pix[it] >= 0) {                                                                     \
                        const size_t off = (size_t)pix[it] * a.ldc + col;                                              \
                        if constexpr (sizeof(T) == 4) OutVec<VECO>::ld_res((const float*)a.res + off, rr[it]);         \
                        else if constexpr (!OUT_F32) {                                                                 \
                            uint4 pv;                                                                                  \
                            if (pre_res) pv = PRE_;                                                                    \
                            else pv = *(const uint4*)((const bf16_t*)a.res + off);                                     \
                            const unsigned u[4] = {pv.x, pv.y, pv.z, pv.w};                                            \
                            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
                                rr[it][2 * e] = __uint_as_float(u[e] << 16);                                           \
                                rr[it][2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);                               \
                            }                                                                                          \
                        } else {                                                                                       \
                            if (res_f32) OutVec<VECO>::ld_res((const float*)a.res + off, rr[it]);                    \
                            else { _Pragma("unroll") for (int e = 0; e < VECO; ++e) rr[it][e] = bf16_to_f32(((const bf16_t*)a.res)[off + e]); } \
                        }                                                                                              \
                    }                                                                                                  \
                }                                                                                                      \
                _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                   \
                    if (pix[it] < 0) continue;                                                                         \
                    const int r = lane / LPR + it * RPP;                                                               \
                    const size_t off = (size_t)pix[it] * a.ldc + col;                                                  \
                    const float* src = ep + r * EP_LD + cq;                                                            \
                    float vv[VECO];                                                                                    \
                    _Pragma("unroll") for (int e = 0; e < VECO; e += 4) {                                              \
                        const float4 q = *(const float4*)(src + e);                                                    \
                        vv[e] = q.x; vv[e + 1] = q.y; vv[e + 2] = q.z; vv[e + 3] = q.w;                                \
                    }                                                                                                  \
                    _Pragma("unroll") for (int e = 0; e < VECO; ++e) vv[e] = epi_act(vv[e], fl, rr[it][e], sigm); \
                    typedef typename std::conditional<VECO == 4, float, bf16_t>::type out_t;                          \
                    OutVec<VECO>::st((out_t*)a.y + off, vv);                                                           \
                }                                                                                                      \
            }                                                                                                          \
        }
        if constexpr (sizeof(T) == 4) LT_EPILOGUE_ROWS(4, true, make_uint4(0, 0, 0, 0))
        else {
            if (store_f32) LT_EPILOGUE_ROWS(4, true, make_uint4(0, 0, 0, 0))
            else if (!pre_res) LT_EPILOGUE_ROWS(8, false, make_uint4(0, 0, 0, 0))
            else {
                // prefetched residual: one explicitly numbered row per named register (no indexing by a loop variable)
                constexpr int LPR = WN / 8, RPP = 64 / LPR, NIT = WM / RPP;
                const int cq = (lane % LPR) * 8;
                const int col = col0 + cq;
                auto row = [&](int it, uint4 pv) {
                    const int r = lane / LPR + it * RPP;
                    const int pixv = row_pix(wm * WM + r);
                    if (pixv < 0) return;
                    const size_t off = (size_t)pixv * a.ldc + col;
                    const float* src = ep + r * EP_LD + cq;
                    const float4 q0 = *(const float4*)src, q1 = *(const float4*)(src + 4);
                    float vv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    const unsigned u[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vv[2 * e] = epi_act(vv[2 * e], fl, __uint_as_float(u[e] << 16), sigm);
                        vv[2 * e + 1] = epi_act(vv[2 * e + 1], fl, __uint_as_float(u[e] & 0xffff0000u), sigm);
                    }
                    OutVec<8>::st((bf16_t*)a.y + off, vv);
                };
                if (col < a.Cout) {
                    if (NIT > 0) row(0, rp0);
                    if (NIT > 1) row(1, rp1);
                    if (NIT > 2) row(2, rp2);
                    if (NIT > 3) row(3, rp3);
                    if (NIT > 4) row(4, rp4);
                    if (NIT > 5) row(5, rp5);
                    if (NIT > 6) row(6, rp6);
                    if (NIT > 7) row(7, rp7);
                }
            }
        }
#undef LT_EPILOGUE_ROWS
    } else {
        // ragged channel counts (Cout = 17, ...): one element per lane, lanes along channels
        for (int idx = lane; idx < WM * WN; idx += 64) {
            const int r = idx / WN, cc = idx - r * WN;
            const int col = col0 + cc;
            const int pix = row_pix(wm * WM + r);
            if (pix < 0 || col >= a.Cout) continue;
            const size_t off = (size_t)pix * a.ldc + col;
            const float rr = has_res ? (res_f32 ? ((const float*)a.res)[off] : elt<T>::ld((const T*)a.res + off)) : -0.0f;
            const float val = epi_act(ep[r * EP_LD + cc], fl, rr, sigm);
            if (store_f32) ((float*)a.y)[off] = val;
            else elt<T>::st((T*)a.y + off, val);
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int MF, int NST, int MODE>
int launch2(ConvArgs a, int cout_pad, int nphase, int max_taps, hipStream_t s) {
    a.tiles_n = cout_pad / BN;
    const long long nblk = cdiv(a.M, BM) * a.tiles_n;
    LT_REQUIRE(nblk < (1ll << 31), LT_ERR_INVALID, "lt_conv_fwd: grid too large");
    constexpr int STAGE = (BM + BN) * ROW_BYTES;
    constexpr int EP_WAVE = WM * (WN + 4) * 4;
    constexpr int REGION = (NST * STAGE > 4 * EP_WAVE) ? NST * STAGE : 4 * EP_WAVE;
    const size_t lds = REGION + BM * sizeof(int) + (size_t)max_taps * sizeof(int4);
    LT_REQUIRE(lds <= 160 * 1024, LT_ERR_UNSUPPORTED, "lt_conv_fwd: tile needs %zu B of LDS", lds);
    auto kern = conv_igemm2_kernel<T, BM, BN, WM, WN, MF, NST, MODE>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, nphase), dim3(256), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(v2)");
    return LT_OK;
}

template <typename T, int BM, int BN, int WM, int WN, int MF>
int launch2_pick(const ConvArgs& a, int cout_pad, int nphase, int max_taps, int nst, int mode, hipStream_t s) {
    LT_REQUIRE(cout_pad % BN == 0, LT_ERR_INVALID, "lt_conv_fwd: cout_pad %d not a multiple of tile N %d", cout_pad, BN);
#define LT_PICK(NST_)                                                                                       \
    switch (mode) {                                                                                         \
        case 1: return launch2<T, BM, BN, WM, WN, MF, NST_, 1>(a, cout_pad, nphase, max_taps, s);           \
        case 2: return launch2<T, BM, BN, WM, WN, MF, NST_, 2>(a, cout_pad, nphase, max_taps, s);           \
        default: return launch2<T, BM, BN, WM, WN, MF, NST_, 0>(a, cout_pad, nphase, max_taps, s);          \
    }
    if (nst == 3) { LT_PICK(3) }
    LT_PICK(2)
#undef LT_PICK
}

template <typename T>
int dispatch2(const ConvArgs& a, int cout_pad, int nphase, int max_taps, int tile, hipStream_t s) {
    // workgroups of each candidate tile; 256 CUs want >= ~2 resident workgroups each to hide the DMA latency of the
    // 2-stage ring, so the largest tile that still gives >= 480 workgroups wins (measured: 288 workgroups of 128x128 on a
    // M=18432, N=256 layer ran 1.6x slower than 576 of 128x64)
    auto blocks = [&](int bm, int bn) { return cdiv(a.M, bm) * (long long)(cout_pad / bn) * nphase; };
    // pointwise fast path: one tap at offset 0, unit strides, dense output rows
    const PhaseArg& p0 = a.phase[0];
    const bool pw = nphase == 1 && p0.ntaps == 1 && a.sd == 1 && a.sh == 1 && a.sw == 1 && a.pd == 0 && a.ph == 0 && a.pw == 0 &&
                    a.osd == 1 && a.osh == 1 && a.osw == 1 && p0.ood == 0 && p0.ooh == 0 && p0.oow == 0 && a.OD == a.Do && a.OH == a.Ho &&
                    a.OW == a.Wo && a.D == a.Do && a.H == a.Ho && a.W == a.Wo && a.k_pad == a.Cin;
    if (tile == LT_TILE_AUTO) {
        static const int minblk = getenv("LT_CONV2_MINBLK") ? atoi(getenv("LT_CONV2_MINBLK")) : 480;          // tuning knob
        if (cout_pad <= 16) tile = LT_TILE2_256x16;
        else if (cout_pad <= 32) tile = LT_TILE2_256x32;
        else if (cout_pad <= 64) tile = blocks(128, 64) >= minblk ? LT_TILE2_128x64 : LT_TILE2_64x64;
        else if (pw && a.k_pad <= 256 && blocks(128, 64) >= minblk) tile = LT_TILE2_128x64;   // short-K expand convs: epilogue-bound,
        // more and smaller workgroups overlap stores with loads (measured 171 vs 204 us on 64->256 at 96^2, 111 vs 120 on 128->512)
        // exact-fp32 (round 6): the 128 x 128 tile is ONE wave per SIMD (320 registers), the 128 x 64 tile two to three.  Measured at 256 images
        // (tools/conv_bench.py --dtype fp32): 1x1 1024 -> 256 712 -> 656 us, 1x1 512 -> 128 785 -> 682, 3x3 128 -> 128 1614 -> 1565; 3x3 256 -> 256 stays (1503 vs 1553)
        else if (sizeof(T) == 4 && (pw || cout_pad <= 128) && blocks(128, 64) >= minblk) tile = LT_TILE2_128x64;
        else tile = blocks(128, 128) >= minblk ? LT_TILE2_128x128 : (blocks(128, 64) >= minblk ? LT_TILE2_128x64 : LT_TILE2_64x64);
    }
    // uniform-tap path: every 128-byte K step lies inside one tap
    static const bool no_ut = getenv("LT_CONV_NO_UT") != nullptr;   // A/B switch
    const int mode = pw ? 1 : (((a.Cin * (int)sizeof(T)) % ROW_BYTES == 0 && !no_ut) ? 2 : 0);
    // ring depth: 3 stages cost a resident workgroup per CU on the big tiles, so they only pay when the grid leaves at most
    // one workgroup per CU anyway (tiny layers: pure latency chains)
    int nst = a.stages;
    if (nst != 2 && nst != 3) {
        long long nblk = 0;
        switch (tile) {
            case LT_TILE2_128x128: nblk = blocks(128, 128); break;
            case LT_TILE2_128x64: nblk = blocks(128, 64); break;
            case LT_TILE2_64x64: nblk = blocks(64, 64); break;
            default: nblk = 1 << 20; break;
        }
        nst = nblk <= 256 ? 3 : 2;
    }
    switch (tile) {
        case LT_TILE2_128x128: return launch2_pick<T, 128, 128, 64, 64, 32>(a, cout_pad, nphase, max_taps, nst, mode, s);
        case LT_TILE2_128x64: return launch2_pick<T, 128, 64, 64, 32, 32>(a, cout_pad, nphase, max_taps, nst, mode, s);
        case LT_TILE2_256x32: return launch2_pick<T, 256, 32, 64, 32, 32>(a, cout_pad, nphase, max_taps, nst, mode, s);
        case LT_TILE2_256x16: return launch2_pick<T, 256, 16, 64, 16, 16>(a, cout_pad, nphase, max_taps, nst, mode, s);
        case LT_TILE2_64x64: return launch2_pick<T, 64, 64, 32, 32, 32>(a, cout_pad, nphase, max_taps, nst, mode, s);
        default: break;
    }
    set_error("lt_conv_fwd: unknown tile id %d", tile);
    return LT_ERR_INVALID;
}

}  // namespace

namespace lt {
int conv2_dispatch(int dtype, const ConvArgs& a, int cout_pad, int nphase, int max_taps, int tile, hipStream_t s) {
    if (dtype == LT_F32) return dispatch2<float>(a, cout_pad, nphase, max_taps, tile, s);
    if (dtype == LT_FP8) return dispatch2<fp8_t>(a, cout_pad, nphase, max_taps, tile, s);
    return dispatch2<bf16_t>(a, cout_pad, nphase, max_taps, tile, s);
}
}  // namespace lt

#ifdef LT_TRACE
extern "C" int lt_trace_read(long long* dst, int n, int clear) {
    if (n > 8 * 1024) n = 8 * 1024;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace), (size_t)n * sizeof(long long)) != hipSuccess) return -2;
    if (clear) {
        static long long zeros[8 * 1024];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace), zeros, sizeof(zeros)) != hipSuccess) return -3;
    }
    return n;
}
#endif
