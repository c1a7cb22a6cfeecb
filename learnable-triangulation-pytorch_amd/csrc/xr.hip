// xr_kernel: the 1x1 EXPAND of one ResNet bottleneck block and the 1x1 REDUCE of the NEXT block in one launch (identity blocks of layer3 of the pose
// backbone, reference mvn/models/pose_resnet.py:75-95: ... conv3 -> bn3 -> += x -> relu | next block: conv1 -> bn1 -> relu ...).
//
// Why (VERDICT r4 "next" 2): layer3's 36 identity blocks are a third of the forward, and their two pointwise layers are bound by the 4 P-channel tensor,
// not by the MFMA: the expand reads the residual (4 P channels) and writes its output (4 P), the next block's reduce reads that output again -- 12 P + 2 P
// channel-units per pixel through HBM for 8 P^2 multiply-adds.  Here a workgroup owns 96 pixels, computes the expand's output in chunks of 128 channels,
// writes each chunk to memory ONCE and, while it is in LDS, feeds it to the reduce's accumulators: the second read of the 4 P-channel tensor (a third of the two
// layers' traffic) is gone.  Pointwise on both sides, so a tile is any run of 96 GEMM rows -- no halo.  Measured (round 5, 256 images): 212 us against
// 153 + 101 us for the two launches inside the forward, +2.5 % end to end (in-session A/B, LT_NO_XR=1).
//
//   t2 tile   [96 px][P = 256]   one LDS-DMA per tile (48 KB), row = pixel, 16-byte slot XOR-swizzled by (pixel & 15)
//   EXPANDER waves 0-3   chunk c (of 8): y[:, 128 c .. + 127] = relu(bn3(W3 t2) + res) -> chunk buffer c & 1 in LDS.  Transposed product D[channel][pixel] on the
//             32x32x16 MFMA: a lane ends up with two runs of 8 consecutive channels of ONE pixel = 16-byte LDS stores.  Wave e: channel block e of the chunk,
//             all three pixel blocks (every W3 fragment feeds three MFMAs); K = P: 16 K blocks; the residual is requested one chunk ahead.
//   REDUCER waves 4-7    chunk k: t1' += W1'[:, chunk k] y[:, chunk k] from the chunk buffer (wave r: output blocks 2 r, 2 r + 1, three pixel blocks, 8 K blocks), and
//             chunk k -> memory from the chunk buffer: 16 lanes x 16 bytes = one pixel's 256-byte run (the expanders' registers hold 32-byte pieces of 32 rows).
//   end       t1' = relu(bn1'(acc)) -> memory (reducers)
// The two roles run one chunk apart (expanders on chunk k + 1 while the reducers consume chunk k): ONE barrier per chunk, and the two waves of a SIMD are never
// in the same phase -- when an expander waits for HBM (vector-memory results return in order: its weight fragments queue behind its residual requests) the
// reducer next to it has the matrix pipe.  106 KB of LDS (t2 48 + 2 x 24 chunk buffers + 10 constants), eight waves, one workgroup per CU.
// Weights: lt_conv_pack_weights_t32 order, straight from global memory (L2) into the first MFMA operand, like bneck_kernel's phase 3; the BatchNorm constants of
// both layers come in with one LDS-DMA.  Rounding points are the separate launches': y is rounded to bf16 where the expand stored it.
//
// What the wave traces of this kernel taught (tools/xr_bench.py --trace, DESIGN.md "Round 5"):
//   * a per-store `if (row < M)` made hipcc park the rows in SCRATCH and wait vmcnt(0) in front of every store; a wave-uniform run-time "full tile" switch still
//     cost a vmcnt(0) at the join (the counter model merges conservatively) -> FULL is a template parameter, the ragged tile its own launch: 233 -> 212 us;
//   * a pointer passed through an empty asm ("+v") loses its address space: the loads behind it become FLAT loads, which count in vmcnt AND lgkmcnt -- every LDS
//     wait turns into lgkmcnt(0) + vmcnt(0);
//   * stores share the in-order vmcnt with loads on gfx950: a weight fragment requested behind a chunk's six row stores is not counted as arrived before HBM has
//     acknowledged the stores;
//   * per CU the vector-memory path moves 64 B / clock: the two roles re-read 128 KB of weight fragments per chunk (1 MB per 96-pixel tile), 42 B / clock at the
//     MFMA-bound pace -- with the residual and the row stores the kernel sits at that limit, not at HBM's (3.56 TB/s of its 755 MB);
//   * persistent workgroups (next tile's t2 rows requested under the last chunk) lost: inside a tile loop hipcc hoists ~200 tile-invariant addresses.
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

using namespace lt;

namespace {

__device__ uint4 g_zero_page_x[2];

typedef __attribute__((address_space(3))) void* lptr_t;

// -DLT_XR_TRACE: shader-clock stamps of every wave of the first LT_XR_TRACE_WG workgroups (profiling builds only: lt_build.build_variant; read back with
// lt_xr_trace_read, tools/xr_bench.py --trace).  Slot 0: entry, 1: t2 tile + constants in LDS; chunk c: 2 + 3 c, 3 + 3 c, 4 + 3 c (see the two roles).
#ifdef LT_XR_TRACE
#define LT_XR_TRACE_WG 2048
__device__ unsigned long long g_xr_trace[LT_XR_TRACE_WG * 8 * 32];
#define XR_STAMP(k)                                                                                                       \
    do {                                                                                                                  \
        if (blockIdx.x < LT_XR_TRACE_WG && lane == 0) g_xr_trace[(blockIdx.x * 8 + wave) * 32 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define XR_STAMP(k)
#endif

struct XrArgs {
    const bf16_t* t2;      // [M][P]
    const bf16_t* res;     // [M][C]
    bf16_t* y;             // [M][C]
    bf16_t* t1;            // [M][P]
    const bf16_t* w3;      // expand, t32 order of [C][P]
    const bf16_t* w1;      // reduce of the next block, t32 order of [P][C]
    const float* sc3; const float* sh3;      // [C]
    const float* sc1; const float* sh1;      // [P]
    const float* consts;                     // optional: [sc3 | sh3 | sc1 | sh1] back to back (16-byte aligned)
    int M, tile0;                            // rows; first tile of this launch
};

__device__ __forceinline__ void dma16x(const void* src, unsigned lds_base) {
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

template <int I0, int I1, typename F>
__device__ __forceinline__ void static_for_x(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for_x<I0 + 1, I1>(f);
    }
}

// FULL: every tile of the launch is complete (no row checks anywhere: a wave-uniform run-time switch made hipcc merge its counter state at the join and wait
// vmcnt(0), a per-store condition made it park the rows in scratch); the ragged last tile of a tensor runs as a second one-tile launch with FULL = false.
// NPB: pixel blocks of 32 per tile.  3 (96 pixels: 147456 rows of the benchmark = 6 whole rounds of 256 tiles) is the throughput shape -- every weight fragment
// feeds three MFMAs; 2 and 1 are for small batches, where 96-pixel tiles leave CUs without a tile (20 images = 120 tiles).
template <int C, int P, bool FULL, int NPB>
__global__ __launch_bounds__(512, 2) void xr_kernel(const XrArgs a) {
    typedef bf16_t T;
    constexpr int TM = 32 * NPB;
    constexpr int CH = 128, NCH = C / CH;                    // expand output channels per chunk / chunks
    constexpr int G3 = P / 16, G1 = CH / 16;                 // K blocks of the expand / of one chunk of the reduce
    constexpr int NOB3 = C / 32, NOB1 = P / 32;              // 32-channel output blocks of the two weight matrices
    constexpr int RB2 = 2 * P, RBY = 2 * CH;                 // bytes per LDS row of the t2 tile / of a chunk buffer
    constexpr int T2_BYTES = TM * RB2, Y_BYTES = TM * RBY;
    constexpr int Y_OFF = T2_BYTES, CST_OFF = Y_OFF + 2 * Y_BYTES;
    static_assert(P == 256 && NOB1 == 8 && CH == 128, "four expander waves = the four channel blocks of a chunk, four reducer waves x two output blocks");
    static_assert(T2_BYTES % 1024 == 0 && Y_BYTES % 1024 == 0 && (T2_BYTES / 1024) % 4 == 0, "XOR addressing / DMA pieces");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    float* cst = (float*)(smem + CST_OFF);                   // [sc3 C | sh3 C | sc1 P | sh1 P]
    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page_x;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n31 = lane & 31, hk = lane >> 5, sw = n31 & 15;
    // One tile per workgroup.  (Measured and dropped, round 5: PERSISTENT workgroups that request the next tile's t2 rows while the reducers finish the current
    // one -- the tile prologue is 17 % of a tile's time here -- were slower, 256 vs 233 us per launch: inside a tile loop hipcc hoists the ~200 tile-invariant
    // fragment / LDS addresses into registers, and with those laundered per tile the weight prefetch still had to shrink from 14 to 7 units to fit 256 VGPRs.)
    const int tile = blockIdx.x + a.tile0;
    XR_STAMP(0);

    // ---- once: the BatchNorm constants of both layers into LDS ----
    if (a.consts) {                                           // [sc3 | sh3 | sc1 | sh1] back to back: ten 1 KiB LDS-DMA pieces, no register round trip
        // issued by the EXPANDER waves only: an LDS-DMA counts in vmcnt, and only those waves wait vmcnt(0) in front of the first barrier (the reducers wait
        // lgkmcnt(0) there -- a piece issued by a reducer could still be in flight when an expander reads sh3 in the chunk-0 epilogue: ADVICE r5)
        constexpr int NPC = (2 * C + 2 * P) * 4 / 1024;
        if (wave < 4)
            for (int p = wave; p < NPC; p += 4) dma16x((const void*)((const char*)a.consts + p * 1024 + lane * 16), lds0 + CST_OFF + p * 1024);
    } else {
        for (int i = t; i < 2 * C + 2 * P; i += 512)
            cst[i] = i < C ? a.sc3[i] : i < 2 * C ? a.sh3[i - C] : i < 2 * C + P ? a.sc1[i - 2 * C] : a.sh1[i - 2 * C - P];
    }

    if (wave < 4) {
        // ======================================== EXPANDER waves: y chunk c = relu(bn3(W3 t2) + res) -> chunk buffer c & 1 ========================================
        // wave e owns channel block e of every chunk (32 channels) for all three pixel blocks: every W3 fragment feeds three MFMAs.  These waves issue ALL the
        // long-latency requests of the kernel (t2 rows, the residual one chunk ahead) next to their weight stream; the reducer waves that share their SIMDs
        // only ever wait for L2 (vector-memory results return in order: a weight fragment queued behind a residual request waits for HBM).
        const int e4 = wave;
        const T* wl3 = a.w3 + (size_t)lane * 8;               // fragment (g, ob): + (g * NOB3 + ob) * 512 elements
        unsigned a2 = lds0 + n31 * RB2 + ((hk ^ sw) << 4);   // t2 fragment: K block g: ^ (g << 5), pixel block pb: + pb * 32 * RB2
        const unsigned ldsE = lds0;
        long long prow[NPB];
        bool pok[NPB];
        uint4 rq[2][NPB][2];                                  // residual vectors [chunk parity][pixel block][run of 8 channels]
        auto load_res = [&](int c, uint4 (&dst)[NPB][2]) {
            const int ch = CH * c + 32 * e4 + 8 * hk;
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
                for (int j = 0; j < 2; ++j) dst[pb][j] = *(const uint4*)(a.res + (pok[pb] ? prow[pb] * C + ch + 16 * j : 0ll));
        };
        // the weight stream: unit u = 16 c + g, requested WD3 units ahead; WD3 + 1 divides the units of a tile, so every tile starts in ring slot 0
        constexpr int NU = NCH * G3, WD3 = 14;
                V16 wf[WD3 + 1];
        auto load_w = [&](int u) -> V16 {
            const int c = u / G3, g = u - c * G3;
            V16 v;
            v.u = *(const uint4*)(wl3 + ((size_t)g * NOB3 + 4 * c + e4) * 512);
            return v;
        };
        auto open_tile = [&](int tl) {                        // everything a tile needs before its first MFMA: t2 rows -> LDS, first fragments, first residual
            const int m0 = tl * TM;
            const T* const t2p = a.t2;
            constexpr int RPP = 1024 / RB2, LPR = RB2 / 16;   // rows per 1 KiB piece (2), lanes per row (32)
#pragma unroll
            for (int i = 0; i < T2_BYTES / 1024 / 4; ++i) {
                const int p = wave + 4 * i;
                const int row = RPP * p + lane / LPR, phys = lane % LPR;
                const int m = m0 + row;
                const void* src = m < a.M ? (const void*)(t2p + (size_t)m * P + (phys ^ (row & 15)) * 8) : zero_page;
                dma16x(src, lds0 + p * 1024);
            }
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) { prow[pb] = (long long)m0 + 32 * pb + n31; pok[pb] = prow[pb] < a.M; }
#pragma unroll
            for (int u = 0; u < WD3; ++u) wf[u] = load_w(u);
            load_res(0, rq[0]);
        };
        open_tile(tile);
        constexpr bool first = true;
        {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // this tile's t2 rows and the constants are in LDS
            if (first) XR_STAMP(1);
            static_for_x<0, NCH>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                f32x16 acc[NPB];
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[pb][e] = 0.f;
                static_for_x<0, G3>([&](auto gc) {
                    constexpr int g = decltype(gc)::value, u = G3 * c + g;
                    if constexpr (u + WD3 < NU) wf[(u + WD3) % (WD3 + 1)] = load_w(u + WD3);
                    V16 xb[NPB];
#pragma unroll
                    for (int pb = 0; pb < NPB; ++pb) xb[pb].u = *(const uint4*)((lptr_t)(size_t)((a2 ^ (g << 5)) + pb * 32 * RB2));
#pragma unroll
                    for (int pb = 0; pb < NPB; ++pb) acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u % (WD3 + 1)].h, xb[pb].h, acc[pb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                // the NEXT chunk's residual goes out HERE: vector-memory results return in order, so every weight fragment requested after this point waits
                // for HBM -- the WD3 fragments the next chunk starts with are already in flight in front of it, and this chunk's epilogue + the barrier (no
                // weight needed) pass before the first one queued behind it is due (requested at g == 0 the MFMA loop stalled ~3800 cycles per chunk: trace)
                if constexpr (c + 1 < NCH) load_res(c + 1, rq[(c + 1) & 1]);
                if (first) XR_STAMP(2 + 3 * c);                  // MFMA loop done
                // epilogue: lane (pixel n31, h) holds channels 32 ob + 16 j + 8 h + e (e < 8) of its three pixels
                const unsigned ybuf = ldsE + Y_OFF + (c & 1) * Y_BYTES;
                const int ob = 4 * c + e4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int chl = 32 * ob + 16 * j + 8 * hk;
                    const float4 s0 = *(const float4*)(cst + chl), s1 = *(const float4*)(cst + chl + 4);
                    const float4 f0 = *(const float4*)(cst + C + chl), f1 = *(const float4*)(cst + C + chl + 4);
                    const float esc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, esf[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                    for (int pb = 0; pb < NPB; ++pb) {
                        const uint4 rv = rq[c & 1][pb][j];
                        const unsigned rr[4] = {rv.x, rv.y, rv.z, rv.w};
                        unsigned o[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const int e = 2 * d;
                            const float v0 = fmaxf(acc[pb][8 * j + e] * esc[e] + esf[e] + __uint_as_float(rr[d] << 16), 0.f);
                            const float v1 = fmaxf(acc[pb][8 * j + e + 1] * esc[e + 1] + esf[e + 1] + __uint_as_float(rr[d] & 0xffff0000u), 0.f);
                            o[d] = pack_bf16x2(v0, v1);
                        }
                        const int px = 32 * pb + n31;
                        *(uint4*)((lptr_t)(size_t)(ybuf + px * RBY + (((4 * e4 + 2 * j + hk) ^ sw) << 4))) = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
                if (first) XR_STAMP(3 + 3 * c);                  // epilogue done
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // chunk c is complete in LDS (and the reducers are done with chunk c - 1)
                if (first) XR_STAMP(4 + 3 * c);                  // through the barrier
            });
        }
        return;
    }

    // ======================================== REDUCER waves: t1' += W1'[:, chunk k] y[:, chunk k], and chunk k -> memory ========================================
    // wave r owns output channel blocks 2 r, 2 r + 1 (of P / 32 = 8) for all three pixel blocks; it also writes chunk k of y to memory from the chunk buffer:
    // 16 lanes x 16 bytes = one pixel's 256-byte run per quarter wave (the expanders' own registers hold 32-byte pieces of 32 different rows).
    const int r4 = wave - 4;
    const T* wl1 = a.w1 + (size_t)lane * 8;                   // fragment (gK, ob2): + (gK * NOB1 + ob2) * 512
    const unsigned ayb = lds0 + Y_OFF + n31 * RBY + ((hk ^ sw) << 4);
    constexpr int NU1 = NCH * G1, WD1 = 4;                    // units of two fragments, requested WD1 units ahead through the chunks
    V16 wf1[WD1 + 1][2];
    auto load_w1 = [&](int u, V16 (&dst)[2]) {
#pragma unroll
        for (int o = 0; o < 2; ++o) dst[o].u = *(const uint4*)(wl1 + ((size_t)u * NOB1 + 2 * r4 + o) * 512);
    };
    // copy-out geometry: this wave's 24 rows of a chunk, four rows per instruction
    const int crow = (TM / 4) * r4 + (lane >> 4), cslot = lane & 15;
    constexpr bool first = true;
    {
        const int m0 = tile * TM;
        f32x16 acc1[2][NPB];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[o][pb][e] = 0.f;
#pragma unroll
        for (int u = 0; u < WD1; ++u) load_w1(u, wf1[u]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // the expanders' "t2 tile is in LDS" barrier
        if (first) XR_STAMP(1);
        static_for_x<0, NCH>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // chunk k is complete in LDS
            if (first) XR_STAMP(2 + 3 * k);                      // through the barrier
            const unsigned ybuf = lds0 + Y_OFF + (k & 1) * Y_BYTES;
            {
                uint4 cv[TM / 16];
#pragma unroll
                for (int i = 0; i < TM / 16; ++i) {
                    const int row = crow + 4 * i;
                    cv[i] = *(const uint4*)((lptr_t)(size_t)(ybuf + row * RBY + ((cslot ^ (row & 15)) << 4)));
                }
                if constexpr (FULL) {
                    T* yp = a.y + ((long long)m0 + crow) * C + CH * k + 8 * cslot;
#pragma unroll
                    for (int i = 0; i < TM / 16; ++i) *(uint4*)(yp + (long long)4 * i * C) = cv[i];
                } else {
#pragma unroll
                    for (int i = 0; i < TM / 16; ++i) {
                        const long long m = (long long)m0 + crow + 4 * i;
                        if (m < a.M) *(uint4*)(a.y + m * C + CH * k + 8 * cslot) = cv[i];
                    }
                }
            }
            if (first) XR_STAMP(3 + 3 * k);                      // copy-out issued
            const unsigned ay = ayb + (k & 1) * Y_BYTES;
            static_for_x<0, G1>([&](auto gc) {
                constexpr int g = decltype(gc)::value, u = G1 * k + g;
                if constexpr (u + WD1 < NU1) load_w1(u + WD1, wf1[(u + WD1) % (WD1 + 1)]);
                V16 xb[NPB];
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) xb[pb].u = *(const uint4*)((lptr_t)(size_t)((ay ^ (g << 5)) + pb * 32 * RBY));
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int pb = 0; pb < NPB; ++pb)
                        acc1[o][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[u % (WD1 + 1)][o].h, xb[pb].h, acc1[o][pb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (first) XR_STAMP(4 + 3 * k);                      // MFMA loop done
        });

        // ---- t1' = relu(bn1'(acc1)) -> memory ----
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int chl = 32 * (2 * r4 + o) + 16 * j + 8 * hk;
                const float4 s0 = *(const float4*)(cst + 2 * C + chl), s1 = *(const float4*)(cst + 2 * C + chl + 4);
                const float4 f0 = *(const float4*)(cst + 2 * C + P + chl), f1 = *(const float4*)(cst + 2 * C + P + chl + 4);
                const float esc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, esf[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    const long long m = (long long)m0 + 32 * pb + n31;
                    unsigned ov[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int e = 2 * d;
                        ov[d] = pack_bf16x2(fmaxf(acc1[o][pb][8 * j + e] * esc[e] + esf[e], 0.f), fmaxf(acc1[o][pb][8 * j + e + 1] * esc[e + 1] + esf[e + 1], 0.f));
                    }
                    if (FULL || m < a.M) *(uint4*)(a.t1 + m * P + chl) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
                }
            }
        if (first) XR_STAMP(26);                                 // t1' stored: the tile is done
    }
}

}  // namespace

#ifdef LT_XR_TRACE
extern "C" int lt_xr_trace_read(void* host, int nbytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_xr_trace), nbytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int lt_expand_reduce_fwd(const lt_xr_desc* d, const void* t2, const void* residual, void* y, void* t1_next, void* stream) {
    LT_REQUIRE(d && t2 && residual && y && t1_next, LT_ERR_INVALID, "lt_expand_reduce_fwd: null argument");
    LT_REQUIRE(d->dtype == LT_BF16, LT_ERR_UNSUPPORTED, "lt_expand_reduce_fwd: bf16 only");
    LT_REQUIRE(d->C == 1024 && d->P == 256, LT_ERR_UNSUPPORTED, "lt_expand_reduce_fwd: widths %d / %d (1024 / 256: the identity blocks of ResNet layer3)", d->C, d->P);
    LT_REQUIRE(d->M > 0 && (long long)d->M * d->C < (1ll << 40), LT_ERR_INVALID, "lt_expand_reduce_fwd: M = %lld", (long long)d->M);
    LT_REQUIRE(y != residual && y != t2 && t1_next != t2, LT_ERR_INVALID, "lt_expand_reduce_fwd: outputs must not alias the inputs (tiles run concurrently)");
    for (int i = 0; i < 2; ++i)
        LT_REQUIRE(d->weight[i] && d->scale[i] && d->shift[i], LT_ERR_INVALID, "lt_expand_reduce_fwd: layer %d: null weight / scale / shift", i);
    XrArgs a;
    a.t2 = (const bf16_t*)t2; a.res = (const bf16_t*)residual; a.y = (bf16_t*)y; a.t1 = (bf16_t*)t1_next;
    a.w3 = (const bf16_t*)d->weight[0]; a.w1 = (const bf16_t*)d->weight[1];
    a.sc3 = d->scale[0]; a.sh3 = d->shift[0]; a.sc1 = d->scale[1]; a.sh1 = d->shift[1];
    a.consts = d->consts;
    LT_REQUIRE(!d->consts || ((size_t)d->consts % 16 == 0), LT_ERR_INVALID, "lt_expand_reduce_fwd: consts must be 16-byte aligned");
    a.M = (int)d->M;
    LT_REQUIRE(d->M < (1ll << 31) - 256, LT_ERR_UNSUPPORTED, "lt_expand_reduce_fwd: row count");
    // tile height: 96 pixels once they fill most of the chip; 64 / 32 when 96-pixel tiles would leave CUs idle -- measured (forward samples/s with 96 / 64 / 32
    // pixels): 2 samples (48 tiles of 96) 446 / 461 / 470, 5 samples (120) 857 / 876 / 836, 10 samples (240) 1186 / 1135 / 1122 (LT_XR_NPB=1|2|3 forces one)
    const char* e = getenv("LT_XR_NPB");                        // A/B switch, read per launch CALL (the tests flip it in-process); a replayed hipGraph never gets here
    const long long t96 = (d->M + 95) / 96;
    const int ncu = device_cu_count8();                         // thresholds in CUs: 7/8 and 7/16 of the chip (224 / 112 tiles on the 256-CU part)
    // (round 6: between those points the tile that still fits ONE round of the chip wins -- 8 / 9 samples = 288 / 324 tiles of 64 rows spill into a second round,
    //  their 192 / 216 tiles of 96 rows do not: 1057 -> 1113 / 1063 -> 1115 samples/s; 6 / 7 samples = 216 / 252 tiles of 64 rows fit: 985 vs 962, 1079 vs 1064;
    //  profiles/r06_ab_xr_tile_height_6_to_9_samples.log)
    const long long t64 = (d->M + 63) / 64, t32 = (d->M + 31) / 32;
    const int npb = e ? (e[0] - '0') : (t96 >= ncu * 7 / 8 ? 3 : t96 >= ncu * 7 / 16 ? (t64 <= ncu ? 2 : 3) : (t32 <= ncu ? 1 : 2));
    LT_REQUIRE(npb >= 1 && npb <= 3, LT_ERR_INVALID, "lt_expand_reduce_fwd: LT_XR_NPB=%s", e ? e : "?");
    auto run = [&](auto npbc) -> int {
        constexpr int NPBH = decltype(npbc)::value, TMH = 32 * NPBH, lds = TMH * 512 + 2 * TMH * 256 + (2 * 1024 + 2 * 256) * 4;
        const long long nfull = d->M / TMH;
        if (nfull > 0) {
            auto kern = xr_kernel<1024, 256, true, NPBH>;
            LT_OPT_IN_LDS(kern, lds);
            a.tile0 = 0;
            hipLaunchKernelGGL(kern, dim3((unsigned)nfull), dim3(512), lds, (hipStream_t)stream, a);
            LT_CHECK_LAUNCH("lt_expand_reduce_fwd");
        }
        if (d->M % TMH) {
            auto kern = xr_kernel<1024, 256, false, NPBH>;
            LT_OPT_IN_LDS(kern, lds);
            a.tile0 = (int)nfull;
            hipLaunchKernelGGL(kern, dim3(1), dim3(512), lds, (hipStream_t)stream, a);
            LT_CHECK_LAUNCH("lt_expand_reduce_fwd(ragged tile)");
        }
        return LT_OK;
    };
    // (round 6, measured and removed: 128-pixel tiles -- NPB = 4, every weight fragment feeds four MFMAs instead of three, 138 KB of LDS, 4.5 rounds of the chip
    //  at 256 images -- 292 vs 213 us per seam, 1372 / 1395 vs 1464 / 1490 samples/s in two interleaved runs: the accumulators of a fourth pixel block do not
    //  fit beside the 15-deep weight ring (256 VGPRs, 26 spilled), and the spill traffic sits on the vector-memory path the kernel is bound by)
    if (npb == 3) return run(std::integral_constant<int, 3>{});
    if (npb == 2) return run(std::integral_constant<int, 2>{});
    return run(std::integral_constant<int, 1>{});
}
