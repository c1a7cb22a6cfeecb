// bneck_kernel: a whole ResNet bottleneck block in ONE launch (identity blocks of layer1 / layer2 of the pose backbone,
// reference mvn/models/pose_resnet.py:57-95: conv1x1 -> bn -> relu -> conv3x3 -> bn -> relu -> conv1x1 -> bn -> += x -> relu).
//
// Why (VERDICT r3 "next" 1): at 96^2 / 48^2 the three launches move 16 channel-units per pixel through HBM (reduce reads 4 C' and
// writes C', the 3x3 reads and writes C', the expand reads C' + the residual 4 C' and writes 4 C'; C' = the bottleneck width P) and
// run at 0.07-0.27 of the MFMA roof because of it.  Here a workgroup owns an 8 x 16 pixel tile, reads x once (10 x 18 halo) and writes
// y once: 8 units per pixel; the two intermediate tensors never leave LDS.
//
//   phase 1  t1[hp][P]  = relu(bn1(W1 . x[hp][C]))   for the 180 halo pixels (6 blocks of 32; out-of-image pixels = 0: conv2's padding)
//            x streams through an LDS-DMA ring of 32-channel stages (180 rows x 64 B, the conv_igemm7 swizzle), W1 fragments come from
//            global memory in fragment order; transposed product D[channel][pixel]: a lane ends up with two runs of 8 consecutive
//            channels of ONE pixel, which is a 16-byte LDS store into t1 (row = halo pixel, slot XOR-swizzled by the pixel index).
//   phase 2  t2[px][P]  = relu(bn2(W2 * t1))         3x3, 128 output pixels (4 blocks of 2 rows x 16 columns); the tap offset is a
//            compile-time LDS immediate, the K block an XOR on the lane's address (one v_xor per fragment), W2 fragments from global
//            memory (each used for 4 / 2 MFMAs).  Odd rows of a pixel block are rotated by two columns so that the 16 lanes of a
//            ds_read_b128 group always hit 16 different halo pixels mod 16 = 16 different swizzle classes (conflict free for every tap).
//   phase 3  y[px][C]   = relu(bn3(W3 . t2) + x)     a wave owns C / 128 output-channel blocks x all 4 pixel blocks (every W3
//            fragment feeds 4 MFMAs); residual and result move as 16-byte channel runs straight from / to global memory.
// Four waves, <= 80 KB of LDS: two workgroups per CU overlap each other's DMA / MFMA / store phases (DESIGN rule 2).
// Weights: lt_conv_pack_weights_t32 order for all three GEMMs ([tap][K / 16][Cout / 32][lane] x 16 B, rows permuted so that MFMA row
// r carries channel 16 (r >> 4) + 8 ((r >> 2) & 1) + 4 ((r >> 3) & 1) + (r & 3)).
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

using namespace lt;

namespace {

__device__ uint4 g_zero_page_b[2];

typedef __attribute__((address_space(3))) void* lptr_t;

struct BneckArgs {
    const bf16_t* x;
    bf16_t* y;
    const bf16_t* w1;
    const bf16_t* w2;
    const bf16_t* w3;
    const float* bias[3];    // may be null (ResNet convolutions carry no bias)
    const float* scale[3];
    const float* shift[3];
    int N, H, W, tiles_x, tiles_y;
};

__device__ __forceinline__ void dma16b(const void* src, unsigned lds_base) {
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

__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void frag_ready_b(V16& f) {
    f32x4 t = f.f;
    asm volatile("" : "+v"(t));
    f.f = t;
}

template <int I0, int I1, typename F>
__device__ __forceinline__ void static_for_b(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for_b<I0 + 1, I1>(f);
    }
}

// wave-uniform base in SGPRs + 32-bit lane offset (conv_igemm7's gload16: s_nop for the readfirstlane -> vector-memory hazard)
__device__ __forceinline__ void gload16b(V16& d, const void* sbase, unsigned voff) {
    f32x4 t;
    const unsigned long long b = (unsigned long long)(size_t)sbase;
    const unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)b);
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(t) : "v"(voff), "s"(ub) : "memory");
    d.f = t;
}

__device__ __forceinline__ int swz64_b(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int C, int P, int NST>
__global__ __launch_bounds__(256, 2) void bneck_kernel(const BneckArgs a) {
    typedef bf16_t T;
    constexpr int TH = 8, TW = 16, HPI = TW + 2, HROWS = (TH + 2) * HPI;   // 10 x 18 = 180 halo pixels
    constexpr int NPB = 4;                                                  // output pixel blocks of 32 (2 rows x 16 columns)
    constexpr int NCB = P / 32, NOB = C / 32, G2 = P / 16, NK1 = C / 32;
    constexpr int RB = 2 * P, NSL = P / 8;                                  // bytes / 16-byte slots per t1 / t2 row
    constexpr int T1_BYTES = HROWS * RB;
    constexpr int STAGE = HROWS * 64;                                       // a ring stage: 180 rows x 32 channels
    constexpr int RING_OFF = T1_BYTES, RING_BYTES = (NST - 1) * STAGE + 192 * 64;
    constexpr int T2_OFF = RING_OFF;                                        // t2 takes the ring's place once phase 1 is over
    constexpr int AHEAD = NST - 1;
    constexpr int NPB1 = NCB == 4 ? 6 : 3, NPB2 = NCB == 4 ? 4 : 2, NOBW = NOB / 4;
    static_assert(NCB == 2 || NCB == 4, "bottleneck width 64 or 128");
    static_assert(NPB * 32 * RB <= RING_BYTES, "t2 fits the ring");
    static_assert(NK1 % 2 == 0 && NK1 >= AHEAD, "K steps");
    static_assert(T1_BYTES % 256 == 0 && STAGE % 256 == 0, "the XOR / bank arguments assume 256-byte aligned regions");
    auto fsw = [](int hp) -> int { return NSL == 16 ? (hp & 15) : ((hp >> 1) & 7); };

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page_b;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int lin = blockIdx.x;
    {   // XCD-aware order: XCD b % 8 walks one contiguous run of tiles (neighbouring halos and the weights meet in one L2)
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = lin & 7, j = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tpi = a.tiles_x * a.tiles_y;
    const int img = lin / tpi, rem = lin - img * tpi;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;          // image coordinates of halo pixel (0, 0)
    const T* __restrict__ x = a.x;

    const int n31 = lane & 31, hk = lane >> 5;
    const int cb = NCB == 4 ? wave : (wave & 1);           // this wave's 32-channel block of the bottleneck width (phases 1 and 2)
    const int whalf = NCB == 4 ? 0 : (wave >> 1);

    // ================================================ phase 1: t1 = relu(bn1(W1 x)) on the halo =======================================
    {
        const int prow = lane >> 2;
        const int kvlog = (lane & 3) ^ swz64_b(prow);
        int dbase[3];
        bool dact[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hp = 16 * (wave + 4 * i) + prow;
            const int hr = hp / HPI, hc = hp - hr * HPI;
            const int iy = y0 + hr, ix = x0 + hc;
            const bool ok = (hp < HROWS) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            dbase[i] = ok ? ((img * a.H + iy) * a.W + ix) * C + kvlog * 8 : -1;
            dact[i] = hp < HROWS;
        }
        auto issue_piece = [&](int ks, unsigned stage_base, auto ic) {
            constexpr int I = decltype(ic)::value;
            if (dact[I]) {
                const void* src = dbase[I] >= 0 ? (const void*)(x + (dbase[I] + ks * 32)) : zero_page;
                dma16b(src, lds0 + stage_base + (wave + 4 * I) * 1024);
            }
        };
        const int hb0 = NCB == 4 ? 0 : 3 * whalf;          // first halo pixel block of this wave
        const T* w1l = a.w1 + (size_t)cb * 512;            // fragment (g, cb): + g * NCB * 512 elements; lane offset in bytes below
        const unsigned wlane = lane * 16;
        V16 fa[2][2];
        auto loadA = [&](int ks, V16 (&dst)[2]) {
            const T* p = w1l + (size_t)(2 * ks) * NCB * 512;
            gload16b(dst[0], p, wlane);
            gload16b(dst[1], p + NCB * 512, wlane);
        };
        const unsigned fo0 = n31 * 64 + (((0 + hk) ^ swz64_b(n31)) << 4);
        const unsigned fo1 = n31 * 64 + (((2 + hk) ^ swz64_b(n31)) << 4);

        f32x16 acc[NPB1];
#pragma unroll
        for (int i = 0; i < NPB1; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

        loadA(0, fa[0]);
#pragma unroll
        for (int s = 0; s < AHEAD; ++s)
            static_for_b<0, 3>([&](auto ic) { issue_piece(s, RING_OFF + s * STAGE, ic); });

        unsigned rbuf = 0, wbuf = AHEAD * STAGE;
        auto step = [&](int ks, auto rc) {
            constexpr int R = decltype(rc)::value;
            const int after = ks == 0 ? (AHEAD - 1) * 3 : (ks + AHEAD - 1 < NK1 ? 3 : 0);
            wait_vm(after);                                  // A(ks) and, older, this wave's pieces of stage ks
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            frag_ready_b(fa[R][0]);
            frag_ready_b(fa[R][1]);
            // no prefetch behind the last step: an asynchronous asm load whose result is never used leaves hipcc free to give its
            // destination registers to something live (here: an LDS fragment of the fully unrolled C = 256 loop), and the data
            // landing later overwrites it
            if (ks + 1 < NK1) loadA(ks + 1, fa[R ^ 1]);
            if (ks + AHEAD < NK1) static_for_b<0, 3>([&](auto ic) { issue_piece(ks + AHEAD, RING_OFF + wbuf, ic); });
            const unsigned rb = lds0 + RING_OFF + rbuf + hb0 * 2048;
            // all fragments of one K half are requested before its MFMAs, the second half's under the first half's MFMAs (left to itself
            // hipcc reads every fragment into the same four registers: ds_read -> lgkmcnt(0) -> MFMA, one LDS round trip per MFMA)
            V16 b0[NPB1], b1[NPB1];
#pragma unroll
            for (int i = 0; i < NPB1; ++i) b0[i].u = *(const uint4*)((lptr_t)(size_t)(rb + fo0 + i * 2048));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NPB1; ++i) b1[i].u = *(const uint4*)((lptr_t)(size_t)(rb + fo1 + i * 2048));
#pragma unroll
            for (int i = 0; i < NPB1; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[R][0].h, b0[i].h, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NPB1; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[R][1].h, b1[i].h, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rbuf = rbuf + STAGE == NST * STAGE ? 0 : rbuf + STAGE;
            wbuf = wbuf + STAGE == NST * STAGE ? 0 : wbuf + STAGE;
        };
        for (int ks = 0; ks < NK1; ks += 2) {
            step(ks, std::integral_constant<int, 0>{});
            step(ks + 1, std::integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");

        // epilogue: lane (pixel n31, h) holds channels 32 cb + 8 h + e (e < 8) and 32 cb + 16 + 8 h + (e - 8) of halo pixel 32 hb + n31
        float esc[16], esf[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * cb + 16 * (e >> 3) + 8 * hk + (e & 7);
            const float bi = a.bias[0] ? a.bias[0][c] : 0.f;
            esc[e] = a.scale[0][c];
            esf[e] = a.shift[0][c];
            if (a.bias[0]) esf[e] = bi * esc[e] + esf[e];   // (acc + b) s + f == acc s + (b s + f) up to one rounding; ResNet has no bias
        }
#pragma unroll
        for (int i = 0; i < NPB1; ++i) {
            const int hp = 32 * (hb0 + i) + n31;
            const int hr = hp / HPI, hc = hp - hr * HPI;
            const int iy = y0 + hr, ix = x0 + hc;
            const bool inimg = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            if (hp < HROWS) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    unsigned o[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int e = 8 * q + 2 * d;
                        const float v0 = fmaxf(acc[i][e] * esc[e] + esf[e], 0.f), v1 = fmaxf(acc[i][e + 1] * esc[e + 1] + esf[e + 1], 0.f);
                        o[d] = inimg ? pack_bf16x2(v0, v1) : 0u;
                    }
                    const int slot = (4 * cb + 2 * q + hk) ^ fsw(hp);
                    *(uint4*)((lptr_t)(size_t)(lds0 + hp * RB + slot * 16)) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
    __syncthreads();

    // output pixel of (block pb, lane n31): tile row 2 pb + r, column c with the odd row rotated by two columns (bank argument above)
    const int prr = n31 >> 4, pcc = ((n31 & 15) - 2 * prr) & 15;

    // ================================================ phase 2: t2 = relu(bn2(W2 * t1)) ================================================
    {
        const int pb0 = NCB == 4 ? 0 : 2 * whalf;
        const int bn = prr * HPI + pcc + 2 * HPI * pb0;      // halo pixel of tap (0, 0) of this lane's pixel in block pb0
        unsigned am[16];                                     // lane address for the tap-offset classes m = T & 15
#pragma unroll
        for (int m = 0; m < 16; ++m) am[m] = lds0 + bn * RB + ((hk ^ fsw(bn + m)) << 4);
        const T* wl = a.w2 + ((size_t)cb * 64 + lane) * 8;
        auto load_w = [&](int u) -> V16 {
            V16 v;
            v.u = *(const uint4*)(wl + (size_t)u * NCB * 512);
            return v;
        };
        constexpr int NU = 9 * G2, WD = 4;
        V16 wf[WD + 1];
#pragma unroll
        for (int u = 0; u < WD; ++u) wf[u] = load_w(u);
        f32x16 acc[NPB2];
#pragma unroll
        for (int i = 0; i < NPB2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        V16 xa[2][NPB2];
        auto load_x = [&](auto uc, V16 (&dst)[NPB2]) {
            constexpr int u = decltype(uc)::value;
            constexpr int tap = u / G2, g = u % G2, dy = tap / 3, dx = tap % 3;
            static_for_b<0, NPB2>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int TT = dy * HPI + dx + 2 * HPI * i;
                const unsigned ad = am[TT & 15] ^ (g << 5);
                dst[i].u = *(const uint4*)((lptr_t)(size_t)(ad + TT * RB));
            });
        };
        load_x(std::integral_constant<int, 0>{}, xa[0]);
        static_for_b<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (u + WD < NU) wf[(u + WD) % (WD + 1)] = load_w(u + WD);
            if constexpr (u + 1 < NU) load_x(std::integral_constant<int, u + 1>{}, xa[(u + 1) & 1]);
#pragma unroll
            for (int i = 0; i < NPB2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u % (WD + 1)].h, xa[u & 1][i].h, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });

        float esc[16], esf[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * cb + 16 * (e >> 3) + 8 * hk + (e & 7);
            esc[e] = a.scale[1][c];
            esf[e] = a.shift[1][c];
            if (a.bias[1]) esf[e] = a.bias[1][c] * esc[e] + esf[e];
        }
#pragma unroll
        for (int i = 0; i < NPB2; ++i) {
            const int px = 32 * (pb0 + i) + n31;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned o[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int e = 8 * q + 2 * d;
                    o[d] = pack_bf16x2(fmaxf(acc[i][e] * esc[e] + esf[e], 0.f), fmaxf(acc[i][e + 1] * esc[e + 1] + esf[e + 1], 0.f));
                }
                const int slot = (4 * cb + 2 * q + hk) ^ fsw(n31);
                *(uint4*)((lptr_t)(size_t)(lds0 + T2_OFF + px * RB + slot * 16)) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    __syncthreads();

    // ================================================ phase 3: y = relu(bn3(W3 t2) + x) ==============================================
    {
        unsigned a2[G2];
#pragma unroll
        for (int g = 0; g < G2; ++g) a2[g] = (lds0 + T2_OFF + n31 * RB + ((hk ^ fsw(n31)) << 4)) ^ (g << 5);
        int poff[NPB];                                       // element offset of this lane's pixel in block pb, channel 8 h
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
            poff[pb] = ((img * a.H + ty * TH + 2 * pb + prr) * a.W + tx * TW + pcc) * C + 8 * hk;
        const T* wl = a.w3 + (size_t)lane * 8;
        auto load_w = [&](int u) -> V16 {                    // unit u = q * G2 + g -> fragment (g, ob = wave + 4 q)
            const int q = u / G2, g = u - q * G2;
            V16 v;
            v.u = *(const uint4*)(wl + ((size_t)g * NOB + wave + 4 * q) * 512);
            return v;
        };
        constexpr int NU = NOBW * G2, WD = 3;
        V16 wf[WD + 1];
#pragma unroll
        for (int u = 0; u < WD; ++u) wf[u] = load_w(u);
        f32x16 acc[NPB];
        uint4 rq[NPB][2];
        static_for_b<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int q = u / G2, g = u % G2;
            const int ob = wave + 4 * q;
            if constexpr (g == 0) {
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[pb][e] = 0.f;
#pragma unroll
                    for (int j = 0; j < 2; ++j) rq[pb][j] = *(const uint4*)(x + poff[pb] + 32 * ob + 16 * j);
                }
            }
            if constexpr (u + WD < NU) wf[(u + WD) % (WD + 1)] = load_w(u + WD);
            V16 xb[NPB];
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) xb[pb].u = *(const uint4*)((lptr_t)(size_t)(a2[g] + pb * 32 * RB));
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
                acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u % (WD + 1)].h, xb[pb].h, acc[pb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g == G2 - 1) {
                float esc[16], esf[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = 32 * ob + 16 * (e >> 3) + 8 * hk + (e & 7);
                    esc[e] = a.scale[2][c];
                    esf[e] = a.shift[2][c];
                    if (a.bias[2]) esf[e] = a.bias[2][c] * esc[e] + esf[e];
                }
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned rr[4] = {rq[pb][j].x, rq[pb][j].y, rq[pb][j].z, rq[pb][j].w};
                        unsigned o[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const int e = 8 * j + 2 * d;
                            const float v0 = fmaxf(acc[pb][e] * esc[e] + esf[e] + __uint_as_float(rr[d] << 16), 0.f);
                            const float v1 = fmaxf(acc[pb][e + 1] * esc[e + 1] + esf[e + 1] + __uint_as_float(rr[d] & 0xffff0000u), 0.f);
                            o[d] = pack_bf16x2(v0, v1);
                        }
                        *(uint4*)(a.y + poff[pb] + 32 * ob + 16 * j) = make_uint4(o[0], o[1], o[2], o[3]);
                    }
            }
        });
    }
}

template <int C, int P, int NST>
int launch_bneck(const BneckArgs& a, hipStream_t s) {
    constexpr int RB = 2 * P;
    constexpr int lds = 180 * RB + (NST - 1) * 180 * 64 + 192 * 64;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    auto kern = bneck_kernel<C, P, NST>;
    LT_OPT_IN_LDS(kern, lds);
    const long long nblk = (long long)a.N * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, a);
    LT_CHECK_LAUNCH("lt_bottleneck_fwd");
    return LT_OK;
}

}  // namespace

extern "C" int lt_bottleneck_fwd(const lt_bneck_desc* d, const void* x, void* y, void* stream) {
    LT_REQUIRE(d && x && y, LT_ERR_INVALID, "lt_bottleneck_fwd: null argument");
    LT_REQUIRE(x != y, LT_ERR_INVALID, "lt_bottleneck_fwd: in-place is not possible (neighbouring tiles read each other's halo)");
    LT_REQUIRE(d->dtype == LT_BF16, LT_ERR_UNSUPPORTED, "lt_bottleneck_fwd: bf16 only");
    LT_REQUIRE((d->C == 256 && d->P == 64) || (d->C == 512 && d->P == 128), LT_ERR_UNSUPPORTED,
               "lt_bottleneck_fwd: widths %d / %d (256 / 64 and 512 / 128: the identity blocks of ResNet layer1 / layer2)", d->C, d->P);
    LT_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->H % 8 == 0 && d->W % 16 == 0, LT_ERR_UNSUPPORTED,
               "lt_bottleneck_fwd: map %d x %d (8 x 16 pixel tiles)", d->H, d->W);
    LT_REQUIRE((long long)d->N * d->H * d->W * d->C < (1ll << 31), LT_ERR_UNSUPPORTED, "lt_bottleneck_fwd: 32-bit element offsets");
    BneckArgs a;
    a.x = (const bf16_t*)x;
    a.y = (bf16_t*)y;
    a.w1 = (const bf16_t*)d->weight[0];
    a.w2 = (const bf16_t*)d->weight[1];
    a.w3 = (const bf16_t*)d->weight[2];
    for (int i = 0; i < 3; ++i) {
        LT_REQUIRE(d->weight[i] && d->scale[i] && d->shift[i], LT_ERR_INVALID, "lt_bottleneck_fwd: layer %d: null weight / scale / shift", i);
        a.bias[i] = d->bias[i];
        a.scale[i] = d->scale[i];
        a.shift[i] = d->shift[i];
    }
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.tiles_x = d->W / 16; a.tiles_y = d->H / 8;
    hipStream_t s = (hipStream_t)stream;
#ifndef LT_BNECK_NST64
#define LT_BNECK_NST64 4
#endif
    if (d->P == 64) return launch_bneck<256, 64, LT_BNECK_NST64>(a, s);
    return launch_bneck<512, 128, 3>(a, s);
}
