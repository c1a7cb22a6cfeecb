// conv_igemm7_kernel: the 288 x 256 implicit-GEMM tile of conv_igemm6 on 32x32x16 MFMAs.
//
// Why (round 2, profiles/r02_ab_igemm6.txt): with NO memory instruction in its K loop conv_igemm6 still needs 75 us where the MFMAs
// alone take 43 us, and inside the forward the layer runs at ~68 % of what 16x16x32 MFMAs can deliver at the clock the chip
// sustains -- v_mfma_f32_16x16x32_bf16 issues at 17-20 cycles per 16 KFLOP where v_mfma_f32_32x32x16_bf16 does 32 KFLOP in 32 (the
// 16x16 shape tops out at ~80 % of the 2.5 PFLOP/s peak) and reads twice the operand registers per FLOP.  144-row wave tiles do not
// divide into 32-row blocks, so the wave layout changes:
//   * eight waves, wave w owns the 32 output columns n0 + 32 w of ALL 288 rows: nine 32x32 accumulator blocks (144 registers, as before);
//   * per 32-element K step a wave reads 18 A fragments from the LDS ring (nine row blocks x two K halves; twice conv_igemm6's reads:
//     every wave now walks the whole row range) but only TWO B fragments from global memory (lt_conv_pack_weights32: one 2 KB run per
//     (K step, column block)) instead of four -- half the vector-memory traffic of the B operand, no B fragment is loaded twice;
//   * 18 MFMAs of 32 cycles per wave and step = the same 1152 cycles per SIMD and step as the 36 16x16x32 of conv_igemm6, at full rate.
// The ring (six stages of 288 rows x 64 B, swizzle kv ^ [0,2,3,1][(row >> 2) & 3]: the 32-row fragment reads are conflict free under the
// same swizzle -- lanes 0-3 / 12-15 / 20-27 of a ds_read_b128 group see the four swizzle classes), the sliced DMA issue, the B
// prefetch one step ahead and the XCD-aware tile order are conv_igemm6's.  Epilogue: nine passes of one 32 x 32 block through a
// wave-private fp32 LDS tile -> 16-byte vectors (a row's 32 columns = four lanes).
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

using namespace lt;

namespace {

__device__ uint4 g_zero_page7[2];

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

__device__ __forceinline__ void wait_vmcnt7(int n) {
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
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // conservative
    }
}

template <int IMM>
__device__ __forceinline__ void lds_read16(V16& d, unsigned addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds_read offset field");
    f32x4 t;
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
// wave-uniform base in SGPRs + 32-bit lane offset + immediate (see conv_igemm3.hip gload16: s_nop for the readfirstlane hazard)
template <int IMM>
__device__ __forceinline__ void gload16(V16& d, const void* sbase, unsigned voff) {
    static_assert(IMM >= 0 && IMM < 4096, "global_load immediate offset");
    f32x4 t;
    const unsigned long long b = (unsigned long long)(size_t)sbase;
    const unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)b);
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(t) : "v"(voff), "s"(ub), "n"(IMM) : "memory");
    d.f = t;
}

__device__ __forceinline__ int swz64_7(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_igemm7_kernel(const ConvArgs a) {
    typedef bf16_t T;
    constexpr bool PW = MODE == 1 || MODE == 3;          // MODE 3: pointwise over TWO sources (ConvArgs::x2): the second one may be a strided map
    constexpr bool PW2 = MODE == 3;
    constexpr int BM = 288, BN = 256, NW = 8, SM = 9, VEC = 8, BK = 32, ROWB = 64, NST = 6;
    constexpr int NPA = BM / 16;                          // 18 DMA pieces of 1 KiB per stage (16 rows of 64 B each)
    constexpr int A_IT = (NPA + NW - 1) / NW;             // 3 (waves 0, 1) / 2
    constexpr int STAGE = BM * ROWB;                      // 18432 B
    constexpr int REGION = NST * STAGE;
    constexpr int AHEAD = NST - 1;
    constexpr int EP_LD = 32 + 4, EP_WAVE = 32 * EP_LD * 4;   // one 32 x 32 block per pass
    static_assert(NW * EP_WAVE <= REGION, "epilogue staging");
    typedef f32x16 acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int4* s_taps = (int4*)(smem + REGION);               // [ntaps] (unused when PW)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;

    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page7;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
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
    const int kv = (lane & 3) ^ swz64_7(prow);
    const bool a_tail = wave < NPA % NW;                 // waves 0, 1 own a third piece
    const int dps = (A_IT - 1) + (a_tail ? 1 : 0);       // 3 or 2 DMA pieces per wave and stage
    int id0[PW ? 1 : A_IT], ih0[PW ? 1 : A_IT], iw0[PW ? 1 : A_IT], baseC[A_IT], cur[PW ? 1 : A_IT], base2[PW2 ? A_IT : 1];
    const T* __restrict__ x2 = (const T*)a.x2;
    const int nk1 = a.Cin / BK;                          // PW2: K steps served by x
    if (!PW)
        for (int i = t; i < ph.ntaps; i += 64 * NW) s_taps[i] = ph.taps[i];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + 16 * (wave + NW * i) + prow;
        const bool own = i < A_IT - 1 || a_tail;
        if (PW) {
            baseC[i] = (own && m < a.M) ? m * a.Cin + kv * VEC : -1;
            if constexpr (PW2) {
                base2[i] = -1;
                if (own && m < a.M) {
                    int n, od, oh, ow;
                    decode_row(a, m, n, od, oh, ow);
                    base2[i] = ((n * a.H2 + oh * a.s2) * a.W2 + ow * a.s2) * a.Cin2 + kv * VEC;
                }
            }
        } else if (own && m < a.M) {
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
    // B fragments of this wave: column block n0 / 32 + wave, K step ks -> ((ks * cout_pad / 32 + block) * 2 + kk) * 1024 + 16 lane bytes
    const size_t wstep = (size_t)a.tiles_n * (BN / 32) * 2 * 64 * VEC;   // elements per K step
    const T* wfrag = (const T*)ph.wfrag32 + (size_t)(n0 / 32 + wave) * 2 * 64 * VEC;
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
        if constexpr (PW2) src = baseC[P] < 0 ? zero_page : ks < nk1 ? (const void*)(x + (baseC[P] + ks * BK)) : (const void*)(x2 + (base2[P] + (ks - nk1) * BK));
        else if constexpr (PW) src = baseC[P] >= 0 ? (const void*)(x + (baseC[P] + ks * BK)) : zero_page;
        else src = cur[P] >= 0 ? (const void*)(x + (cur[P] + c0s)) : zero_page;
        dma16(src, lds0 + sbuf + (wave + NW * P) * 1024);
    };

    // A fragment of row block i, K half kk: lane (r = lane & 31, h = lane >> 5) reads row 32 i + r, K vector 2 kk + h
    const int r31 = lane & 31, hk = lane >> 5;
    const unsigned fo0 = r31 * ROWB + (((0 + hk) ^ swz64_7(r31)) << 4);   // kk = 0
    const unsigned fo1 = r31 * ROWB + (((2 + hk) ^ swz64_7(r31)) << 4);   // kk = 1

    acc_t acc[SM];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    V16 fb[2][2];                                         // [register set = ks & 1][kk]
    gload16<0>(fb[0][0], wfrag, wlane);
    gload16<1024>(fb[0][1], wfrag, wlane);
#pragma unroll
    for (int sgi = 0; sgi < AHEAD; ++sgi)
        if (sgi < nk) {
            stage_prep(sgi);
            static_for<0, A_IT>([&](auto pc) { stage_piece(sgi, sgi * STAGE, pc); });
        }

    constexpr int NRD = 2 * SM;                           // 18 A fragment reads per step: u = kk * 9 + i
    constexpr int LOOK = 3, RA = LOOK + 1;
    unsigned rbuf = 0, wbuf = AHEAD * STAGE;
    auto step = [&](int ks, auto rc) {
        constexpr int R = decltype(rc)::value;
        const int after = ks == 0 ? ((nk < AHEAD ? nk : AHEAD) - 1) * dps : (ks + AHEAD - 1 < nk ? dps : 0);
        wait_vmcnt7(after);                               // B(ks) and, older, this wave's pieces of stage ks
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        frag_ready(fb[R][0]); frag_ready(fb[R][1]);
        {
            const T* wn1 = wfrag + (size_t)(ks + 1 < nk ? ks + 1 : ks) * wstep;
            gload16<0>(fb[R ^ 1][0], wn1, wlane);
            gload16<1024>(fb[R ^ 1][1], wn1, wlane);
        }
        const bool more = ks + AHEAD < nk;
        if (more) stage_prep(ks + AHEAD);
        const unsigned ab0 = lds0 + rbuf + fo0, ab1 = lds0 + rbuf + fo1;
        V16 fa[RA];
        auto issue = [&](auto kc) {
            constexpr int K = decltype(kc)::value, KK = K / SM, I = K % SM;
            lds_read16<I * 32 * ROWB>(fa[K % RA], KK ? ab1 : ab0);
        };
        static_for<0, NRD>([&](auto uc) {
            constexpr int u = decltype(uc)::value, KK = u / SM, I = u % SM;
            constexpr int prev_target = u == 0 ? 0 : (u + LOOK < NRD ? u + LOOK : NRD);
            constexpr int target = u + 1 + LOOK < NRD ? u + 1 + LOOK : NRD;
            static_for<prev_target, target>([&](auto kc) { issue(kc); });
            lgkm_wait<target - u - 1>();
            frag_ready(fa[u % RA]);
            acc[I] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u % RA].h, fb[R][KK].h, acc[I], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u % 4 == 1 && u / 4 < A_IT) {   // one DMA piece of stage ks+AHEAD behind fragments 1, 5, 9
                if (more) stage_piece(ks + AHEAD, wbuf, std::integral_constant<int, u / 4>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        rbuf = rbuf + STAGE == REGION ? 0 : rbuf + STAGE;
        wbuf = wbuf + STAGE == REGION ? 0 : wbuf + STAGE;
    };
    for (int ks = 0; ks < nk; ks += 2) {                  // nk is even (k_pad % 64 == 0)
        step(ks, std::integral_constant<int, 0>{});
        step(ks + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring becomes the epilogue staging area

    // ---- epilogue: nine passes of one 32 x 32 block through this wave's private fp32 LDS tile -> 16-byte vectors ----
    const int colv = n0 + wave * 32 + (lane & 3) * 8;    // this lane's eight output channels
    const EpiFloors fl = epi_floors(a.flags);
    const bool has_res = a.res != nullptr;
    float* ep = (float*)(smem + wave * EP_WAVE);
    const int colj = n0 + wave * 32 + r31;               // < cout_pad: the constant arrays are padded
    const float bi = a.bias ? a.bias[colj] : 0.f, sc = a.scale ? a.scale[colj] : 1.f, sf = a.shift ? a.shift[colj] : 0.f;
    auto out_off = [&](int i, int k) -> long long {
        const int m = m0 + i * 32 + k * 16 + (lane >> 2);
        if (m >= a.M || colv >= a.Cout) return -1;
        long long pix = m;
        if (!PW) {
            int n, od, oh, ow;
            decode_row(a, m, n, od, oh, ow);
            pix = ((long long)(n * a.OD + od * a.osd + ph.ood) * a.OH + oh * a.osh + ph.ooh) * a.OW + ow * a.osw + ph.oow;
        }
        return pix * a.ldc + colv;
    };
#pragma unroll
    for (int i = 0; i < SM; ++i) {
        long long off[2];
        uint4 rv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {                    // this pass's residual vectors first: independent round trips
            off[k] = out_off(i, k);
            rv[k] = (has_res && off[k] >= 0) ? *(const uint4*)((const T*)a.res + off[k]) : make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)                     // C layout of the 32x32 MFMA: row 8 (e >> 2) + 4 (lane >> 5) + (e & 3), column lane & 31
            ep[(8 * (e >> 2) + 4 * hk + (e & 3)) * EP_LD + r31] = (acc[i][e] + bi) * sc + sf;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (off[k] < 0) continue;
            const float* src = ep + (k * 16 + (lane >> 2)) * EP_LD + (lane & 3) * 8;
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

// lt_conv_fwd packing [cout_pad][k_pad] -> B-fragment order of the 32x32x16 MFMA: [k_pad / 32][cout_pad / 32][2 K halves][64 lanes][8];
// lane l of fragment (step, block, kk) holds column 32 block + (l & 31), K elements 32 step + 16 kk + 8 (l >> 5) .. + 7
__global__ void conv_pack_b32_kernel(const bf16_t* __restrict__ w, int cout_pad, int k_pad, bf16_t* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // ((step * cout_pad / 32 + block) * 2 + kk) * 64 + lane
    const long long total = (long long)(k_pad / 32) * (cout_pad / 32) * 2 * 64;
    if (g >= total) return;
    const int l = (int)(g & 63);
    const int kk = (int)((g >> 6) & 1);
    const long long fb = g >> 7;
    const int block = (int)(fb % (cout_pad / 32)), stepk = (int)(fb / (cout_pad / 32));
    *(uint4*)(out + g * 8) = *(const uint4*)(w + (size_t)(32 * block + (l & 31)) * k_pad + 32 * stepk + 16 * kk + 8 * (l >> 5));
}

template <int MODE>
int launch7(ConvArgs a, int cout_pad, int max_taps, hipStream_t s) {
    a.tiles_n = cout_pad / 256;
    const long long nblk = cdiv(a.M, 288) * a.tiles_n;
    LT_REQUIRE(nblk < (1ll << 31), LT_ERR_INVALID, "lt_conv_fwd: grid too large");
    const size_t lds = 6 * (size_t)288 * 64 + (size_t)max_taps * sizeof(int4);
    LT_REQUIRE(lds <= 160 * 1024, LT_ERR_UNSUPPORTED, "lt_conv_fwd: 288x256 tile needs %zu B of LDS", lds);
    auto kern = conv_igemm7_kernel<MODE>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(v7)");
    return LT_OK;
}

}  // namespace

namespace lt {
// 1 = launched, 0 = not applicable, < 0 = error.  Called by conv3_try where conv_igemm6's 288-row variant would run.
int conv7_try(const ConvArgs& a, int cout_pad, int max_taps, bool pw, hipStream_t s) {
    if (!a.phase[0].wfrag32 || cout_pad % 256 || a.k_pad % 64) return 0;
    if (a.x2 && !pw) return 0;
    const int rc = a.x2 ? launch7<3>(a, cout_pad, max_taps, s) : pw ? launch7<1>(a, cout_pad, max_taps, s) : launch7<2>(a, cout_pad, max_taps, s);
    return rc == LT_OK ? 1 : rc;
}
}  // namespace lt

extern "C" int lt_conv_pack_weights32(const void* weight, int32_t cout_pad, int32_t k_pad, void* packed, void* stream) {
    LT_REQUIRE(weight && packed, LT_ERR_INVALID, "lt_conv_pack_weights32: null argument");
    LT_REQUIRE(cout_pad >= 32 && cout_pad % 32 == 0 && k_pad >= 32 && k_pad % 32 == 0, LT_ERR_INVALID,
               "lt_conv_pack_weights32: cout_pad %d / k_pad %d (multiples of 32; bf16 weights [cout_pad][k_pad])", cout_pad, k_pad);
    const long long total = (long long)(k_pad / 32) * (cout_pad / 32) * 2 * 64;
    hipLaunchKernelGGL(conv_pack_b32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)weight,
                       cout_pad, k_pad, (bf16_t*)packed);
    LT_CHECK_LAUNCH("lt_conv_pack_weights32");
    return LT_OK;
}
