// Stride-1 "same" 3D convolution (3^3 / 7^3) for the V2V hourglass: the input HALO TILE lives in LDS.
//
// The implicit-GEMM kernels stream one K step of the im2col matrix per barrier, i.e. every input voxel crosses
// L2 -> LDS once per filter tap (27x / 343x).  At the 64^3 / 32^3 levels of V2V the GEMM is narrow (16..64 output
// channels), so that stream -- not the MFMAs -- sets the time (measured: 3^3 32->32 at 64^3 = 350 TF/s, 7^3 = 330 TF/s).
// Here a workgroup owns a TD x TH x TW block of output voxels (256 GEMM rows), DMAs the (T+K-1)^3 input halo into
// LDS ONCE (zero filled outside the volume), and then walks the taps: the A fragment of tap (kd,kh,kw) is the same
// LDS image read at a shifted voxel index, so the per-tap traffic is LDS -> VGPR only.  Weights stream through a
// double-buffered LDS ring, a few taps per barrier.
//
// LDS images (both filled by LDS-DMA, so both are lane-linear and swizzled on the SOURCE side):
//   halo   : voxel-major (row pitch PW voxels), CINB = Cin*sizeof(T) bytes per voxel = NVV 16-byte vectors; vector lv of the
//            voxel at halo position (hd,hh,hw) is stored in slot lv ^ f, f = (FA*hh + FB*hd + ((hw + FC*hh) >> FSH)) % NVV.
//            The constants per (kernel size, CINB, MFMA shape) were found by exhaustive search over all taps and all
//            ds_read_b128 lane groups (tools/halo_bank_search.py): every A-fragment read is bank-conflict free (the
//            linear (hv/VPR)%NVV swizzle of the first version was 3-way conflicted: rows of a lane group are 4 partial
//            lines of the tile, not 16 consecutive voxels);
//   weights: per tap a [cout_pad][CINB] slab, slot lv ^ ((-(col/VPR)) % NVV) (conflict free for both MFMA shapes).
// Weight ring: NBUF chunk buffers, NBUF-1 chunks of DMA in flight (counted vmcnt), one barrier per chunk.
//
// Kernels in this file (dispatch: conv3d_halo_try at the end):
//   conv3d_halo_kernel          one tile per workgroup, weight ring in LDS, optional loader waves: fp32, and the bf16 shapes below miss
//   conv3d_halo_persist_kernel  3^3 32->32: persistent, weights resident, double-buffered halo (small grids)
//   conv3d_halo_col_kernel      3^3 32->32: column walk, ring of 4-plane halo groups, epilogue under the next tile's MFMAs
//   conv3d_halo7b_kernel        7^3 32->16: loader waves, kd-register-blocked
//   conv3d_halo_wreg_kernel     3^3 64->64, 32->64, 128->128: halo-only LDS, weights as fragments from global memory
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

#ifndef LT_ACC64_MASK
#define LT_ACC64_MASK 3          // fp32 (parity) kernels: fp64 flush of the MFMA accumulators every LT_ACC64_MASK + 1 taps / K steps
#endif

using namespace lt;

namespace {

__device__ uint4 g_zero_page_h[4];   // 64 zero bytes: DMA source of padding voxels, and the 'residual' of layers without one

#ifdef LT_TRACE
// profiling build: shader-clock accounting of the persistent kernel's per-tile phases, written by wave 0 of every 8th workgroup:
// [total, top wait+barrier, halo issue, residual issue, tap loop, barrier, epilogue, tiles]
__device__ long long g_trace_h[8 * 64];
#define LT_CLKH() ((long long)__builtin_amdgcn_s_memtime())
#endif

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16h(const void* src, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);   // wave-uniform by construction; make it provable
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

// ---- hand-scheduled LDS fragment reads -------------------------------------------------------------------------------------
// hipcc's own s_waitcnt insertion degrades to lgkmcnt(0) as soon as more than one tap of fragment reads is in flight (seen
// in the ISA: every third tap drained the whole queue).  The deep-lookahead paths therefore issue ds_read_b128 themselves
// and wait with an explicit count; frag_ready() ties the wait to the registers so that no MFMA can be scheduled above it.
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
    f32x4 t = f.f;   // a native vector: HIP's uint4 is a struct, which inline asm can only take indirectly
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

// swizzle constants {FA, FB, FC, FSH, pitch multiple}
template <int KS, int CINB, int MF> struct HaloSwz { static constexpr int FA = 0, FB = 0, FC = 0, FSH = 1, PAD = 1; };   // (3, 64 B, 32x32)
template <> struct HaloSwz<7, 64, 16> { static constexpr int FA = 0, FB = 0, FC = 0, FSH = 0, PAD = 1; };
template <> struct HaloSwz<3, 128, 32> { static constexpr int FA = 3, FB = 0, FC = 2, FSH = 1, PAD = 1; };
template <> struct HaloSwz<3, 32, 32> { static constexpr int FA = 0, FB = 0, FC = 0, FSH = 2, PAD = 4; };
// (7, 32 B, 32x32) -- round 6: the input gradient of V2V's 7^3 front layer, 16 -> 32 -- takes the default: the 7^3 tap loop needs a swizzle that depends on kw only,
// and the best of those is 2-way conflicted (tools/halo_bank_search.py: 2.0 for every kw-only candidate; the conflict-free ones mix in kh)

template <typename T, int KS, int CIN, int CP, int TD, int TH, int TW, int TPC, int NBUF>
struct HaloCfg {
    static constexpr int ES = sizeof(T);
    static constexpr int VEC = 16 / ES;
    // outputs / residuals are the operand type, except that fp8 (an operand type only: the 'fp8v2v' training step) stores bf16
    typedef typename std::conditional<sizeof(T) == 1, bf16_t, T>::type OT;
    static constexpr int OVEC = 16 / (int)sizeof(OT);
    static constexpr int CINB = CIN * ES;
    static constexpr int NVV = CINB / 16;          // 16-byte vectors per voxel
    static constexpr int VPR = 16 / NVV;           // voxels per 256-byte bank row
    static constexpr int MF = CP == 16 ? 16 : 32;
    typedef HaloSwz<KS, CINB, MF> SW;
    static constexpr int HD = TD + KS - 1, HH = TH + KS - 1, HW = TW + KS - 1;
    static constexpr int PW = (HW + SW::PAD - 1) / SW::PAD * SW::PAD;       // row pitch in voxels
    static constexpr int HV = HD * HH * PW;        // halo voxel slots
    static constexpr int HALO_BYTES = ((HV * CINB + 1023) / 1024) * 1024;   // whole DMA wave-instructions
    static constexpr int NTAPS = KS * KS * KS;
    static constexpr int NCH = (NTAPS + TPC - 1) / TPC;
    static constexpr int SLAB = CP * CINB;         // bytes of one tap's weights
    static constexpr int WCH = ((TPC * SLAB + 1023) / 1024) * 1024;
    static constexpr int G = MF == 32 ? NVV / 2 : NVV / 4;   // fragment groups per tap (K = Cin)
    static constexpr int SM = 64 / MF;             // sub-tiles per wave along M (wave = 64 rows)
    static constexpr int SN = CP / MF;
    static constexpr int NACC = MF == 32 ? 16 : 4;
    static constexpr int EP_LD = CP + 4;
    static constexpr int EP_BYTES = 4 * 64 * EP_LD * 4;
    static constexpr int MAIN_BYTES = HALO_BYTES + NBUF * WCH;
    static constexpr int LDS_BYTES = MAIN_BYTES > EP_BYTES ? MAIN_BYTES : EP_BYTES;
    static_assert(TD * TH * TW == 256, "256 rows per workgroup");
    static_assert(NVV >= 1 && (MF == 32 ? NVV >= 2 : NVV >= 4), "Cin too small for the MFMA K");
    static_assert((NVV & (NVV - 1)) == 0 && NVV <= 16, "NVV must be a power of two <= 16");
    static_assert(NBUF >= 2 && NBUF <= 4, "2..4 weight chunk buffers");
    static __device__ __forceinline__ int fswz(int hd, int hh, int hw) {
        return (SW::FA * hh + SW::FB * hd + ((hw + SW::FC * hh) >> SW::FSH)) & (NVV - 1);
    }
};

// s_waitcnt vmcnt(n), n wave-uniform
__device__ __forceinline__ void wait_vmcnt_h(int n) {
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
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // conservative
    }
}

#ifdef LT_ABL_NO_MMA
#define LT_HMMA(c_, a_, b_) (void)0
#else
#define LT_HMMA(c_, a_, b_) Mma<T, MF>::run(c_, a_, b_)
#endif

// All MFMAs of one tap for a wave's SM x SN accumulator blocks over G fragment groups.  bf16: one MFMA per (group, block).  fp32 (round 6): a 16-byte
// fragment is FOUR exact-fp32 MFMAs (K pairs e = 0 .. 3); issued back to back on one accumulator they wait for each other (~10 % below the issue rate,
// measured in conv_igemm2's K loop: 4740 cycles for 64 MFMAs of 64) -- the K pair goes outermost, so that consecutive MFMAs write different blocks.  The
// summation order of every accumulator is unchanged (results are bit-identical).  -DLT_FP32_NO_PIPE: the old order (A/B builds).
template <typename T, int MF, int SM, int SN, int G, typename ACC, typename FA, typename FB>
__device__ __forceinline__ void mma_tap_blocks(ACC (&acc)[SM][SN], const FA& fa, const FB& fb) {
#if !defined(LT_ABL_NO_MMA)
#if !defined(LT_FP32_NO_PIPE)
    if constexpr (sizeof(T) == 4 && SM * SN > 1) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < SM; ++i)
#pragma unroll
                    for (int j = 0; j < SN; ++j) {
                        if constexpr (MF == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][i].f[e], fb[g][j].f[e], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[g][i].f[e], fb[g][j].f[e], acc[i][j], 0, 0, 0);
                    }
        return;
    }
#endif
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int j = 0; j < SN; ++j) Mma<T, MF>::run(acc[i][j], fa[g][i], fb[g][j]);
#endif
}

struct HaloArgs {
    const void* x;
    const void* w;      // [cout_pad][k_pad], k = tap*Cin + ci (the lt_conv_fwd packing)
    const void* wfrag;  // the same weights packed by lt_conv_pack_weights_t32 (conv3d_halo_wreg_kernel), or null
    void* y;
    const void* res;
    const float* bias;
    const float* scale;
    const float* shift;
    int N, D, H, W, Cout, ldc, k_pad, flags;
    int tiles_d, tiles_h, tiles_w;   // per sample
    int xcd_pin;
    const void* skip_x;   // lt_conv_skip_fwd (column-walk kernel only): the residual is W_skip . skip_x[voxel] (16 channels per voxel), or null
    const void* skip_w;   // its weights in lt_conv_pack_weights_t32 order ([32][16] -> one fragment)
};

// per-column epilogue constants of the lane (column j*MF + lane % MF), loaded once per kernel, well before they are needed
template <int SN>
struct HaloCst {
    float bi[SN], sc[SN], sf[SN];
    __device__ __forceinline__ void load(const HaloArgs& a, int lane, int MF) {
#pragma unroll
        for (int j = 0; j < SN; ++j) {
            const int colj = j * MF + (lane & (MF - 1));   // < cout_pad: the constant arrays are padded
            bi[j] = a.bias ? a.bias[colj] : 0.f;
            sc[j] = a.scale ? a.scale[colj] : 1.f;
            sf[j] = a.shift ? a.shift[colj] : 0.f;
        }
    }
};

// Epilogue shared by the halo kernels (same scheme as conv_igemm2): (acc + bias)*scale + shift into this wave's fp32 LDS tile
// (64 rows, padded), then 16-byte vectors: optional pre-activation ReLU, residual, ReLU, store.  rp0..rp7: the lane's residual
// vectors when they were prefetched (pre_res), in named registers (an array was kept in scratch memory by hipcc).
template <typename T, int CP, int MF, int SM, int SN, int NACC, int TH, int TW, bool ACC64, typename ACC, typename DACC>
__device__ __forceinline__ void halo_epilogue(unsigned char* smem, const HaloArgs& a, int wave, int lane, int n, int d0, int h0, int w0,
                                              ACC& acc, DACC& dacc, bool pre_res, uint4 rp0, uint4 rp1, uint4 rp2, uint4 rp3, uint4 rp4,
                                              uint4 rp5, uint4 rp6, uint4 rp7, const HaloCst<SN>& cst) {
    constexpr int EP_LD = CP + 4;
    typedef typename std::conditional<sizeof(T) == 1, bf16_t, T>::type OT;          // fp8 operands: bf16 outputs and residuals
    constexpr int VEC_ = 16 / (int)sizeof(OT);
    // ---- epilogue (same scheme as conv_igemm2: per-wave fp32 LDS tile -> 16-byte vectors) ----
    float* ep = (float*)(smem + wave * (64 * EP_LD * 4));
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int colj = j * MF + (lane & (MF - 1));
        const float bi = cst.bi[j], sc = cst.sc[j], sf = cst.sf[j];
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                const int r = i * MF + ((MF == 32) ? ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) : ((lane >> 4) * 4 + e));
                float val;
                if (ACC64) val = (float)((dacc[i][j][e] + (double)bi) * (double)sc + (double)sf);
                else val = (acc[i][j][e] + bi) * sc + sf;
                ep[r * EP_LD + colj] = val;
            }
    }
    // (the tile is wave-private and LDS operations of one wave execute in order: no barrier)

    const EpiFloors fl = epi_floors(a.flags);
    const bool has_res = a.res != nullptr;
    auto row_pix = [&](int r) -> size_t {   // r = row inside the workgroup tile
        const int tw = r % TW, th = (r / TW) % TH, td = r / (TW * TH);
        return (((size_t)n * a.D + d0 + td) * a.H + h0 + th) * a.W + w0 + tw;
    };
    constexpr int VECO = VEC_;              // fp32: 4 channels, bf16: 8 channels per 16 bytes
    if ((a.Cout % VECO == 0) && (a.ldc % VECO == 0)) {
        constexpr int LPR = CP / VECO, RPP = 64 / LPR;
        static_assert(RPP % TW == 0 && TW * TH == 64, "a wave owns one d-plane of the tile; an iteration advances whole rows of it");
        const int cq = (lane % LPR) * VECO;
        if (cq < a.Cout) {
            constexpr int NIT = 64 / RPP;
            // the wave's 64 rows are the d-plane td = wave of the tile: row lr + it*RPP -> (th, tw) = (lr / TW + it*RPP/TW, lr % TW),
            // so the element offset of iteration `it` is off0 + it * step (one 64-bit multiply per tile, not per row)
            const int lr = lane / LPR;
            const size_t off0 = ((((size_t)n * a.D + d0 + wave) * a.H + h0 + lr / TW) * a.W + w0 + lr % TW) * a.ldc + cq;
            const size_t step = (size_t)(RPP / TW) * a.W * a.ldc;
            const unsigned no_res = has_res ? 0u : 0x80008000u;   // zeros -> -0.0 pairs: v + -0.0 == v
            auto row = [&](int it, uint4 resv) {      // resv: this row's residual vector (zeros when there is none)
                const size_t off = off0 + it * step;
                const float* src = ep + (lr + it * RPP) * EP_LD + cq;
                uint4 ov;
                if constexpr (sizeof(OT) == 4) {
                    const float4 q = *(const float4*)src;
                    const float nr = has_res ? 0.f : -0.0f;
                    float4 o;
                    o.x = epi_apply(q.x, fl, has_res ? __uint_as_float(resv.x) : nr);
                    o.y = epi_apply(q.y, fl, has_res ? __uint_as_float(resv.y) : nr);
                    o.z = epi_apply(q.z, fl, has_res ? __uint_as_float(resv.z) : nr);
                    o.w = epi_apply(q.w, fl, has_res ? __uint_as_float(resv.w) : nr);
                    ov = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w));
                } else {
                    const float4 q0 = *(const float4*)src, q1 = *(const float4*)(src + 4);
                    const float vv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    const unsigned ru[4] = {resv.x | no_res, resv.y | no_res, resv.z | no_res, resv.w | no_res};
                    unsigned ou[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ou[e] = pack_bf16x2(epi_apply(vv[2 * e], fl, __uint_as_float(ru[e] << 16)),
                                            epi_apply(vv[2 * e + 1], fl, __uint_as_float(ru[e] & 0xffff0000u)));
                    ov = make_uint4(ou[0], ou[1], ou[2], ou[3]);
                }
#ifdef LT_ABL_NO_STORE
                if (a.N < 0)
#endif
                *(uint4*)((OT*)a.y + off) = ov;
            };
            if (pre_res || !has_res) {
                if (NIT > 0) row(0, rp0);
                if (NIT > 1) row(1, rp1);
                if (NIT > 2) row(2, rp2);
                if (NIT > 3) row(3, rp3);
                if (NIT > 4) row(4, rp4);
                if (NIT > 5) row(5, rp5);
                if (NIT > 6) row(6, rp6);
                if (NIT > 7) row(7, rp7);
                if (NIT > 8) {   // fp32 with a narrow tile: no prefetch (PRE_OK false), rows 8.. have no residual here
#pragma unroll
                    for (int it = 8; it < NIT; ++it) row(it, make_uint4(0, 0, 0, 0));
                }
            } else {
                uint4 rv[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it)      // all residual loads first: independent HBM round trips
                    rv[it] = *(const uint4*)((const OT*)a.res + off0 + it * step);
#pragma unroll
                for (int it = 0; it < NIT; ++it) row(it, rv[it]);
            }
        }
    } else {
        for (int idx = lane; idx < 64 * CP; idx += 64) {
            const int r = idx / CP, cc = idx - r * CP;
            if (cc >= a.Cout) continue;
            const size_t off = row_pix(64 * wave + r) * a.ldc + cc;
            const float rr = has_res ? elt<OT>::ld((const OT*)a.res + off) : -0.0f;
            elt<OT>::st((OT*)a.y + off, epi_apply(ep[r * EP_LD + cc], fl, rr));
        }
    }
}

// LDR: eight waves, waves 4-7 are loaders (halo + weight chunk ring, nothing else); waves 0-3 then issue no DMA and wait on
// no vmcnt inside the tap loop (same idea as the 7^3 and the persistent kernels; non-ring path only)
// NPH > 1 (round 6, fp32 7^3 32 -> 16): CHANNEL PHASES.  The tensor has CIN * NPH channels per voxel; the halo tile of all of them does not fit LDS
// (10 x 14 x 14 voxels x 128 B = 245 KB), the tile of CIN of them does (123 KB, the byte geometry of the bf16 kernel).  The kernel walks all taps over
// channels [0, CIN), reloads the halo image with channels [CIN, 2 CIN) and walks the taps again; the accumulators live across the phases.  Every input
// voxel enters LDS once per tile and phase -- on the generic 256 x 16 tile it crossed L2 -> LDS once per TAP (343 times), and the eight-piece DMA issue
// per K step cost as much as its 32 MFMAs (51 % of the fp32 MFMA peak).
template <typename T, int KS, int CIN, int CP, int TD, int TH, int TW, int TPC, int NBUF, int PD, bool LDR, int NPH = 1>
__global__ __launch_bounds__(LDR ? 512 : 256) void conv3d_halo_kernel(const HaloArgs a) {
    typedef HaloCfg<T, KS, CIN, CP, TD, TH, TW, TPC, NBUF> C;
    static_assert(!LDR || PD == 1, "loader waves: non-ring path");
    static_assert(NPH == 1 || !LDR, "channel phases: no loader waves");
    constexpr int XLD = CIN * NPH;                       // channels per voxel of the tensor (and per tap of a weight row)
    // fp64 flush of the fp32 accumulators: every 4 taps of 32 channels = after 128 products; the 16-channel 7^3 and two-phase kernels flush every 8 taps -- the same 128
    // products (the 16 cvt + 16 v_add_f64 of a flush next to the 16 MFMAs of a 16-channel tap cost 10 % at every fourth tap)
    constexpr int AMASK = (sizeof(T) == 4 && CIN <= 16 && (KS == 7 || NPH > 1)) ? 2 * LT_ACC64_MASK + 1 : LT_ACC64_MASK;
    int cph = 0;                                         // the channel phase whose halo / weight images are in LDS (or on their way)
    constexpr bool ACC64 = sizeof(T) == 4;
    constexpr int MF = C::MF, SM = C::SM, SN = C::SN, G = C::G, NACC = C::NACC, NVV = C::NVV, VPR = C::VPR, CINB = C::CINB;
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned char* s_halo = smem;
    unsigned char* s_w = smem + C::HALO_BYTES;

    // zero-page address made opaque once (else each use is an s_load from the GOT + s_waitcnt lgkmcnt(0) in the tap loop)
    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page_h;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool is_loader = LDR && wave >= 4;
    const bool do_dma = !LDR || is_loader;
    const int dw = wave & 3;                             // index of this wave among the four that issue DMAs

    // ---- workgroup -> (sample, tile); with N % 8 == 0 sample n is pinned to XCD n % 8 (its d-slabs stay in that L2) ----
    const int tps = a.tiles_d * a.tiles_h * a.tiles_w;
    int n, tix;
    if (a.xcd_pin) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        n = xcd + 8 * (j / tps);
        tix = j % tps;
    } else {
        // other batch sizes: every XCD takes one contiguous run of the (sample, tile) raster
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        n = lin / tps;
        tix = lin % tps;
    }
    const int w0 = (tix % a.tiles_w) * TW;
    const int h0 = ((tix / a.tiles_w) % a.tiles_h) * TH;
    const int d0 = (tix / (a.tiles_w * a.tiles_h)) * TD;
    constexpr int P = KS / 2;

    const T* __restrict__ x = (const T*)a.x + (size_t)n * a.D * a.H * a.W * XLD;
    const T* __restrict__ w = (const T*)a.w;

    // ---- halo DMA: vector q = hv*NVV + pv, wave-instruction i covers q in [64 i, 64 i + 64) ----
    constexpr int NI_H = C::HALO_BYTES / 1024;
    auto issue_halo = [&]() {
#ifndef LT_ABL_NO_A   // -DLT_ABL_*: timing ablations for profiling builds (results are WRONG with any of them)
        for (int i = dw; i < NI_H; i += 4) {
            const int q = i * 64 + lane;
            const int hv = q / NVV, pv = q % NVV;
            const int hw_ = hv % C::PW, hh_ = (hv / C::PW) % C::HH, hd_ = hv / (C::PW * C::HH);
            const int lv = pv ^ C::fswz(hd_, hh_, hw_);
            const int id = d0 - P + hd_, ih = h0 - P + hh_, iw = w0 - P + hw_;
            const bool ok = hv < C::HV && hw_ < C::HW && ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            const void* src = ok ? (const void*)(x + (((size_t)id * a.H + ih) * a.W + iw) * XLD + cph * CIN + lv * C::VEC) : zero_page;
            dma16h(src, lds0 + i * 1024);
        }
#endif
    };
    if (do_dma) issue_halo();
    // ---- weight chunk DMA: vector q = (tap_in_chunk*CP + col)*NVV + pv ----
    constexpr int NI_W = C::WCH / 1024;
    auto stage_w = [&](int ch, int buf) {
#ifdef LT_ABL_NO_B
        return;
#endif
        for (int i = dw; i < NI_W; i += 4) {
            const int q = i * 64 + lane;
            const int pv = q % NVV, col = (q / NVV) % CP, tj = q / (NVV * CP);
            const int tap = ch * TPC + tj;
            const int lv = pv ^ ((-(col / VPR)) & (NVV - 1));
            const bool ok = tj < TPC && tap < C::NTAPS;
            const void* src = ok ? (const void*)(w + (size_t)col * a.k_pad + tap * XLD + cph * CIN + lv * C::VEC) : zero_page;
            dma16h(src, lds0 + C::HALO_BYTES + buf * C::WCH + i * 1024);
        }
    };
#ifdef LT_ABL_NO_B
    const int dpc = 0;
#else
    const int dpc = (NI_W - dw + 3) / 4;          // weight DMA instructions per chunk issued by this wave (wave-uniform)
#endif
    if (do_dma) {
#pragma unroll
        for (int c = 0; c < NBUF - 1; ++c)
            if (c < C::NCH) stage_w(c, c);
    }
    if (is_loader) {
        // ================================= loader waves: the chunk ring, then out =================================
        for (int ch = 0; ch < C::NCH; ++ch) {
            int younger = C::NCH - 1 - ch;
            if (younger > NBUF - 2) younger = NBUF - 2;
            wait_vmcnt_h(younger * dpc);
            asm volatile("s_barrier" ::: "memory");
            if (ch + NBUF - 1 < C::NCH) stage_w(ch + NBUF - 1, (ch + NBUF - 1) % NBUF);
        }
        asm volatile("s_barrier" ::: "memory");          // the compute waves' "done with the LDS images" barrier
        return;
    }

    HaloCst<SN> cst;
    cst.load(a, lane, MF);
    // ---- residual prefetch: the lane's 16-byte residual vectors (up to 8) are requested before the tap loop, in named
    // registers (see conv_igemm2.hip for why not an array); they are consumed in the epilogue ----
    constexpr int E_VECO = C::OVEC, E_LPR = CP / E_VECO, E_RPP = 64 / E_LPR, E_NIT = 64 / E_RPP;
    static_assert(E_NIT <= 16, "epilogue rows per lane");
    // (the 64-channel loader-wave configuration is at its 256-VGPR budget: it loads the residual in the epilogue instead)
    constexpr bool PRE_OK = E_NIT <= 8 && !(LDR && CINB == 128);
    const bool vec_epi = (a.Cout % E_VECO == 0) && (a.ldc % E_VECO == 0);
    const bool pre_res = PRE_OK && vec_epi && a.res != nullptr && !(a.flags & LT_EPI_NO_RES_PREFETCH);
    uint4 rp0, rp1, rp2, rp3, rp4, rp5, rp6, rp7;
    rp0 = rp1 = rp2 = rp3 = rp4 = rp5 = rp6 = rp7 = make_uint4(0, 0, 0, 0);
    if (pre_res) {
        const int cqp = (lane % E_LPR) * E_VECO;
        auto pf = [&](int it) -> uint4 {
            const int r = 64 * wave + lane / E_LPR + it * E_RPP;
            const int tw = r % TW, th = (r / TW) % TH, td = r / (TW * TH);
            const size_t pixv = (((size_t)n * a.D + d0 + td) * a.H + h0 + th) * a.W + w0 + tw;
            const void* src = cqp < a.Cout ? (const void*)((const typename C::OT*)a.res + pixv * a.ldc + cqp) : zero_page;
            return *(const uint4*)src;
        };
        if (E_NIT > 0) rp0 = pf(0);
        if (E_NIT > 1) rp1 = pf(1);
        if (E_NIT > 2) rp2 = pf(2);
        if (E_NIT > 3) rp3 = pf(3);
        if (E_NIT > 4) rp4 = pf(4);
        if (E_NIT > 5) rp5 = pf(5);
        if (E_NIT > 6) rp6 = pf(6);
        if (E_NIT > 7) rp7 = pf(7);
    }

    // ---- per-lane fragment addresses, hoisted out of the tap loop ----
    // PMC on the first version: 8.4 VALU + 6 SALU per MFMA (the 7^3 kernel was issue-bound on address arithmetic, not on
    // LDS or MFMA).  Now: A address = abase[kwv][g][i] (per lane: own voxel + swizzle term, which depends on the tap only
    // through kw -- or (kh,kw) for the 128-byte-voxel swizzle) + per-chunk scalar (kd / (kd,kh) plane or row offset)
    // + compile-time immediate (the rest of the tap offset).  The tap loop itself is ds_reads and MFMAs only.
    constexpr bool ROWCH = TPC == KS;                       // a chunk is one (kd,kh) row of taps; else one kd plane (TPC == KS*KS)
    static_assert(TPC == KS || TPC == KS * KS, "chunk = one tap row or one tap plane");
    constexpr bool KW_ONLY = C::SW::FA == 0 && C::SW::FC == 0;   // swizzle term independent of kh
    static_assert(C::SW::FB == 0, "swizzle must not depend on kd");
    static_assert(KW_ONLY || KS == 3, "kh-dependent swizzle only instantiated for 3^3");
    constexpr int NXV = KW_ONLY ? KS : KS * KS;             // swizzle variants
    const int lvb = (MF == 32) ? (lane >> 5) : (lane >> 4);   // logical vector of group 0; group g adds (MF==32 ? 2g : 4g)
    int abase[NXV][G][SM];                                  // bytes: own voxel * CINB + ((lv_g ^ f) << 4)
#pragma unroll
    for (int i = 0; i < SM; ++i) {
        const int r = 64 * wave + i * MF + (lane & (MF - 1));
        const int tw = r % TW, th = (r / TW) % TH, td = r / (TW * TH);
        const int own = ((td * C::HH + th) * C::PW + tw) * CINB;
#pragma unroll
        for (int v = 0; v < NXV; ++v) {
            const int kw = KW_ONLY ? v : v % KS, kh = KW_ONLY ? 0 : v / KS;
            const int f = C::fswz(0, th + kh, tw + kw);
#pragma unroll
            for (int g = 0; g < G; ++g) abase[v][g][i] = own + (((lvb + ((MF == 32) ? 2 * g : 4 * g)) ^ f) << 4);
        }
    }
    int bbase[G][SN];                                       // bytes inside a tap slab
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int col = j * MF + (lane & (MF - 1));
        const int bsw = (-(col / VPR)) & (NVV - 1);
#pragma unroll
        for (int g = 0; g < G; ++g) bbase[g][j] = col * CINB + (((lvb + ((MF == 32) ? 2 * g : 4 * g)) ^ bsw) << 4);
    }

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

    if constexpr (PD > 1) {
        // ---- fragment RING (7^3): one workgroup per CU (the halo takes 123 KB) = one wave per SIMD, so nothing but the wave's
        // own lookahead hides the LDS latency; with the one-tap lookahead below the kernel ran at ~290 cycles per tap against
        // 64 cycles of MFMA.  Here the fragments of tap t+PD are requested before the MFMAs of tap t, across chunk boundaries:
        // ring slot = tap index inside the chunk (TPC slots), the halo image is static, and the weight ring runs one chunk
        // further ahead (chunk ch+1 has landed at the barrier of chunk ch) so that its fragments may be read early.
        static_assert(ROWCH && KW_ONLY && PD < TPC && NBUF >= 4, "ring path: row chunks, kw-only swizzle, >= 4 weight buffers");
        V16 fa[TPC][G][SM], fb[TPC][G][SN];
        constexpr int RPT = G * (SM + SN);                  // ds_reads per tap
        static_assert((PD + 1) * RPT <= 15, "lookahead exceeds the lgkmcnt counter");
        const unsigned lds_w = lds0 + C::HALO_BYTES;
        // tap TJ of the chunk whose halo row offset is coff_x and whose weight buffer starts at LDS byte wb_x
        auto load_ring = [&](int coff_x, unsigned wb_x, auto tjc) {
            constexpr int TJ = decltype(tjc)::value;
            static_for<0, G>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                static_for<0, SM>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    lds_read16<TJ * CINB>(fa[TJ][g][i], lds0 + abase[TJ][g][i] + coff_x);
                });
                static_for<0, SN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    lds_read16<TJ * C::SLAB>(fb[TJ][g][j], wb_x + bbase[g][j]);
                });
            });
        };
        auto mma_tap = [&](auto tjc) {
            constexpr int TJ = decltype(tjc)::value;
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int i = 0; i < SM; ++i) frag_ready(fa[TJ][g][i]);
#pragma unroll
                for (int j = 0; j < SN; ++j) frag_ready(fb[TJ][g][j]);
            }
            mma_tap_blocks<T, MF, SM, SN, G>(acc, fa[TJ], fb[TJ]);
        };
        // LAST_ (compile time): no chunk follows.  Two copies of the chunk body instead of a uniform branch around the
        // cross-chunk loads, so that the number of reads in flight at every wait is a compile-time constant.
#define LT_HALO_RING_CHUNK(LAST_)                                                                                       \
    {                                                                                                                   \
        /* chunks <= ch+1 must have landed; at most NBUF-3 younger chunks stay in flight */                             \
        int younger = C::NCH - 2 - ch;                                                                                  \
        if (younger > NBUF - 3) younger = NBUF - 3;                                                                     \
        if (younger < 0) younger = 0;                                                                                   \
        wait_vmcnt_h(younger * dpc);                                                                                    \
        asm volatile("s_barrier" ::: "memory"); /* all waves: chunks <= ch+1 landed, chunk ch-1 fully consumed */       \
        if (ch + NBUF - 1 < C::NCH) stage_w(ch + NBUF - 1, (ch + NBUF - 1) % NBUF);                                     \
        const int coff = (((ch / KS) * C::HH + (ch % KS)) * C::PW) * CINB; /* bytes, wave-uniform */                    \
        const int coff_n = ((((ch + 1) / KS) * C::HH + ((ch + 1) % KS)) * C::PW) * CINB;                                \
        const unsigned wb = lds_w + (ch % NBUF) * C::WCH;                                                               \
        const unsigned wb_n = lds_w + ((ch + 1) % NBUF) * C::WCH;                                                       \
        static_for<0, TPC>([&](auto tjc) {                                                                              \
            constexpr int tj = decltype(tjc)::value;                                                                    \
            constexpr int ahead = (tj + PD < TPC) ? PD : (LAST_ ? TPC - 1 - tj : PD); /* taps in flight behind tj */    \
            if constexpr (tj + PD < TPC) load_ring(coff, wb, std::integral_constant<int, tj + PD>{});                   \
            else if constexpr (!LAST_) load_ring(coff_n, wb_n, std::integral_constant<int, (tj + PD) % TPC>{});         \
            lgkm_wait<ahead * RPT>();                                                                                   \
            mma_tap(tjc);                                                                                               \
            if (ACC64 && (((ch * TPC + tj) & AMASK) == AMASK || ch * TPC + tj + 1 == C::NTAPS)) {                                               \
                _Pragma("unroll") for (int i = 0; i < SM; ++i)                                                          \
                    _Pragma("unroll") for (int j = 0; j < SN; ++j)                                                      \
                        _Pragma("unroll") for (int e = 0; e < NACC; ++e) {                                              \
                            dacc[i][j][e] += (double)acc[i][j][e];                                                      \
                            acc[i][j][e] = 0.f;                                                                         \
                        }                                                                                               \
            }                                                                                                           \
        });                                                                                                             \
    }
        for (int phase = 0; phase < NPH; ++phase) {
        if (NPH > 1 && phase) {          // next channel phase: every wave is done with the halo / weight images, then the same start-up as above
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cph = phase;
            issue_halo();
#pragma unroll
            for (int c = 0; c < NBUF - 1; ++c)
                if (c < C::NCH) stage_w(c, c);
        }
        // prologue: the first PD taps of chunk 0 (needs the halo and chunk 0: same wait as the loop's first barrier)
        {
            int younger = C::NCH - 2;
            if (younger > NBUF - 3) younger = NBUF - 3;
            if (younger < 0) younger = 0;
            wait_vmcnt_h(younger * dpc);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            static_for<0, PD>([&](auto tjc) { load_ring(0, lds_w, tjc); });
        }
        int ch = 0;
        for (; ch < C::NCH - 1; ++ch) LT_HALO_RING_CHUNK(false)
        LT_HALO_RING_CHUNK(true)
        }
#undef LT_HALO_RING_CHUNK
    } else {
        for (int phase = 0; phase < NPH; ++phase) {
        if (NPH > 1 && phase) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cph = phase;
            issue_halo();
#pragma unroll
            for (int c = 0; c < NBUF - 1; ++c)
                if (c < C::NCH) stage_w(c, c);
        }
        for (int ch = 0; ch < C::NCH; ++ch) {
            // chunk ch (and, the first time, the halo issued before it) must have landed; up to NBUF-2 younger chunks stay in flight
            int younger = C::NCH - 1 - ch;
            if (younger > NBUF - 2) younger = NBUF - 2;
            if (!LDR) wait_vmcnt_h(younger * dpc);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all waves: chunk ch landed, chunk ch-1 fully consumed
            if (!LDR && ch + NBUF - 1 < C::NCH) stage_w(ch + NBUF - 1, (ch + NBUF - 1) % NBUF);
            // per-chunk scalar parts
            const int kd = ROWCH ? ch / KS : ch, kh_row = ROWCH ? ch % KS : 0;
            const int coff = ((kd * C::HH + kh_row) * C::PW) * CINB;           // bytes, wave-uniform
            const unsigned char* wb = s_w + (ch % NBUF) * C::WCH;
            // Fragments of tap tj+1 are requested before the MFMAs of tap tj are issued (register double buffer, order pinned
            // with sched_barrier): with one wave per SIMD nothing else hides the ~100-cycle ds_read latency.
            // kh-dependent swizzle with row chunks: pick this row's variants with a wave-uniform switch (keeps every register
            // index static: a select chain here was turned into a dynamically indexed array = scratch memory)
            int arow[(ROWCH && !KW_ONLY) ? KS : 1][G][SM];
            if (ROWCH && !KW_ONLY) {
                switch (kh_row) {
                    case 0:
    #pragma unroll
                        for (int k = 0; k < KS; ++k)
    #pragma unroll
                            for (int g = 0; g < G; ++g)
    #pragma unroll
                                for (int i = 0; i < SM; ++i) arow[k][g][i] = abase[(0 * KS + k) % NXV][g][i];
                        break;
                    case 1:
    #pragma unroll
                        for (int k = 0; k < KS; ++k)
    #pragma unroll
                            for (int g = 0; g < G; ++g)
    #pragma unroll
                                for (int i = 0; i < SM; ++i) arow[k][g][i] = abase[(1 * KS + k) % NXV][g][i];
                        break;
                    default:
    #pragma unroll
                        for (int k = 0; k < KS; ++k)
    #pragma unroll
                            for (int g = 0; g < G; ++g)
    #pragma unroll
                                for (int i = 0; i < SM; ++i) arow[k][g][i] = abase[(2 * KS + k) % NXV][g][i];
                        break;
                }
            }
            V16 fa[2][G][SM], fb[2][G][SN];
            auto load_tap = [&](int tj, int slot) {     // tj is a compile-time constant after unrolling
                const int kw = ROWCH ? tj : tj % KS, kh = ROWCH ? 0 : tj / KS;
                const int imm = (kh * C::PW + kw) * CINB;                       // compile-time immediate
    #pragma unroll
                for (int g = 0; g < G; ++g) {
    #pragma unroll
                    for (int i = 0; i < SM; ++i) {
                        int ab;
                        if (KW_ONLY) ab = abase[kw][g][i];
                        else if (!ROWCH) ab = abase[kh * KS + kw][g][i];
                        else ab = arow[kw][g][i];
                        fa[slot][g][i].u = *(const uint4*)(s_halo + (ab + coff) + imm);
                    }
    #pragma unroll
                    for (int j = 0; j < SN; ++j) fb[slot][g][j].u = *(const uint4*)(wb + bbase[g][j] + tj * C::SLAB);
                }
            };
            load_tap(0, 0);
    #pragma unroll
            for (int tj = 0; tj < TPC; ++tj) {
                const int tap = ch * TPC + tj;
                if (tap < C::NTAPS) {
                    if (tj + 1 < TPC && tap + 1 < C::NTAPS) load_tap(tj + 1, (tj + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_tap_blocks<T, MF, SM, SN, G>(acc, fa[tj & 1], fb[tj & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ACC64 && ((tap & AMASK) == AMASK || tap + 1 == C::NTAPS)) {
    #pragma unroll
                        for (int i = 0; i < SM; ++i)
    #pragma unroll
                            for (int j = 0; j < SN; ++j)
    #pragma unroll
                                for (int e = 0; e < NACC; ++e) {
                                    dacc[i][j][e] += (double)acc[i][j][e];
                                    acc[i][j][e] = 0.f;
                                }
                    }
                }
            }
        }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the halo / weight images

#ifdef LT_ABL_NO_EPI
    if (a.N < 0)
#endif
    halo_epilogue<T, CP, MF, SM, SN, NACC, TH, TW, ACC64>(smem, a, wave, lane, n, d0, h0, w0, acc, dacc, pre_res, rp0, rp1, rp2, rp3, rp4, rp5,
                                                         rp6, rp7, cst);
}

// ---- persistent 3^3 kernel: weights resident in LDS, halo double-buffered, loader waves -----------------------------------
// The one-tile-per-workgroup kernel above spends most of a tile's life NOT in MFMAs at the 32-channel levels: a tile is only
// 27 taps x 4 MFMAs per wave (~3.4k matrix cycles), but its halo DMA round trip, the 27 x 2 KB weight stream with a barrier
// (and an L2 round trip) per chunk, and the epilogue are all serial inside the workgroup (measured 17 us per tile and
// workgroup).  Here a workgroup stays on its CU and walks tiles:
//   * all 27 tap slabs (54 KB at 32->32 bf16) are DMA'd once;
//   * waves 4-7 are LOADERS: they do nothing but request the halo of tile i+1 into the other halo buffer while waves 0-3
//     (CONSUMERS) compute tile i.  Shader-clock accounting of the first version, where the four compute waves also issued the
//     DMAs: 4.4k cycles per tile in the halo issue (the wave sits in the issue stage while the load path queues the requests),
//     3.4k in the tap loop, 5.9k in the epilogue (most of it the wait for that same DMA) -- all serial;
//   * the tap loop has no barrier, and runs a fragment pipeline PDU (k-group) units deep with hand-counted lgkmcnt;
//   * the MFMAs compute the transposed product (weights first): a lane ends up with 16 channels of its own voxel and stores
//     them as 8-byte bf16 groups straight from the accumulators -- no LDS staging tile.
// One workgroup barrier per tile (halo i landed / everybody done with the tap loop of tile i-1).  Tiles are dealt so that
// workgroup b (XCD b % 8) keeps to the samples / raster run of that XCD, as above.
template <typename T, int CIN, int CP>
__global__ __launch_bounds__(512) void conv3d_halo_persist_kernel(const HaloArgs a, const int total_tiles) {
    constexpr int KS = 3, TD = 4, TH = 8, TW = 8;
    typedef HaloCfg<T, KS, CIN, CP, TD, TH, TW, 9, 2> C;
    static_assert(sizeof(T) == 2, "bf16 only: fp32 slabs do not fit beside two halo buffers");
    constexpr int MF = C::MF, SM = C::SM, SN = C::SN, G = C::G, NACC = C::NACC, NVV = C::NVV, VPR = C::VPR, CINB = C::CINB;
    constexpr int W_BYTES = ((C::NTAPS * C::SLAB + 1023) / 1024) * 1024;
    static_assert(C::SW::FA == 0 && C::SW::FC == 0 && C::SW::FB == 0, "kw-only swizzle");
    static_assert(C::EP_BYTES <= C::HALO_BYTES, "epilogue staging must fit a halo buffer");
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lds_halo = lds0 + W_BYTES;            // two halo buffers follow the weights

    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page_h;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool loader = wave >= 4;
    const int wl = wave & 3;                             // index inside the role
    constexpr int P = 1;
    const int tps = a.tiles_d * a.tiles_h * a.tiles_w;
    const T* __restrict__ w = (const T*)a.w;

    auto tile_of = [&](int v, int& n, int& d0, int& h0, int& w0) {   // v: virtual workgroup index (v % 8 = XCD of this workgroup)
        int tix;
        if (a.xcd_pin) {
            const int xcd = v & 7, j = v >> 3;
            n = xcd + 8 * (j / tps);
            tix = j % tps;
        } else {
            const int nb = total_tiles, q = nb >> 3, r = nb & 7, xcd = v & 7, j = v >> 3;
            const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
            n = lin / tps;
            tix = lin % tps;
        }
        w0 = (tix % a.tiles_w) * TW;
        h0 = ((tix / a.tiles_w) % a.tiles_h) * TH;
        d0 = (tix / (a.tiles_w * a.tiles_h)) * TD;
    };
    constexpr int NI_H = C::HALO_BYTES / 1024;
    auto issue_halo = [&](int n, int d0, int h0, int w0, int buf) {   // loaders only: piece wl, wl + 4, ...
        const T* __restrict__ x = (const T*)a.x + (size_t)n * a.D * a.H * a.W * CIN;
        for (int i = wl; i < NI_H; i += 4) {
            const int q = i * 64 + lane;
            const int hv = q / NVV, pv = q % NVV;
            const int hw_ = hv % C::PW, hh_ = (hv / C::PW) % C::HH, hd_ = hv / (C::PW * C::HH);
            const int lv = pv ^ C::fswz(hd_, hh_, hw_);
            const int id = d0 - P + hd_, ih = h0 - P + hh_, iw = w0 - P + hw_;
            const bool ok = hv < C::HV && hw_ < C::HW && ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            const void* src = ok ? (const void*)(x + (((size_t)id * a.H + ih) * a.W + iw) * CIN + lv * C::VEC) : zero_page;
            dma16h(src, lds_halo + buf * C::HALO_BYTES + i * 1024);
        }
    };

    // ---- weights: every tap slab, once (all eight waves) ----
    constexpr int NI_W = W_BYTES / 1024;
    for (int i = wave; i < NI_W; i += 8) {
        const int q = i * 64 + lane;
        const int pv = q % NVV, col = (q / NVV) % CP, tap = q / (NVV * CP);
        const int lv = pv ^ ((-(col / VPR)) & (NVV - 1));
        const void* src = tap < C::NTAPS ? (const void*)(w + (size_t)col * a.k_pad + tap * CIN + lv * C::VEC) : zero_page;
        dma16h(src, lds0 + i * 1024);
    }

    int v = blockIdx.x;
    int n, d0, h0, w0;
    tile_of(v, n, d0, h0, w0);

#ifdef LT_TRACE
    long long tr[7] = {0, 0, 0, 0, 0, 0, 0};
    const long long tr_begin = LT_CLKH();
#define LT_TRH(k_) { const long long c_ = LT_CLKH(); tr[k_] += c_ - tr_last; tr_last = c_; }
    long long tr_last = tr_begin;
#else
#define LT_TRH(k_)
#endif

    if (loader) {
        // ================================= loader waves =================================
        issue_halo(n, d0, h0, w0, 0);
        for (int it = 0;; ++it) {
            const int vn = v + gridDim.x;
            const bool have_next = vn < total_tiles;
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // A: halo(it) (and the weights) landed
#ifndef LT_ABL_NO_A
            if (have_next) {
                int nn, nd0, nh0, nw0;
                tile_of(vn, nn, nd0, nh0, nw0);
                issue_halo(nn, nd0, nh0, nw0, (it & 1) ^ 1);               // lands while the consumers compute tile it
            }
#endif
            if (!have_next) break;
            v = vn;
        }
        return;
    }

    // ================================= consumer waves =================================
    // ---- per-lane fragment bases (halo part is rebased per tile: the buffer alternates) ----
    const int lvb = (MF == 32) ? (lane >> 5) : (lane >> 4);
    int abase[KS][G][SM];
#pragma unroll
    for (int i = 0; i < SM; ++i) {
        const int r = 64 * wave + i * MF + (lane & (MF - 1));
        const int tw = r % TW, th = (r / TW) % TH, td = r / (TW * TH);
        const int own = ((td * C::HH + th) * C::PW + tw) * CINB;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int f = C::fswz(0, th, tw + kw);
#pragma unroll
            for (int g = 0; g < G; ++g) abase[kw][g][i] = own + (((lvb + ((MF == 32) ? 2 * g : 4 * g)) ^ f) << 4);
        }
    }
    unsigned bbase[G][SN];
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int col = j * MF + (lane & (MF - 1));
        const int bsw = (-(col / VPR)) & (NVV - 1);
#pragma unroll
        for (int g = 0; g < G; ++g) bbase[g][j] = lds0 + col * CINB + (((lvb + ((MF == 32) ? 2 * g : 4 * g)) ^ bsw) << 4);
    }

    // fragment pipeline: unit u = tap * G + g (SM voxel-fragment reads + SN weight-fragment reads, SM*SN MFMAs); PDU units of lookahead
    constexpr int RPU = SM + SN;
    constexpr int PDU = 15 / RPU - 1 >= 4 ? 4 : 15 / RPU - 1;
    constexpr int RING = PDU + 1;
    constexpr int NU = C::NTAPS * G;
    static_assert(PDU >= 1 && (PDU + 1) * RPU <= 15, "lookahead exceeds the lgkmcnt counter");

    // ---- epilogue without LDS: the MFMAs compute the TRANSPOSED product D[co][voxel] (weights as the first operand), so lane
    // (voxel v = lane & 31 of fragment i, half h = lane >> 5) ends up with the channels c(e) = (e & 3) + 8 (e >> 2) + 4 h of ITS
    // voxel: four runs of four consecutive channels = four 8-byte bf16 stores (and four 8-byte residual loads) per fragment.
    // No staging tile, no second workgroup barrier per tile.
    static_assert(MF == 32 && SN == 1 && NACC == 16 && CP == 32, "transposed epilogue: one 32-channel column block");
    const int vl = lane & 31, hh = lane >> 5;
    float ebi[16], esc[16], esf[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = (e & 3) + 8 * (e >> 2) + 4 * hh;   // < 32 = cout_pad: the constant arrays are padded
        ebi[e] = a.bias ? a.bias[c] : 0.f;
        esc[e] = a.scale ? a.scale[c] : 1.f;
        esf[e] = a.shift ? a.shift[c] : 0.f;
    }
    const EpiFloors fl = epi_floors(a.flags);
    const bool has_res = a.res != nullptr;
    const unsigned no_res = has_res ? 0u : 0x80008000u;  // zeros -> -0.0 pairs: v + -0.0 == v
    // element offset of the lane's voxel of fragment i inside the tile, relative to the tile origin (td = wave, th = 4 i + v/8, tw = v%8)
    const size_t ldc = (size_t)a.ldc;
    const size_t voff0 = (((size_t)wave * a.H + (vl >> 3)) * a.W + (vl & 7)) * ldc + 4 * hh;
    const size_t vstep = (size_t)4 * a.W * ldc;          // fragment i -> th + 4

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the weight slabs (and the constants)
    for (int it = 0;; ++it) {
        const int buf = it & 1;
        const int vn = v + gridDim.x;
        const bool have_next = vn < total_tiles;
        // A: halo(it) has landed (the loaders waited for it); every consumer is done with the tap loop of tile it-1 (its stores may still fly)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LT_TRH(1)
        int nn = 0, nd0 = 0, nh0 = 0, nw0 = 0;
        if (have_next) tile_of(vn, nn, nd0, nh0, nw0);
        LT_TRH(2)
        const size_t tbase = ((((size_t)n * a.D + d0) * a.H + h0) * a.W + w0) * ldc + voff0;
        // residual of this tile: 8-byte pieces in named registers (an array stayed in scratch memory), consumed after the tap loop
        uint2 rq00, rq01, rq02, rq03, rq10, rq11, rq12, rq13;
        rq00 = rq01 = rq02 = rq03 = rq10 = rq11 = rq12 = rq13 = make_uint2(0, 0);
        if (has_res) {
            const T* rb = (const T*)a.res + tbase;
            rq00 = *(const uint2*)(rb + 0); rq01 = *(const uint2*)(rb + 8); rq02 = *(const uint2*)(rb + 16); rq03 = *(const uint2*)(rb + 24);
            rb += vstep;
            rq10 = *(const uint2*)(rb + 0); rq11 = *(const uint2*)(rb + 8); rq12 = *(const uint2*)(rb + 16); rq13 = *(const uint2*)(rb + 24);
        }
        LT_TRH(3)
        acc_t acc[SM];
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int e = 0; e < NACC; ++e) acc[i][e] = 0.f;

        unsigned ha[KS][G][SM];
        const unsigned hbase = lds_halo + buf * C::HALO_BYTES;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int i = 0; i < SM; ++i) ha[kw][g][i] = hbase + abase[kw][g][i];

        V16 fa[RING][SM], fb[RING][SN];
        auto load_unit = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int tap = u / G, g = u % G, slot = u % RING;
            constexpr int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            constexpr int imm = ((kd * C::HH + kh) * C::PW + kw) * CINB;
            static_for<0, SM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                lds_read16<imm>(fa[slot][i], ha[kw][g][i]);
            });
            static_for<0, SN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                lds_read16<tap * C::SLAB>(fb[slot][j], bbase[g][j]);
            });
        };
#ifndef LT_ABL_NO_MMA
        static_for<0, PDU>([&](auto uc) { load_unit(uc); });
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int slot = u % RING;
            constexpr int ahead = (u + PDU < NU) ? PDU : NU - 1 - u;     // units in flight behind u at the wait
            if constexpr (u + PDU < NU) load_unit(std::integral_constant<int, u + PDU>{});
            lgkm_wait<ahead * RPU>();
#pragma unroll
            for (int i = 0; i < SM; ++i) frag_ready(fa[slot][i]);
            frag_ready(fb[slot][0]);
#pragma unroll
            for (int i = 0; i < SM; ++i) Mma<T, MF>::run(acc[i], fb[slot][0], fa[slot][i]);   // D[co][voxel]: weights first
        });
#endif
        LT_TRH(4)
        LT_TRH(5)
        // ---- epilogue straight from the accumulators ----
#ifdef LT_ABL_NO_EPI
        if (a.N < 0)
#endif
        {
            T* yb = (T*)a.y + tbase;
#define LT_HALO_T_ROW(I_, G_, RQ_)                                                                                  \
            {                                                                                                       \
                const unsigned r0 = RQ_.x | no_res, r1 = RQ_.y | no_res;                                            \
                float v0 = (acc[I_][4 * G_ + 0] + ebi[4 * G_ + 0]) * esc[4 * G_ + 0] + esf[4 * G_ + 0];             \
                float v1 = (acc[I_][4 * G_ + 1] + ebi[4 * G_ + 1]) * esc[4 * G_ + 1] + esf[4 * G_ + 1];             \
                float v2 = (acc[I_][4 * G_ + 2] + ebi[4 * G_ + 2]) * esc[4 * G_ + 2] + esf[4 * G_ + 2];             \
                float v3 = (acc[I_][4 * G_ + 3] + ebi[4 * G_ + 3]) * esc[4 * G_ + 3] + esf[4 * G_ + 3];             \
                v0 = epi_apply(v0, fl, __uint_as_float(r0 << 16));                                                  \
                v1 = epi_apply(v1, fl, __uint_as_float(r0 & 0xffff0000u));                                          \
                v2 = epi_apply(v2, fl, __uint_as_float(r1 << 16));                                                  \
                v3 = epi_apply(v3, fl, __uint_as_float(r1 & 0xffff0000u));                                          \
                *(uint2*)(yb + I_ * vstep + 8 * G_) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));         \
            }
            LT_HALO_T_ROW(0, 0, rq00) LT_HALO_T_ROW(0, 1, rq01) LT_HALO_T_ROW(0, 2, rq02) LT_HALO_T_ROW(0, 3, rq03)
            LT_HALO_T_ROW(1, 0, rq10) LT_HALO_T_ROW(1, 1, rq11) LT_HALO_T_ROW(1, 2, rq12) LT_HALO_T_ROW(1, 3, rq13)
#undef LT_HALO_T_ROW
        }
        LT_TRH(6)
#ifdef LT_TRACE
        tr[0] += 1;
#endif
        if (!have_next) break;
        v = vn; n = nn; d0 = nd0; h0 = nh0; w0 = nw0;
    }
#ifdef LT_TRACE
    if (wave == 0 && lane == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64) {
        long long* o = g_trace_h + (blockIdx.x >> 3) * 8;
        o[0] = LT_CLKH() - tr_begin;
        for (int k = 1; k < 7; ++k) o[k] = tr[k];
        o[7] = tr[0];
    }
#endif
#undef LT_TRH
}

// ---- column-walking 3^3 kernel: sliding window of halo planes, epilogue of tile i under the MFMAs of tile i+1 ------------------
// Shader-clock accounting of the persistent kernel above (32 samples, 64^3): 6.8k cycles per tile = 0.7k waiting for the halo
// + 0.5k tile arithmetic + 0.5k residual issue + 3.6k tap loop (108 MFMAs = 3.5k) + 1.5k epilogue, all serial in the consumer
// waves; and the halo of tile i+1, requested right after the barrier of tile i, lands ~5.8k cycles later: one 38 KB halo in
// flight per CU is what the memory system is given (Little's law: 256 x 38 KB / 2.7 us = 3.6 TB/s, the measured rate).  Here:
//   * a workgroup walks COLUMNS of tiles along d: consecutive tiles share two of their six halo planes, so the halo becomes a
//     stream of 4-plane groups (25 KB per tile instead of 38: -33 % L2->LDS traffic) in a ring of four groups -- two being
//     read, two in flight or landed (51 KB requested ahead per CU);
//   * the epilogue of tile i-1 (affine, ReLU floors, residual, 8-byte stores) is issued in eight pieces between the MFMA
//     units of tile i, out of a copy of the accumulators; the residual of tile i is requested in the middle of its own tap
//     loop, after the pieces have consumed the previous one; tile coordinates advance by a stride (no divisions per tile).
// F32OUT: LT_EPI_STORE_F32 (the mixed-precision training step) as its own instantiation -- the inference kernel keeps its registers and schedule.
// SKIP: the residual is not read but COMPUTED -- the 1x1x1 skip convolution of a Res3DBlock whose channel count changes (v2v.py:33-42, 16 -> 32 at the
// 64^3 level): one more MFMA per fragment on the 16 input channels of the lane's own voxel (a 16-byte load instead of the 32-byte residual), its
// BatchNorm scale folded into the bf16 weights and its shift into `shift` by the caller.  The skip convolution's launch, its 32-channel output and that
// tensor's read here all disappear (lt_conv_skip_fwd).
template <typename T, bool F32OUT, bool SKIP>
__global__ __launch_bounds__(512) void conv3d_halo_col_kernel(const HaloArgs a, const int total_cols) {
    constexpr int KS = 3, CIN = 32, CP = 32, TD = 4, TH = 8, TW = 8;
    typedef HaloCfg<T, KS, CIN, CP, TD, TH, TW, 9, 2> C;
    static_assert(sizeof(T) == 2, "bf16 only");
    constexpr int MF = C::MF, SM = C::SM, SN = C::SN, G = C::G, NACC = C::NACC, NVV = C::NVV, VPR = C::VPR, CINB = C::CINB;
    static_assert(MF == 32 && SM == 2 && SN == 1 && G == 2 && NACC == 16 && NVV == 4, "3^3 32->32 bf16 layout");
    static_assert(C::SW::FA == 0 && C::SW::FC == 0 && C::SW::FB == 0, "the swizzle must not depend on the plane");
    constexpr int W_BYTES = ((C::NTAPS * C::SLAB + 1023) / 1024) * 1024;
    constexpr int PLANE_V = C::HH * C::PW;               // voxel slots of one halo plane
    constexpr int PLANE_B = PLANE_V * CINB;
    constexpr int GROUP_B = TD * PLANE_B;                // one tile step = TD new planes
    constexpr int NI_G = GROUP_B / 1024;
    static_assert(GROUP_B % 1024 == 0, "a plane group must be whole DMA wave-instructions");
    static_assert(W_BYTES + 4 * GROUP_B <= 160 * 1024, "weights + four plane groups must fit LDS");
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lds_halo = lds0 + W_BYTES;            // ring of four plane groups

    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page_h;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool loader = wave >= 4;
    const int wl = wave & 3;
    const T* __restrict__ w = (const T*)a.w;
    const int tpc = a.tiles_d, ngc = a.tiles_d + 1;      // tiles / plane groups per column
    const int cps = a.tiles_h * a.tiles_w;               // columns per sample
    const int ncol = (total_cols - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    auto col_of = [&](int v, int& n, int& h0, int& w0) {  // v % 8 = XCD of this workgroup (gridDim.x % 8 == 0)
        const int xcd = v & 7, j = v >> 3;
        int cix;
        if (a.xcd_pin) {
            n = xcd + 8 * (j / cps);
            cix = j % cps;
        } else {
            const int lin = xcd * (total_cols >> 3) + j;     // total_cols % 8 == 0
            n = lin / cps;
            cix = lin % cps;
        }
        w0 = (cix % a.tiles_w) * TW;
        h0 = (cix / a.tiles_w) * TH;
    };

    // ---- weights: every tap slab, once (all eight waves) ----
    constexpr int NI_W = W_BYTES / 1024;
    for (int i = wave; i < NI_W; i += 8) {
        const int q = i * 64 + lane;
        const int pv = q % NVV, col = (q / NVV) % CP, tap = q / (NVV * CP);
        const int lv = pv ^ ((-(col / VPR)) & (NVV - 1));
        const void* src = tap < C::NTAPS ? (const void*)(w + (size_t)col * a.k_pad + tap * CIN + lv * C::VEC) : zero_page;
        dma16h(src, lds0 + i * 1024);
    }

    if (loader) {
        // ================================= loader waves =================================
        // stream of plane groups: column c of this workgroup contributes groups 0..tpc (group g = planes 4g-1 .. 4g+2 of the
        // column), stream index S = c * ngc + g lives in ring slot S & 3; tile (c, k) reads S and S + 1.
        // The (plane, row, column, k-vector) a lane fetches for DMA piece i is the same for every group: decode it ONCE (the
        // divisions below cost ~60 VALU instructions per piece, and the loaders share their SIMDs' issue slots with the
        // consumers' MFMA stream); per group only the plane validity and one base pointer change.
        const int total_groups = ncol * ngc;
        constexpr int MAXP = (NI_G + 3) / 4;              // pieces per wave and group
        int poff[MAXP], pmeta[MAXP];                      // element offset from (plane 4g-1, row h0, column w0); pj << 16 | hh << 8 | hw
#pragma unroll
        for (int m = 0; m < MAXP; ++m) {
            const int q = (wl + 4 * m) * 64 + lane;
            const int pj = q / (PLANE_V * NVV), r = q - pj * (PLANE_V * NVV);
            const int hv = r / NVV, pv = r % NVV;
            const int hh_ = hv / C::PW, hw_ = hv - hh_ * C::PW;
            const int lv = pv ^ C::fswz(0, hh_, hw_);
            poff[m] = ((pj * a.H + hh_ - 1) * a.W + hw_ - 1) * CIN + lv * C::VEC;
            pmeta[m] = (pj << 16) | (hh_ << 8) | hw_;
        }
        int issued = 0, icol = 0, ig = 0, in, ih0, iw0;
        unsigned okhw = 0;                                // bit m: piece m's (row, column) lies inside the sample for this column
        auto enter_col = [&](int v) {
            col_of(v, in, ih0, iw0);
            okhw = 0;
#pragma unroll
            for (int m = 0; m < MAXP; ++m) {
                const int hh_ = (pmeta[m] >> 8) & 255, hw_ = pmeta[m] & 255;
                const bool ok = hw_ < C::HW && ((unsigned)(ih0 - 1 + hh_) < (unsigned)a.H) & ((unsigned)(iw0 - 1 + hw_) < (unsigned)a.W);
                okhw |= ok ? (1u << m) : 0u;
            }
        };
        enter_col(blockIdx.x);
        auto issue_next = [&]() {
            const int d_first = 4 * ig - 1;
            // plane d_first, row h0, column w0 of sample n (outside the tensor for the padding planes: only formed, never read)
            const T* __restrict__ gb = (const T*)a.x + (((ptrdiff_t)in * a.D + d_first) * a.H + ih0) * (ptrdiff_t)a.W * CIN + (ptrdiff_t)iw0 * CIN;
            const unsigned dst = lds_halo + (issued & 3) * GROUP_B;
#pragma unroll
            for (int m = 0; m < MAXP; ++m) {
                if (wl + 4 * m < NI_G) {
                    const bool ok = ((okhw >> m) & 1u) && (unsigned)(d_first + (pmeta[m] >> 16)) < (unsigned)a.D;
                    const void* src = ok ? (const void*)(gb + poff[m]) : zero_page;
#ifdef LT_ABL_NO_A
                    if (a.N < 0)
#endif
                    dma16h(src, dst + (wl + 4 * m) * 1024);
                }
            }
            ++issued;
            if (++ig == ngc) {
                ig = 0;
                if (++icol < ncol) enter_col(blockIdx.x + icol * gridDim.x);
            }
        };
        const int mine = (NI_G - wl + 3) / 4;             // DMA pieces of one group requested by this wave
        for (int k = 0; k < 4 && issued < total_groups; ++k) issue_next();
        int sidx = 0, k = 0;
        const int ntile = ncol * tpc;
        for (int ti = 0; ti < ntile; ++ti) {
            const int ahead = issued - sidx - 2;          // groups requested beyond the two this tile reads: 0, 1 or 2
            wait_vmcnt_h(ahead > 0 ? ahead * mine : 0);
            asm volatile("s_barrier" ::: "memory");       // A: groups sidx, sidx + 1 landed; the consumers are done with tile ti - 1
            while (issued <= sidx + 3 && issued < total_groups) issue_next();
            ++sidx;
            if (++k == tpc) { k = 0; ++sidx; }
        }
        return;
    }

    // ================================= consumer waves =================================
    // wave = output plane td of the tile; fragment i covers rows th = 4 i + v / 8, tw = v % 8 of that plane
    const int lvb = lane >> 5;
    int abase[KS][G][SM];                                 // in-plane byte offset of (voxel, k-vector), per kw
#pragma unroll
    for (int i = 0; i < SM; ++i) {
        const int r = i * MF + (lane & (MF - 1));
        const int tw = r % TW, th = r / TW;
        const int own = (th * C::PW + tw) * CINB;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int f = C::fswz(0, th, tw + kw);
#pragma unroll
            for (int g = 0; g < G; ++g) abase[kw][g][i] = own + (((lvb + 2 * g) ^ f) << 4);
        }
    }
    // MFMA row r of the transposed product (= the weight row this lane feeds) carries output channel chan(r), chosen so that
    // the 16 result registers of lane (voxel, h) are channels 8h .. 8h+7 and 16+8h .. 16+8h+7: two 16-byte runs, loaded
    // (residual) and stored as such.  The MFMA's own row order (r = 8q + 4h + j in register 4q + j) would give four 8-byte runs
    // per lane: twice the vector-memory instructions, and their issue (~65 cycles each for 64 scattered lanes) is paid by the
    // wave that also issues the MFMAs.
    unsigned bbase[G][SN];
    {
        const int r = lane & 31;
        const int chan = 16 * (r >> 4) + 8 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1) + (r & 3);
        const int bsw = (-(chan / VPR)) & (NVV - 1);
#pragma unroll
        for (int g = 0; g < G; ++g) bbase[g][0] = lds0 + chan * CINB + (((lvb + 2 * g) ^ bsw) << 4);
    }
    constexpr int RPU = SM + SN;
    constexpr int PDU = 4;
    constexpr int RING = PDU + 1;
    constexpr int NU = C::NTAPS * G;
    static_assert((PDU + 1) * RPU <= 15, "lookahead exceeds the lgkmcnt counter");

    // transposed product D[co][voxel]: lane (voxel v = lane & 31 of fragment i, half h = lane >> 5) holds channels
    // c(e) = 8 h + e (e < 8) and 16 + 8 h + (e - 8) of its voxel; (acc + bias) * scale + shift folded into one fma per value
    const int vl = lane & 31, hh = lane >> 5;
    float esc[16], esf[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = 16 * (e >> 3) + 8 * hh + (e & 7);
        const float bi = a.bias ? a.bias[c] : 0.f, sc = a.scale ? a.scale[c] : 1.f, sf = a.shift ? a.shift[c] : 0.f;
        esc[e] = sc;
        esf[e] = bi * sc + sf;
    }
    const EpiFloors fl = epi_floors(a.flags);
    const bool has_res = a.res != nullptr;
    const unsigned no_res = has_res ? 0u : 0x80008000u;  // zeros -> -0.0 pairs: v + -0.0 == v
    const size_t ldc = (size_t)a.ldc;
    const size_t voff0 = (((size_t)wave * a.H + (vl >> 3)) * a.W + (vl & 7)) * ldc + 8 * hh;
    const size_t vstep = (size_t)4 * a.W * ldc;          // fragment i -> th + 4
    const size_t dstep = (size_t)TD * a.H * a.W * ldc;   // next tile of the column
    const size_t rstep = has_res ? vstep : 0;            // without a residual every piece reads the zero page
    constexpr int SKC = 16;                              // channels per voxel of the skip tensor
    const size_t svoff0 = (((size_t)wave * a.H + (vl >> 3)) * a.W + (vl & 7)) * SKC + 8 * hh;
    const size_t svstep = (size_t)4 * a.W * SKC, sdstep = (size_t)TD * a.H * a.W * SKC;
    V16 wsk;
    if constexpr (SKIP) wsk.u = *(const uint4*)((const T*)a.skip_w + (size_t)lane * 8);

    int n, h0, w0;
    col_of(blockIdx.x, n, h0, w0);
    size_t tbase = (((size_t)n * a.D * a.H + h0) * a.W + w0) * ldc + voff0;
    size_t sbase = (((size_t)n * a.D * a.H + h0) * a.W + w0) * SKC + svoff0;
    size_t pbase = 0;                                    // output offset of the tile whose epilogue is pending
    acc_t sacc;                                          // SKIP: W_skip . x_skip of the fragment whose pieces are being issued

    acc_t acc[SM], pacc[SM];
    uint4 rq[4], rqn[4];                                  // residual of the pending tile / of the tile being computed
    unsigned ha[KS][G][SM];
    unsigned hd1 = 0, hd2 = 0;                            // plane base of kd = 1 minus kd = 0, kd = 2 minus kd = 1
    V16 fa[RING][SM], fb[RING][SN];

    auto load_unit = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int tap = u / G, g = u % G, slot = u % RING;
        constexpr int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        constexpr int imm = (kh * C::PW + kw) * CINB;
        if constexpr (u > 0 && u % (9 * G) == 0) {       // first unit of the next kd: move the twelve fragment bases one plane on
            const unsigned dlt = kd == 1 ? hd1 : hd2;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int gg = 0; gg < G; ++gg)
#pragma unroll
                    for (int ii = 0; ii < SM; ++ii) ha[kk][gg][ii] += dlt;
        }
        static_for<0, SM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            lds_read16<imm>(fa[slot][i], ha[kw][g][i]);
        });
        lds_read16<tap * C::SLAB>(fb[slot][0], bbase[g][0]);
    };
    auto skip_mma = [&](auto ic) {                        // SKIP: fragment I of the pending tile (its 16 skip channels are in rq[I])
        constexpr int I = decltype(ic)::value;
        if constexpr (SKIP) {
            V16 xs;
            xs.u = rq[I];
            acc_t z;
#pragma unroll
            for (int e = 0; e < NACC; ++e) z[e] = 0.f;
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wsk.h, xs.h, z, 0, 0, 0);
        }
    };
    auto epi_piece = [&](auto pc) {                       // piece p of the pending tile: fragment p / 2, 8-channel run p % 2
        constexpr int p = decltype(pc)::value, I = p >> 1, Q = p & 1;
        const unsigned rr[4] = {rq[SKIP ? 0 : p].x | no_res, rq[SKIP ? 0 : p].y | no_res, rq[SKIP ? 0 : p].z | no_res, rq[SKIP ? 0 : p].w | no_res};
        unsigned o[4];
        float vf[F32OUT ? 8 : 1];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int e = 8 * Q + 2 * d;
            float v0 = fmaf(pacc[I][e], esc[e], esf[e]);
            float v1 = fmaf(pacc[I][e + 1], esc[e + 1], esf[e + 1]);
            v0 = epi_apply(v0, fl, SKIP ? sacc[e] : __uint_as_float(rr[d] << 16));
            v1 = epi_apply(v1, fl, SKIP ? sacc[e + 1] : __uint_as_float(rr[d] & 0xffff0000u));
            if constexpr (F32OUT) { vf[2 * d] = v0; vf[2 * d + 1] = v1; }
            else o[d] = pack_bf16x2(v0, v1);
        }
#ifdef LT_ABL_NO_STORE
        if (a.N < 0)
#endif
        if constexpr (F32OUT) {   // the same eight channels as two float4
            float* yp = (float*)a.y + pbase + I * vstep + 16 * Q;
            *(float4*)yp = make_float4(vf[0], vf[1], vf[2], vf[3]);
            *(float4*)(yp + 4) = make_float4(vf[4], vf[5], vf[6], vf[7]);
        } else {
            *(uint4*)((T*)a.y + pbase + I * vstep + 16 * Q) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };
    auto tap_loop = [&](auto epi_c) {
        constexpr bool EPI = decltype(epi_c)::value;
        static_for<0, PDU>([&](auto uc) { load_unit(uc); });
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int slot = u % RING;
            constexpr int ahead = (u + PDU < NU) ? PDU : NU - 1 - u;
            if constexpr (u + PDU < NU) load_unit(std::integral_constant<int, u + PDU>{});
            lgkm_wait<ahead * RPU>();
#pragma unroll
            for (int i = 0; i < SM; ++i) frag_ready(fa[slot][i]);
            frag_ready(fb[slot][0]);
#pragma unroll
            for (int i = 0; i < SM; ++i) Mma<T, MF>::run(acc[i], fb[slot][0], fa[slot][i]);   // D[co][voxel]: weights first
#ifndef LT_ABL_NO_EPI
            if constexpr (EPI && SKIP && (u == 0 || u == 12)) skip_mma(std::integral_constant<int, u / 12>{});   // two units in front of the pieces that use it
            if constexpr (EPI && u >= 2 && u < 26 && (u - 2) % 6 == 0) epi_piece(std::integral_constant<int, (u - 2) / 6>{});
#endif
            if constexpr (u == 0) {
                // residual of THIS tile, requested a whole tap loop before the pieces under the next tile consume it (requested
                // at unit 30 it arrived late: the first piece stalled ~1.3k cycles per tile).  Explicitly GLOBAL loads: a
                // pointer selected between the residual and the zero page is a flat pointer to the compiler, and flat loads
                // also count in lgkmcnt, which the fragment pipeline counts by hand
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(1))) u32x4* gq_t;
                if constexpr (SKIP) {                            // the lane's own voxel of fragments 0 and 1: 8 of the 16 skip channels each (half hh)
                    const unsigned long long sb = (unsigned long long)(size_t)((const T*)a.skip_x + sbase);
                    static_for<0, 2>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        const u32x4 rv = *(gq_t)(sb + (p * svstep) * sizeof(T));
                        rqn[p] = make_uint4(rv[0], rv[1], rv[2], rv[3]);
                    });
                } else {
                const unsigned long long rb = has_res ? (unsigned long long)(size_t)((const T*)a.res + tbase) : zp_bits;
                static_for<0, 4>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    const u32x4 rv = *(gq_t)(rb + ((p >> 1) * rstep + 16 * (p & 1)) * sizeof(T));
                    rqn[p] = make_uint4(rv[0], rv[1], rv[2], rv[3]);
                });
                }
            }
        });
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the weight slabs (and the constants)
    {   // the pointers the tile loop uses, in SGPRs HERE: left to itself hipcc fetches a.y from the kernel arguments right in front of the loop and puts the
        // s_waitcnt lgkmcnt(0) for it in front of the first store INSIDE the loop -- where it drains the hand-counted fragment pipeline once per tile
        const void* yp_ = a.y;
        const void* sx_ = SKIP ? a.skip_x : a.res;
        asm volatile("" ::"s"(yp_), "s"(sx_));
    }
#ifdef LT_TRACE
    long long tr[7] = {0, 0, 0, 0, 0, 0, 0};
    const long long tr_begin = LT_CLKH();
    long long tr_last = tr_begin;
#define LT_TRC(k_) { const long long c_ = LT_CLKH(); tr[k_] += c_ - tr_last; tr_last = c_; }
#else
#define LT_TRC(k_)
#endif
    int sidx = 0, k = 0, icol = 0;
    const int ntile = ncol * tpc;
    for (int ti = 0; ti < ntile; ++ti) {
        // A: plane groups sidx, sidx + 1 have landed; every consumer is done with the tap loop of the previous tile
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LT_TRC(1)
        {
            const unsigned p0 = lds_halo + ((sidx + (wave >> 2)) & 3) * GROUP_B + (wave & 3) * PLANE_B;
            const unsigned p1 = lds_halo + ((sidx + ((wave + 1) >> 2)) & 3) * GROUP_B + ((wave + 1) & 3) * PLANE_B;
            const unsigned p2 = lds_halo + ((sidx + ((wave + 2) >> 2)) & 3) * GROUP_B + ((wave + 2) & 3) * PLANE_B;
            hd1 = p1 - p0; hd2 = p2 - p1;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < SM; ++i) ha[kw][g][i] = p0 + abase[kw][g][i];
        }
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int e = 0; e < NACC; ++e) acc[i][e] = 0.f;
        LT_TRC(2)
        if (ti == 0) tap_loop(std::false_type{});
        else tap_loop(std::true_type{});
        LT_TRC(4)
#pragma unroll
        for (int i = 0; i < SM; ++i) pacc[i] = acc[i];
#pragma unroll
        for (int p = 0; p < 4; ++p) rq[p] = rqn[p];
        pbase = tbase;
        ++sidx;
        tbase += dstep;
        sbase += sdstep;
        if (++k == tpc) {                                // next column
            k = 0; ++sidx;
            if (++icol < ncol) {
                col_of(blockIdx.x + icol * gridDim.x, n, h0, w0);
                tbase = (((size_t)n * a.D * a.H + h0) * a.W + w0) * ldc + voff0;
                sbase = (((size_t)n * a.D * a.H + h0) * a.W + w0) * SKC + svoff0;
            }
        }
        LT_TRC(6)
    }
    static_for<0, 4>([&](auto pc) {                      // the last tile's epilogue has nothing to hide under
        if constexpr ((decltype(pc)::value & 1) == 0) skip_mma(std::integral_constant<int, decltype(pc)::value / 2>{});
        epi_piece(pc);
    });
#ifdef LT_TRACE
    if (wave == 0 && lane == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64) {   // same record as the persistent kernel
        long long* o = g_trace_h + (blockIdx.x >> 3) * 8;
        o[0] = LT_CLKH() - tr_begin;
        for (int q = 1; q < 7; ++q) o[q] = tr[q];
        o[7] = ntile;
    }
#endif
#undef LT_TRC
}

// ---- 3^3 64 -> 64: halo in LDS, weights from global memory in fragment order, two workgroups per CU ----------------------------
// The loader-wave kernel spends 37k cycles on a 64 -> 64 tile whose MFMAs take 13.8k: with the 77 KB halo AND a 72 KB weight ring
// in LDS only one workgroup fits a CU, so the halo round trip, the nine weight chunks (a barrier and an L2 round trip each) and
// the epilogue are all exposed.  Here the weights never touch LDS: lt_conv_pack_weights_t32 stores them once in the fragment
// order of the transposed product ([tap][16-channel K block][32-row Cout block][lane] x 16 bytes), a wave reads the one fragment
// it needs per (tap, K block) with a coalesced 1 KB global load (L1/L2 resident: 221 KB shared by every workgroup), and LDS
// holds the halo alone -- 75 KB, TWO workgroups per CU, whose load / MFMA / store phases overlap each other.  Wave = (Cout half,
// pair of output planes): four voxel fragments x one weight fragment per unit = four MFMAs per global load.  Voxel fragments keep
// the 128-byte-voxel swizzle of the kernels above (slot ^ f(row, column)); the K block enters the address as XOR (g << 5), one
// v_xor per read instead of 72 precomputed address registers.  Transposed product with permuted weight rows: the epilogue moves
// 16-byte channel runs straight from the accumulators (see conv3d_halo_col_kernel).
// Instantiated for 64 -> 64 (two Cout blocks: wave = (Cout half, pair of output planes), four voxel fragments), 32 -> 64 (same
// roles, two K blocks) and 128 -> 128 (four Cout blocks: wave = Cout block, all eight voxel fragments; 256-byte voxels fill
// LDS with the halo alone, one workgroup per CU with up to 512 registers per wave).
// Round 5: also 16 -> 32 (the first 3^3 layer of V2V at 64^3: one Cout block, wave = output plane, two voxel fragments per weight fragment; a 23 KB halo and
// 110 registers let four workgroups share a CU -- the one-tile loader-wave kernel it replaces there was latency-bound at 0.25 MFMA-busy).
template <typename T, int CIN, int CP>
__global__ __launch_bounds__(256, (CIN == 128 ? 1 : CIN == 16 ? 4 : 2)) void conv3d_halo_wreg_kernel(const HaloArgs a) {
    constexpr int KS = 3, TD = 4, TH = 8, TW = 8, G = CIN / 16, NB = CP / 32;
    typedef HaloCfg<T, KS, CIN, CP, TD, TH, TW, 3, 3> C;
    static_assert(sizeof(T) == 2 && (NB == 1 || NB == 2 || NB == 4) && (C::CINB == 32 || C::CINB == 64 || C::CINB == 128 || C::CINB == 256),
                  "bf16; 16/32/64/128 -> 32/64/128");
    static_assert(C::SW::FB == 0, "plane-independent swizzle");
    constexpr int PLANE_B = C::HH * C::PW * C::CINB;
    constexpr int NI_H = C::HALO_BYTES / 1024;
    constexpr int FR = NB == 1 ? 2 : NB == 2 ? 4 : 8;     // voxel fragments per wave
    constexpr bool PRE_RES = FR <= 4;                     // residual requested before the tap loop (register budget)
    // swizzle of the 16-byte vectors of a voxel: the searched ones for 64- and 128-byte voxels; 256-byte voxels start at bank 0
    // each, so the 16 lanes of a read phase (2 rows x 8 columns) must use 16 different slots: column & 7 | (row & 1) << 3
    auto wswz = [](int hh_, int hw_) -> int { return C::CINB == 256 ? ((hw_ & 7) | ((hh_ & 1) << 3)) : C::fswz(0, hh_, hw_); };

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page_h;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tps = a.tiles_d * a.tiles_h * a.tiles_w;
    int n, tix;
    if (a.xcd_pin) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        n = xcd + 8 * (j / tps);
        tix = j % tps;
    } else {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        n = lin / tps;
        tix = lin % tps;
    }
    const int w0 = (tix % a.tiles_w) * TW;
    const int h0 = ((tix / a.tiles_w) % a.tiles_h) * TH;
    const int d0 = (tix / (a.tiles_w * a.tiles_h)) * TD;
    const T* __restrict__ x = (const T*)a.x + (size_t)n * a.D * a.H * a.W * CIN;

    // ---- halo DMA, all four waves ----
    for (int i = wave; i < NI_H; i += 4) {
        const int q = i * 64 + lane;
        const int hv = q / C::NVV, pv = q % C::NVV;
        const int hw_ = hv % C::PW, hh_ = (hv / C::PW) % C::HH, hd_ = hv / (C::PW * C::HH);
        const int lv = pv ^ wswz(hh_, hw_);
        const int id = d0 - 1 + hd_, ih = h0 - 1 + hh_, iw = w0 - 1 + hw_;
        const bool ok = hv < C::HV && ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        const void* src = ok ? (const void*)(x + (((size_t)id * a.H + ih) * a.W + iw) * CIN + lv * C::VEC) : zero_page;
        dma16h(src, lds0 + i * 1024);
    }

    // ---- roles ----
    const int cb = NB == 1 ? 0 : NB == 2 ? (wave & 1) : wave;             // Cout block of 32
    const int p0 = NB == 1 ? wave : NB == 2 ? 2 * (wave >> 1) : 0;        // first output plane of this wave's fragments
    const int vl = lane & 31, hh = lane >> 5;
    // voxel-fragment addresses: fragment f = (plane p0 + f / 2, rows 4 (f & 1) + vl / 8, column vl % 8); tap (kd, kh, kw), K block g:
    //   (lp[kh*3+kw][f & 1] ^ (g << 5)) + (f / 2 + kd) planes
    unsigned lp[9][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int th = 4 * i + (vl >> 3), tw = vl & 7;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                lp[kh * 3 + kw][i] = lds0 + ((p0 * C::HH + th + kh) * C::PW + tw + kw) * C::CINB + ((hh ^ wswz(th + kh, tw + kw)) << 4);
    }
    // weight fragments: unit u = tap * G + g -> 1 KB at ((u * NB + cb) * 64 + lane) * 16 bytes
    const T* wl = (const T*)a.wfrag + ((size_t)cb * 64 + lane) * 8;
    auto load_w = [&](int u) -> V16 {
        V16 v;
        v.u = *(const uint4*)(wl + (size_t)u * NB * 64 * 8);
        return v;
    };
    constexpr int NU = 27 * G, WD = 4;
    V16 wf[WD + 1];
#pragma unroll
    for (int u = 0; u < WD; ++u) wf[u] = load_w(u);

    // ---- output offsets; the residual is requested before the tap loop when the register budget allows ----
    const size_t ldc = (size_t)a.ldc;
    const bool has_res = a.res != nullptr;
    auto out_off = [&](int f) -> size_t {
        const int td = p0 + (f >> 1), th = 4 * (f & 1) + (vl >> 3), tw = vl & 7;
        return ((((size_t)n * a.D + d0 + td) * a.H + h0 + th) * a.W + w0 + tw) * ldc + 32 * cb + 8 * hh;
    };
    uint4 rq[PRE_RES ? FR : 1][2];
    if (PRE_RES) {
#pragma unroll
        for (int f = 0; f < FR; ++f)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                rq[PRE_RES ? f : 0][q] = has_res ? *(const uint4*)((const T*)a.res + out_off(f) + 16 * q) : make_uint4(0, 0, 0, 0);
    }

    f32x16 acc[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;

    // the halo pieces are the oldest vector-memory operations of this wave: wait for everything once (weights and residual are
    // needed soon anyway), then the workgroup barrier publishes the image
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    V16 xa[2][FR];
    auto load_x = [&](auto uc, V16 (&dst)[FR]) {
        constexpr int u = decltype(uc)::value;
        constexpr int tap = u / G, g = u % G;
        constexpr int kd = tap / 9, khkw = tap % 9;
        const unsigned a0 = lp[khkw][0] ^ (g << 5), a1 = lp[khkw][1] ^ (g << 5);
        // planes whose offset does not fit the 16-bit ds_read immediate (256-byte voxels: 25600 B per plane) go through a
        // second base three planes up, so that no read needs its own address register
        constexpr int HI = 3 * PLANE_B;
        const unsigned b0 = a0 + HI, b1 = a1 + HI;
        static_for<0, FR>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr int off = ((f >> 1) + kd) * PLANE_B;
            if constexpr (off + 16 <= 65536) dst[f].u = *(const uint4*)((lptr_t)(size_t)(((f & 1) ? a1 : a0) + off));
            else dst[f].u = *(const uint4*)((lptr_t)(size_t)(((f & 1) ? b1 : b0) + (off - HI)));
        });
    };
    load_x(std::integral_constant<int, 0>{}, xa[0]);
    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if constexpr (u + WD < NU) wf[(u + WD) % (WD + 1)] = load_w(u + WD);
        if constexpr (u + 1 < NU) load_x(std::integral_constant<int, u + 1>{}, xa[(u + 1) & 1]);
#pragma unroll
        for (int f = 0; f < FR; ++f)
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u % (WD + 1)].h, xa[u & 1][f].h, acc[f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);               // keep the prefetch distances (the scheduler sinks the loads otherwise)
    });

    // ---- epilogue from the accumulators: lane (voxel, h) holds channels 32 cb + 8 h + e (e < 8) and 32 cb + 16 + 8 h + (e - 8) ----
    float esc[16], esf[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = 32 * cb + 16 * (e >> 3) + 8 * hh + (e & 7);
        const float bi = a.bias ? a.bias[c] : 0.f, sc = a.scale ? a.scale[c] : 1.f, sf = a.shift ? a.shift[c] : 0.f;
        esc[e] = sc; esf[e] = bi * sc + sf;
    }
    const EpiFloors fl = epi_floors(a.flags);
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const size_t oo = out_off(f);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint4 rv;
            if (PRE_RES) rv = rq[PRE_RES ? f : 0][q];
            else rv = has_res ? *(const uint4*)((const T*)a.res + oo + 16 * q) : make_uint4(0, 0, 0, 0);
            const unsigned rr[4] = {rv.x, rv.y, rv.z, rv.w};
            unsigned o[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int e = 8 * q + 2 * d;
                const float v0 = epi_apply(fmaf(acc[f][e], esc[e], esf[e]), fl, __uint_as_float(rr[d] << 16));
                const float v1 = epi_apply(fmaf(acc[f][e + 1], esc[e + 1], esf[e + 1]), fl, __uint_as_float(rr[d] & 0xffff0000u));
                o[d] = pack_bf16x2(v0, v1);
            }
            *(uint4*)((T*)a.y + oo + 16 * q) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// lt_conv_fwd packing [cout_pad][k_pad] (k = tap * cin + ci) -> fragments of the transposed product:
// [tap][cin / 16][cout_pad / 32][64 lanes][8]; lane (r = l & 31, h = l >> 5) holds row chan(r) + 32 block, K elements 16 g + 8 h .. + 7
__global__ void conv_pack_t32_kernel(const bf16_t* __restrict__ w, int cout_pad, int k_pad, int cin, int ntaps, bf16_t* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nbk = cout_pad / 32, ng = cin / 16;
    if (g >= (long long)ntaps * ng * nbk * 64) return;
    const int l = (int)(g & 63);
    long long ft = g >> 6;
    const int nb = (int)(ft % nbk); ft /= nbk;
    const int kg = (int)(ft % ng);
    const int tap = (int)(ft / ng);
    const int r = l & 31, h = l >> 5;
    const int chan = 32 * nb + 16 * (r >> 4) + 8 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1) + (r & 3);
    *(uint4*)(out + g * 8) = *(const uint4*)(w + (size_t)chan * k_pad + (size_t)tap * cin + 16 * kg + 8 * h);
}

// ---- 7^3 kernel with loader waves -------------------------------------------------------------------------------------------
// PMC on the ring version above (7^3 32->16 at 64^3): 3.3 VALU + 3.4 SALU + 1.3 LDS instructions per MFMA and no LDS bank
// conflicts -- with one workgroup per CU (the halo takes 123 KB) and so one compute wave per SIMD, the kernel was bound by
// the INSTRUCTIONS a wave issues between MFMAs (address arithmetic per chunk, the weight DMA and its ~100-cycle issue
// stalls, waitcnt bookkeeping): ~210 cycles per tap against 64 cycles of MFMA.  Here:
//   * waves 4-7 are loaders: halo + the whole weight stream (343 KB per tile, NBUF-deep chunk ring); waves 0-3 never issue
//     a memory instruction inside the tap loop;
//   * a chunk is one (kd, kh) row of 7 taps; kd is the only runtime loop: (kh, kw) offsets are ds_read immediates, the
//     fragment base registers are rebased once per kd (16 v_add per 49 taps), the weight buffer base once per chunk;
//   * fragments: hand-issued reads two taps ahead across chunk boundaries (15 reads in flight = the lgkmcnt counter).
// Measured (B = 16, 64^3): 2261 -> 1550 us (~105 cycles per tap; the LDS fragment reads alone need ~81: 20 ds_read_b128 per
// tap and CU).  Tried and dropped: a 5-deep instead of 4-deep weight ring (no change, kept), walking several tiles per
// workgroup with the next tile's first halo planes prefetched into the planes the tap loop has passed, and separate halo /
// weight loader waves (both ~10 % slower: the plane bursts delay the weight pieces queued behind them).  The next lever is
// LDS traffic: keeping the 7 kd-taps of one (kh, kw) in registers and sweeping the 10 input planes (17 reads per 28 MFMAs).
// (conv3d_halo7_kernel, the tap-major kernel these notes describe, was superseded by the kd-register-blocked variant below and removed
// in round 2; its measurements stay because they motivate that variant.)

// ---- 7^3 kernel, kd-register-blocked variant ------------------------------------------------------------------------------------
// conv3d_halo7_kernel reads five fragments from LDS per four MFMAs (one voxel fragment per output plane + the tap's weights) and
// is bound by those reads (81 of 105 cycles per tap).  Output plane td and tap plane kd meet in halo plane hd = td + kd, so for a
// fixed (kh, kw) ONE voxel fragment of plane hd serves every (td, kd) pair on that diagonal: a compute wave now owns two h rows
// of all four output planes (four accumulator tiles, one per td), keeps the seven kd weight fragments of the current (kh, kw)
// in registers and sweeps the ten halo planes -- 10 + 7 fragment reads per 28 MFMAs instead of 35.  A chunk of the weight ring
// is the seven kd slabs of one (kh, kw); the next chunk's slabs are read into the other register set while this one computes;
// the whole tap nest is unrolled (49 chunks), every LDS offset is an immediate.
constexpr int h7_lpos(int hd) { return hd < 6 ? hd : 2 * hd - 5; }                  // position of A_hd in a chunk's read stream
// global stream position of A_hd of chunk c (49 chunks, 17 entries each but the last, after the 7 entries of B(0)), and how far the
// stream must have been issued before that fragment is waited for
constexpr int h7_gpos(int c, int hd) { return 7 + 17 * c + (c == 48 ? hd : h7_lpos(hd)); }
constexpr int h7_target(int c, int hd, int look, int gend) { return h7_gpos(c, hd) + 1 + look < gend ? h7_gpos(c, hd) + 1 + look : gend; }
template <typename T>
__global__ __launch_bounds__(512) void conv3d_halo7b_kernel(const HaloArgs a) {
    constexpr int KS = 7, CIN = 32, CP = 16, TD = 4, TH = 8, TW = 8, TPC = 7, NBUF = 5;
    typedef HaloCfg<T, KS, CIN, CP, TD, TH, TW, TPC, 4> C;
    static_assert(sizeof(T) == 2, "bf16 only");
    constexpr int MF = C::MF, NVV = C::NVV, VPR = C::VPR, CINB = C::CINB;
    static_assert(MF == 16 && NVV == 4 && C::SW::FSH == 0 && C::SW::FA == 0 && C::SW::FC == 0 && C::SLAB == 1024 && C::WCH == TPC * 1024, "7^3 32->16 layout");
    static_assert(C::HALO_BYTES + NBUF * C::WCH <= 160 * 1024, "halo + weight ring must fit LDS");
    typedef typename Mma<T, MF>::acc_t acc_t;
    constexpr int NCH = KS * KS;                         // 49 chunks: c = kh*7 + kw, each the 7 kd slabs of that (kh, kw)
    static_assert(NCH == 49, "h7_gpos assumes 49 chunks");
    constexpr int KDSTEP = C::HH * C::PW * CINB;         // bytes per d-plane of the halo image
    constexpr int NI_H = C::HALO_BYTES / 1024;
    constexpr int HDN = TD + KS - 1;                     // 10 halo planes

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lds_w = lds0 + C::HALO_BYTES;

    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page_h;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool loader = wave >= 4;
    const int wl = wave & 3;
    constexpr int P = KS / 2;

    const int tps = a.tiles_d * a.tiles_h * a.tiles_w;
    int n, tix;
    if (a.xcd_pin) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        n = xcd + 8 * (j / tps);
        tix = j % tps;
    } else {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        n = lin / tps;
        tix = lin % tps;
    }
    const int w0 = (tix % a.tiles_w) * TW;
    const int h0 = ((tix / a.tiles_w) % a.tiles_h) * TH;
    const int d0 = (tix / (a.tiles_w * a.tiles_h)) * TD;

    if (loader) {
        // ================================= loader waves =================================
        const T* __restrict__ x = (const T*)a.x + (size_t)n * a.D * a.H * a.W * CIN;
        const T* __restrict__ w = (const T*)a.w;
        for (int i = wl; i < NI_H; i += 4) {
            const int q = i * 64 + lane;
            const int hv = q / NVV, pv = q % NVV;
            const int hw_ = hv % C::PW, hh_ = (hv / C::PW) % C::HH, hd_ = hv / (C::PW * C::HH);
            const int lv = pv ^ C::fswz(hd_, hh_, hw_);
            const int id = d0 - P + hd_, ih = h0 - P + hh_, iw = w0 - P + hw_;
            const bool ok = hv < C::HV && hw_ < C::HW && ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            const void* src = ok ? (const void*)(x + (((size_t)id * a.H + ih) * a.W + iw) * CIN + lv * C::VEC) : zero_page;
            dma16h(src, lds0 + i * 1024);
        }
        // chunk c = (kh, kw): slab kd is tap kd*49 + c; this loader carries kd = wl (and wl + 4 when < 7)
        const int pv = lane % NVV, col = lane / NVV;
        const int lv = pv ^ ((-(col / VPR)) & (NVV - 1));
        const T* wsrc0 = w + (size_t)col * a.k_pad + (size_t)wl * NCH * CIN + lv * C::VEC;
        const T* wsrc1 = wsrc0 + (size_t)4 * NCH * CIN;
        const bool two = wl + 4 < KS;
        const int dpc = two ? 2 : 1;
        auto stage_w = [&](int c) {
            const unsigned dst = lds_w + (c % NBUF) * C::WCH;
            dma16h(wsrc0 + (size_t)c * CIN, dst + wl * 1024);
            if (two) dma16h(wsrc1 + (size_t)c * CIN, dst + (wl + 4) * 1024);
        };
#pragma unroll
        for (int c = 0; c < NBUF - 1; ++c) stage_w(c);
        for (int c = 0; c < NCH; ++c) {
            // chunks <= c+1 (and, before them, the halo) must have landed; at most NBUF-3 younger chunks stay in flight
            int younger = NCH - 2 - c;
            if (younger > NBUF - 3) younger = NBUF - 3;
            if (younger < 0) younger = 0;
            wait_vmcnt_h(younger * dpc);
            asm volatile("s_barrier" ::: "memory");
            if (c + NBUF - 1 < NCH) stage_w(c + NBUF - 1);
        }
        asm volatile("s_barrier" ::: "memory");          // the compute waves' "done with the LDS images" barrier
        return;
    }

    // ================================= compute waves =================================
    // wave -> output rows th = 2 wave + (r15 >> 3), tw = r15 & 7 of ALL four planes td (accumulator tile td)
    const int r15 = lane & 15, lvb = lane >> 4;
    const int th = 2 * wave + (r15 >> 3), tw = r15 & 7;
    float cbi, csc, csf;
    cbi = a.bias ? a.bias[r15] : 0.f; csc = a.scale ? a.scale[r15] : 1.f; csf = a.shift ? a.shift[r15] : 0.f;
    constexpr int E_LPR = CP / C::VEC, E_RPP = 64 / E_LPR;   // 2 lanes per output row, 32 rows per iteration, 2 iterations
    const bool vec_epi = (a.Cout % C::VEC == 0) && (a.ldc % C::VEC == 0);
    const int cq = (lane % E_LPR) * C::VEC;
    auto row_off = [&](int row) -> size_t {               // staging row = td*16 + rr  ->  element offset of (voxel, channel cq)
        const int td = row >> 4, rr = row & 15;
        return ((((size_t)n * a.D + d0 + td) * a.H + h0 + 2 * wave + (rr >> 3)) * a.W + w0 + (rr & 7)) * a.ldc + cq;
    };
    const bool has_res = a.res != nullptr;
    uint4 rp0 = make_uint4(0, 0, 0, 0), rp1 = rp0;
    if (vec_epi && has_res && cq < a.Cout) {
        rp0 = *(const uint4*)((const T*)a.res + row_off(lane / E_LPR));
        rp1 = *(const uint4*)((const T*)a.res + row_off(lane / E_LPR + E_RPP));
    }
    unsigned base0[4], base1[4];                         // halo fragment bases per swizzle variant (kw & 3): planes 0-4 / 5-9
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) {
        base0[vv] = lds0 + (th * C::PW + tw) * CINB + ((lvb ^ ((tw + vv) & 3)) << 4);
        base1[vv] = base0[vv] + 5 * KDSTEP;
    }
    const int bsw = (-(r15 / VPR)) & (NVV - 1);
    const unsigned bbase = lds_w + r15 * CINB + ((lvb ^ bsw) << 4);

    acc_t acc[TD];
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;

    // global read stream: [B(0) x 7] + per chunk c < 48: [A0..A5, B'0, A6, B'1, A7, B'2, A8, B'3, A9, B'4, B'5, B'6] (B' = chunk c+1)
    //                                + last chunk: [A0..A9];  LOOK entries of lookahead (the first B' of a chunk sits at position 6,
    //                                so the lookahead out of the previous chunk never reaches weights that may not have landed)
    constexpr int LOOK = 6, SLEN = 17, GEND = 7 + SLEN * (NCH - 1) + HDN;
    V16 fa[HDN], fbk[2][KS];
    auto issue = [&](auto gic) {
        constexpr int gi = decltype(gic)::value;
        if constexpr (gi < 7) {
            lds_read16<gi * 1024>(fbk[0][gi], bbase);                              // chunk 0 lives in ring slot 0
        } else {
            constexpr int c = (gi - 7) / SLEN < NCH - 1 ? (gi - 7) / SLEN : NCH - 1;
            constexpr int q = gi - 7 - c * SLEN;
            constexpr int kh = c / KS, kw = c % KS;
            constexpr bool isA = q < 6 || c == NCH - 1 || (q <= 13 && (q & 1));   // positions 7, 9, 11, 13 are A6..A9
            if constexpr (isA) {
                constexpr int hd = (q < 6 || c == NCH - 1) ? q : (q + 5) / 2;
                constexpr int imm = (hd < 5 ? hd : hd - 5) * KDSTEP + (kh * C::PW + kw) * CINB;
                if constexpr (hd < 5) lds_read16<imm>(fa[hd], base0[kw & 3]);
                else lds_read16<imm>(fa[hd], base1[kw & 3]);
            } else {
                constexpr int kd = q <= 12 ? (q - 6) / 2 : q - 10;                 // 6,8,10,12 -> 0..3; 14,15,16 -> 4..6
                lds_read16<((c + 1) % NBUF) * C::WCH + kd * 1024>(fbk[(c + 1) & 1][kd], bbase);
            }
        }
    };
    static_for<0, NCH>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        asm volatile("s_barrier" ::: "memory");          // the loaders have chunks <= c+1 (and, the first time, the halo) in LDS
        if constexpr (c == 0) {
            static_for<0, 7 + LOOK>([&](auto gic) { issue(gic); });   // B(0) and the first LOOK entries of chunk 0
        }
        static_for<0, HDN>([&](auto hc) {
            constexpr int hd = decltype(hc)::value;
            constexpr int gp = h7_gpos(c, hd);
            constexpr int prev_target = (c == 0 && hd == 0) ? 7 + LOOK : (hd == 0 ? h7_target(c - 1, HDN - 1, LOOK, GEND) : h7_target(c, hd - 1, LOOK, GEND));
            constexpr int target = h7_target(c, hd, LOOK, GEND);
            static_for<prev_target, target>([&](auto gic) { issue(gic); });
            lgkm_wait<target - gp - 1>();
            frag_ready(fa[hd]);
            if constexpr (hd == 0) {
#pragma unroll
                for (int kd = 0; kd < KS; ++kd) frag_ready(fbk[c & 1][kd]);
            }
#pragma unroll
            for (int td = 0; td < TD; ++td) {
                if (hd - td >= 0 && hd - td < KS) LT_HMMA(acc[td], fa[hd], fbk[c & 1][hd - td]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the halo / weight images

    // ---- epilogue: this wave's 64 rows (td*16 + rr) x 16 channels through a private fp32 LDS tile ----
    constexpr int EP_LD = CP + 4;
    float* ep = (float*)(smem + wave * (64 * EP_LD * 4));
#pragma unroll
    for (int td = 0; td < TD; ++td)
#pragma unroll
        for (int e = 0; e < 4; ++e) ep[(td * 16 + (lane >> 4) * 4 + e) * EP_LD + r15] = (acc[td][e] + cbi) * csc + csf;
    const EpiFloors fl = epi_floors(a.flags);
    if (vec_epi) {
        if (cq < a.Cout) {
            const unsigned no_res = has_res ? 0u : 0x80008000u;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = lane / E_LPR + it * E_RPP;
                const uint4 resv = it == 0 ? rp0 : rp1;
                const float* src = ep + row * EP_LD + cq;
                const float4 q0 = *(const float4*)src, q1 = *(const float4*)(src + 4);
                const float vv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const unsigned ru[4] = {resv.x | no_res, resv.y | no_res, resv.z | no_res, resv.w | no_res};
                unsigned ou[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ou[e] = pack_bf16x2(epi_apply(vv[2 * e], fl, __uint_as_float(ru[e] << 16)), epi_apply(vv[2 * e + 1], fl, __uint_as_float(ru[e] & 0xffff0000u)));
                *(uint4*)((T*)a.y + row_off(row)) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
            }
        }
    } else {
        for (int idx = lane; idx < 64 * CP; idx += 64) {
            const int row = idx / CP, cc = idx - row * CP;
            if (cc >= a.Cout) continue;
            const size_t off = row_off(row) - cq + cc;
            const float rr = has_res ? elt<T>::ld((const T*)a.res + off) : -0.0f;
            elt<T>::st((T*)a.y + off, epi_apply(ep[row * EP_LD + cc], fl, rr));
        }
    }
}

int launch_halo7(const HaloArgs& a, hipStream_t s) {
    typedef HaloCfg<bf16_t, 7, 32, 16, 4, 8, 8, 7, 4> C;
    constexpr int LDS = C::HALO_BYTES + 5 * C::WCH;
    const long long nblk = (long long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
    auto kern_b = conv3d_halo7b_kernel<bf16_t>;
    LT_OPT_IN_LDS(kern_b, 160 * 1024);
    hipLaunchKernelGGL(kern_b, dim3((unsigned)nblk), dim3(512), LDS, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(halo 7^3, kd-blocked)");
    return LT_OK;
}

template <typename T, int CIN, int CP>
int launch_halo_persist(const HaloArgs& a, hipStream_t s) {
    typedef HaloCfg<T, 3, CIN, CP, 4, 8, 8, 9, 2> C;
    constexpr int W_BYTES = ((C::NTAPS * C::SLAB + 1023) / 1024) * 1024;
    constexpr int LDS = W_BYTES + 2 * C::HALO_BYTES;
    static_assert(LDS <= 160 * 1024, "weights + two halo buffers do not fit LDS");
    auto kern = conv3d_halo_persist_kernel<T, CIN, CP>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    const int n_cu = lt::device_cu_count8();
    const long long total = (long long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
    const int grid = (int)(total < n_cu ? total - total % 8 : n_cu);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, a, (int)total);
    LT_CHECK_LAUNCH("lt_conv_fwd(halo, persistent)");
    return LT_OK;
}

template <typename T, int CIN, int CP>
int launch_halo_wreg(const HaloArgs& a, hipStream_t s) {
    typedef HaloCfg<T, 3, CIN, CP, 4, 8, 8, 3, 3> C;
    static_assert(C::HALO_BYTES <= 160 * 1024, "the halo must fit LDS");
    auto kern = conv3d_halo_wreg_kernel<T, CIN, CP>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    const long long nblk = (long long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), C::HALO_BYTES, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(halo, weights from registers)");
    return LT_OK;
}

template <typename T>
int launch_halo_col(const HaloArgs& a, hipStream_t s) {
    typedef HaloCfg<T, 3, 32, 32, 4, 8, 8, 9, 2> C;
    constexpr int W_BYTES = ((C::NTAPS * C::SLAB + 1023) / 1024) * 1024;
    constexpr int LDS = W_BYTES + 4 * 4 * C::HH * C::PW * C::CINB;
    const int n_cu = lt::device_cu_count8();
    const int total_cols = a.N * a.tiles_h * a.tiles_w;  // % 8 == 0 (checked by the caller)
    const int grid = total_cols < n_cu ? total_cols : n_cu;
    if (a.skip_x) {
        auto kern = conv3d_halo_col_kernel<T, false, true>;
        LT_OPT_IN_LDS(kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, a, total_cols);
    } else if (a.flags & LT_EPI_STORE_F32) {
        auto kern = conv3d_halo_col_kernel<T, true, false>;
        LT_OPT_IN_LDS(kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, a, total_cols);
    } else {
        auto kern = conv3d_halo_col_kernel<T, false, false>;
        LT_OPT_IN_LDS(kern, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, a, total_cols);
    }
    LT_CHECK_LAUNCH("lt_conv_fwd(halo, column walk)");
    return LT_OK;
}

template <typename T, int KS, int CIN, int CP, int TD, int TH, int TW, int TPC, int NBUF, int PD, bool LDR, int NPH = 1>
int launch_halo(const HaloArgs& a, hipStream_t s) {
    typedef HaloCfg<T, KS, CIN, CP, TD, TH, TW, TPC, NBUF> C;
    static_assert(C::LDS_BYTES <= 160 * 1024, "halo tile does not fit LDS");
    auto kern = conv3d_halo_kernel<T, KS, CIN, CP, TD, TH, TW, TPC, NBUF, PD, LDR, NPH>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    const long long nblk = (long long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(LDR ? 512 : 256), C::LDS_BYTES, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(halo)");
    return LT_OK;
}

}  // namespace

namespace lt {

// Returns 1 and launches when the problem matches one of the instantiated halo configurations, 0 when the caller should
// fall back to the implicit-GEMM path, negative on error.
int conv3d_halo_try(int dtype, const ConvArgs& c, int cout_pad, int nphase, bool forced, hipStream_t s) {
    const PhaseArg& p0 = c.phase[0];
    if (nphase != 1 || c.sd != 1 || c.sh != 1 || c.sw != 1 || c.osd != 1 || c.osh != 1 || c.osw != 1) return 0;
    if (p0.ood || p0.ooh || p0.oow || c.D != c.Do || c.H != c.Ho || c.W != c.Wo || c.OD != c.Do || c.OH != c.Ho || c.OW != c.Wo) return 0;
    if (c.flags & LT_EPI_SIGMOID) return 0;
    // fp32 output (LT_EPI_STORE_F32: the mixed-precision training step) exists in the column-walk kernel only, and without a residual (the kernels read
    // the residual in the activation type)
    const bool f32_out = (c.flags & LT_EPI_STORE_F32) != 0;
    if (f32_out && (c.res || dtype != LT_BF16)) return 0;
    int ks = 0;
    if (p0.ntaps == 27 && c.pd == 1 && c.ph == 1 && c.pw == 1) ks = 3;
    else if (p0.ntaps == 343 && c.pd == 3 && c.ph == 3 && c.pw == 3) ks = 7;
    else return 0;
    if (c.D % 4 || c.H % 8 || c.W % 8) return 0;
    const long long nblk = (long long)c.N * (c.D / 4) * (c.H / 8) * (c.W / 8);
    if (nblk < 256 && !forced) return 0;   // too few workgroups: the 64x64 implicit-GEMM tile fills the chip better
    HaloArgs a;
    a.x = c.x; a.w = p0.w; a.wfrag = p0.wfrag_t; a.y = c.y; a.res = c.res; a.bias = c.bias; a.scale = c.scale; a.shift = c.shift;
    a.N = c.N; a.D = c.D; a.H = c.H; a.W = c.W; a.Cout = c.Cout; a.ldc = c.ldc; a.k_pad = c.k_pad; a.flags = c.flags;
    a.tiles_d = c.D / 4; a.tiles_h = c.H / 8; a.tiles_w = c.W / 8;
    a.xcd_pin = (c.N % 8 == 0) ? 1 : 0;
    a.skip_x = c.skip_x; a.skip_w = c.skip_w;
    const bool bf = dtype == LT_BF16, f8 = dtype == LT_FP8;
    if (c.skip_x && (f32_out || c.res || (c.flags & LT_EPI_RELU_PRE) || !bf)) return LT_ERR_UNSUPPORTED;
#define HALO_CASE_L(T_, KS_, CIN_, CP_, TPC_, NBUF_, PD_, LDR_)                                  \
    if (ks == KS_ && c.Cin == CIN_ && cout_pad == CP_) {                                        \
        int rc = launch_halo<T_, KS_, CIN_, CP_, 4, 8, 8, TPC_, NBUF_, PD_, LDR_>(a, s);        \
        return rc == LT_OK ? 1 : rc;                                                            \
    }
#define HALO_CASE(T_, KS_, CIN_, CP_, TPC_, NBUF_, PD_)                                          \
    if (ks == KS_ && c.Cin == CIN_ && cout_pad == CP_) {                                        \
        int rc = launch_halo<T_, KS_, CIN_, CP_, 4, 8, 8, TPC_, NBUF_, PD_, false>(a, s);       \
        return rc == LT_OK ? 1 : rc;                                                            \
    }
    // persistent variant: needs a few tiles per workgroup to amortise the weight load, and total % 8 == 0 for the XCD dealing
    static const bool no_persist = getenv("LT_HALO_NO_PERSIST") != nullptr;   // A/B
    if (bf && ks == 3 && cout_pad == 32 && c.Cout == 32 && c.ldc % 4 == 0 && c.Cin == 32 && nblk >= 1024 && nblk % 8 == 0 && !no_persist) {
        // column walk (sliding plane window + overlapped epilogue) when every workgroup gets whole columns of >= 2 tiles
        const char* nocol = getenv("LT_HALO_NO_COL");    // A/B, read per call
        const long long cols = (long long)c.N * a.tiles_h * a.tiles_w;
        if (!nocol && a.tiles_d >= 2 && cols % 8 == 0 && cols >= 256 && c.ldc % 8 == 0) {
            int rc = launch_halo_col<bf16_t>(a, s);
            return rc == LT_OK ? 1 : rc;
        }
        if (c.skip_x) return LT_ERR_UNSUPPORTED;
        if (f32_out) return 0;
        int rc = launch_halo_persist<bf16_t, 32, 32>(a, s);
        return rc == LT_OK ? 1 : rc;
    }
    if (c.skip_x) return LT_ERR_UNSUPPORTED;             // the computed residual exists in the column-walk kernel only
    if (f32_out) return 0;
    // halo-only LDS, weights as fragments from global memory (lt_conv_pack_weights_t32): 64 -> 64, 32 -> 64, 128 -> 128
    if (bf && ks == 3 && c.Cout == cout_pad && c.ldc % 8 == 0 && a.wfrag && !getenv("LT_HALO_NO_WREG")) {
        int rc = 1;
        if (c.Cin == 64 && cout_pad == 64) rc = launch_halo_wreg<bf16_t, 64, 64>(a, s);
        else if (c.Cin == 16 && cout_pad == 32 && !getenv("LT_HALO_NO_WREG16")) rc = launch_halo_wreg<bf16_t, 16, 32>(a, s);
        else if (c.Cin == 32 && cout_pad == 64) rc = launch_halo_wreg<bf16_t, 32, 64>(a, s);
        else if (c.Cin == 128 && cout_pad == 128) rc = launch_halo_wreg<bf16_t, 128, 128>(a, s);
        if (rc != 1) return rc == LT_OK ? 1 : rc;
    }
    static const bool row_chunks = getenv("LT_HALO_ROW") != nullptr;   // A/B: 3-tap weight chunks -> 51 KB of LDS -> 3 workgroups per CU
    if (f8) {
        // e4m3 operands (train_precision 'fp8v2v'): the one-tile loader-wave kernel at half the bytes per voxel / per weight slab; bf16 stores.  The byte
        // geometries are the bf16 ones of half the channel count: (64, 64) = bf16 (32 -> 64), (32, 32) = bf16 (16 -> 32)
        if (c.Cout % 8 || c.ldc % 8 || getenv("LT_HALO_NO_FP8")) return 0;
        HALO_CASE_L(fp8_t, 3, 64, 64, 9, 2, 1, true)
        HALO_CASE_L(fp8_t, 3, 32, 32, 9, 2, 1, true)
        HALO_CASE_L(fp8_t, 3, 32, 64, 9, 2, 1, true)
        return 0;
    }
    if (bf) {
        static const bool no_ldr = getenv("LT_HALO_NO_LDR") != nullptr;   // A/B: no loader waves in the one-tile kernel
        if (!no_ldr) {
            HALO_CASE_L(bf16_t, 3, 64, 64, 3, 3, 1, true)
            HALO_CASE_L(bf16_t, 3, 32, 32, 9, 2, 1, true)
            HALO_CASE_L(bf16_t, 3, 16, 32, 9, 2, 1, true)
            HALO_CASE_L(bf16_t, 3, 32, 64, 9, 2, 1, true)
        }
        if (row_chunks) { HALO_CASE(bf16_t, 3, 32, 32, 3, 2, 1) }
        HALO_CASE(bf16_t, 3, 32, 32, 9, 2, 1)
        HALO_CASE(bf16_t, 3, 16, 32, 9, 2, 1)
        HALO_CASE(bf16_t, 3, 64, 64, 3, 2, 1)
        HALO_CASE(bf16_t, 3, 32, 64, 9, 2, 1)
        static const bool no_ring = getenv("LT_HALO_NO_RING") != nullptr;   // A/B: 1-tap fragment lookahead for 7^3
        static const bool no_h7 = getenv("LT_HALO_NO_H7") != nullptr;       // A/B: no loader-wave 7^3 kernel
        if (no_ring) { HALO_CASE(bf16_t, 7, 32, 16, 7, 4, 1) }
        if (ks == 7 && c.Cin == 32 && cout_pad == 16 && !no_h7) {
            int rc = launch_halo7(a, s);
            return rc == LT_OK ? 1 : rc;
        }
        HALO_CASE(bf16_t, 7, 32, 16, 7, 4, 2)
        // round 6: 7^3 16 -> 32 = the INPUT GRADIENT of the front layer in the 16-bit training step (the flipped / transposed filter): it ran on the generic
        // 256 x 32 implicit-GEMM tile with four taps per 128-byte K step (1.78 ms at 8 samples, 4.2 % of the step's kernel time); LT_HALO_NO_D7=1: that tile again (A/B)
        if (!getenv("LT_HALO_NO_D7")) { HALO_CASE(bf16_t, 7, 16, 32, 7, 4, 2) }
    } else {
        // round 6: 3^3 32 -> 32 (nine layers at 64^3, 23 % of the exact-fp32 forward) in two channel phases of 16: 38 KB of halo + 18 KB of weight ring instead of
        // 77 + 36 KB -> TWO workgroups per CU = two waves per SIMD, so that one tile's halo load, chunk barriers, fp64 flushes and epilogue run under the other
        // tile's MFMAs (with one workgroup per CU all of that was exposed: 59 % of the fp32 MFMA peak).  LT_HALO_F3=1: the one-phase kernel; =9: plane chunks
        {
            const char* f3 = getenv("LT_HALO_F3");
            if (ks == 3 && c.Cin == 32 && cout_pad == 32 && !(f3 && f3[0] == '1')) {
                int rc = (f3 && f3[0] == '9') ? launch_halo<float, 3, 16, 32, 4, 8, 8, 9, 2, 1, false, 2>(a, s)
                                              : launch_halo<float, 3, 16, 32, 4, 8, 8, 3, 3, 1, false, 2>(a, s);
                return rc == LT_OK ? 1 : rc;
            }
        }
        HALO_CASE(float, 3, 32, 32, 3, 3, 1)
        HALO_CASE(float, 3, 16, 32, 9, 2, 1)
        // round 6: the 7^3 layers of the exact-fp32 kernel set.  32 -> 16 (V2V's front layer, 18 % of the fp32 forward on the generic 256 x 16 tile at 51 % of
        // the fp32 MFMA peak): two channel phases of 16 over a 123 KB halo image (NPH = 2, see the kernel).  16 -> 32 (its input gradient in the fp32
        // training step): one phase, one-tap lookahead, two weight buffers (151 KB).  LT_HALO_NO_F7=1: the generic tiles again (A/B).
        if (ks == 7 && !getenv("LT_HALO_NO_F7")) {
            if (c.Cin == 32 && cout_pad == 16) {
                int rc = launch_halo<float, 7, 16, 16, 4, 8, 8, 7, 4, 2, false, 2>(a, s);
                return rc == LT_OK ? 1 : rc;
            }
            HALO_CASE(float, 7, 16, 32, 7, 2, 1)
        }
    }
#undef HALO_CASE
#undef HALO_CASE_L
    return 0;
}

}  // namespace lt

#ifdef LT_TRACE
extern "C" int lt_trace_read_halo(long long* dst, int n) {
    if (n > 8 * 64) n = 8 * 64;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace_h), (size_t)n * sizeof(long long)) != hipSuccess) return -2;
    return n;
}
#endif

extern "C" int lt_conv_pack_weights_t32(const void* weight, int32_t cout_pad, int32_t k_pad, int32_t cin, int32_t ntaps, void* packed,
                                        void* stream) {
    LT_REQUIRE(weight && packed, LT_ERR_INVALID, "lt_conv_pack_weights_t32: null argument");
    LT_REQUIRE(cout_pad >= 32 && cout_pad % 32 == 0 && cin >= 16 && cin % 16 == 0 && ntaps >= 1 && (long long)ntaps * cin <= k_pad && k_pad % 8 == 0,
               LT_ERR_INVALID, "lt_conv_pack_weights_t32: cout_pad %d / cin %d / ntaps %d / k_pad %d", cout_pad, cin, ntaps, k_pad);
    const long long total = (long long)ntaps * (cin / 16) * (cout_pad / 32) * 64;
    hipLaunchKernelGGL(conv_pack_t32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)weight,
                       cout_pad, k_pad, cin, ntaps, (bf16_t*)packed);
    LT_CHECK_LAUNCH("lt_conv_pack_weights_t32");
    return LT_OK;
}
