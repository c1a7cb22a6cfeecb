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
#include <stdlib.h>

#include "conv_common.h"

using namespace lt;

namespace {

__device__ uint4 g_zero_page_h[2];

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

__device__ __forceinline__ float epi_act_h(float v, bool relu_pre, bool has_res, float r, bool relu_post) {
    if (relu_pre) v = fmaxf(v, 0.f);
    if (has_res) v += r;
    if (relu_post) v = fmaxf(v, 0.f);
    return v;
}

// swizzle constants {FA, FB, FC, FSH, pitch multiple}
template <int KS, int CINB, int MF> struct HaloSwz { static constexpr int FA = 0, FB = 0, FC = 0, FSH = 1, PAD = 1; };   // (3, 64 B, 32x32)
template <> struct HaloSwz<7, 64, 16> { static constexpr int FA = 0, FB = 0, FC = 0, FSH = 0, PAD = 1; };
template <> struct HaloSwz<3, 128, 32> { static constexpr int FA = 3, FB = 0, FC = 2, FSH = 1, PAD = 1; };
template <> struct HaloSwz<3, 32, 32> { static constexpr int FA = 0, FB = 0, FC = 0, FSH = 2, PAD = 4; };

template <typename T, int KS, int CIN, int CP, int TD, int TH, int TW, int TPC, int NBUF>
struct HaloCfg {
    static constexpr int ES = sizeof(T);
    static constexpr int VEC = 16 / ES;
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
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // conservative
    }
}

struct HaloArgs {
    const void* x;
    const void* w;      // [cout_pad][k_pad], k = tap*Cin + ci (the lt_conv_fwd packing)
    void* y;
    const void* res;
    const float* bias;
    const float* scale;
    const float* shift;
    int N, D, H, W, Cout, ldc, k_pad, flags;
    int tiles_d, tiles_h, tiles_w;   // per sample
    int xcd_pin;
};

template <typename T, int KS, int CIN, int CP, int TD, int TH, int TW, int TPC, int NBUF>
__global__ __launch_bounds__(256) void conv3d_halo_kernel(const HaloArgs a) {
    typedef HaloCfg<T, KS, CIN, CP, TD, TH, TW, TPC, NBUF> C;
    constexpr bool ACC64 = sizeof(T) == 4;
    constexpr int MF = C::MF, SM = C::SM, SN = C::SN, G = C::G, NACC = C::NACC, NVV = C::NVV, VPR = C::VPR, CINB = C::CINB;
    typedef typename Mma<T, MF>::acc_t acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned char* s_halo = smem;
    unsigned char* s_w = smem + C::HALO_BYTES;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    // ---- workgroup -> (sample, tile); with N % 8 == 0 sample n is pinned to XCD n % 8 (its d-slabs stay in that L2) ----
    const int tps = a.tiles_d * a.tiles_h * a.tiles_w;
    int n, tix;
    if (a.xcd_pin) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        n = xcd + 8 * (j / tps);
        tix = j % tps;
    } else {
        n = blockIdx.x / tps;
        tix = blockIdx.x % tps;
    }
    const int w0 = (tix % a.tiles_w) * TW;
    const int h0 = ((tix / a.tiles_w) % a.tiles_h) * TH;
    const int d0 = (tix / (a.tiles_w * a.tiles_h)) * TD;
    constexpr int P = KS / 2;

    const T* __restrict__ x = (const T*)a.x + (size_t)n * a.D * a.H * a.W * CIN;
    const T* __restrict__ w = (const T*)a.w;

    // ---- halo DMA: vector q = hv*NVV + pv, wave-instruction i covers q in [64 i, 64 i + 64) ----
    constexpr int NI_H = C::HALO_BYTES / 1024;
    for (int i = wave; i < NI_H; i += 4) {
        const int q = i * 64 + lane;
        const int hv = q / NVV, pv = q % NVV;
        const int hw_ = hv % C::PW, hh_ = (hv / C::PW) % C::HH, hd_ = hv / (C::PW * C::HH);
        const int lv = pv ^ C::fswz(hd_, hh_, hw_);
        const int id = d0 - P + hd_, ih = h0 - P + hh_, iw = w0 - P + hw_;
        const bool ok = hv < C::HV && hw_ < C::HW && ((unsigned)id < (unsigned)a.D) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        const void* src = ok ? (const void*)(x + (((size_t)id * a.H + ih) * a.W + iw) * CIN + lv * C::VEC) : (const void*)g_zero_page_h;
        dma16h(src, lds0 + i * 1024);
    }
    // ---- weight chunk DMA: vector q = (tap_in_chunk*CP + col)*NVV + pv ----
    constexpr int NI_W = C::WCH / 1024;
    auto stage_w = [&](int ch, int buf) {
        for (int i = wave; i < NI_W; i += 4) {
            const int q = i * 64 + lane;
            const int pv = q % NVV, col = (q / NVV) % CP, tj = q / (NVV * CP);
            const int tap = ch * TPC + tj;
            const int lv = pv ^ ((-(col / VPR)) & (NVV - 1));
            const bool ok = tj < TPC && tap < C::NTAPS;
            const void* src = ok ? (const void*)(w + (size_t)col * a.k_pad + tap * CIN + lv * C::VEC) : (const void*)g_zero_page_h;
            dma16h(src, lds0 + C::HALO_BYTES + buf * C::WCH + i * 1024);
        }
    };
    const int dpc = (NI_W - wave + 3) / 4;        // weight DMA instructions per chunk issued by this wave (wave-uniform)
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < C::NCH) stage_w(c, c);

    // ---- residual prefetch: the lane's 16-byte residual vectors (up to 8) are requested before the tap loop, in named
    // registers (see conv_igemm2.hip for why not an array); they are consumed in the epilogue ----
    constexpr int E_VECO = C::VEC, E_LPR = CP / E_VECO, E_RPP = 64 / E_LPR, E_NIT = 64 / E_RPP;
    static_assert(E_NIT <= 16, "epilogue rows per lane");
    constexpr bool PRE_OK = E_NIT <= 8;
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
            const void* src = cqp < a.Cout ? (const void*)((const T*)a.res + pixv * a.ldc + cqp) : (const void*)g_zero_page_h;
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

    for (int ch = 0; ch < C::NCH; ++ch) {
        // chunk ch (and, the first time, the halo issued before it) must have landed; up to NBUF-2 younger chunks stay in flight
        int younger = C::NCH - 1 - ch;
        if (younger > NBUF - 2) younger = NBUF - 2;
        wait_vmcnt_h(younger * dpc);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all waves: chunk ch landed, chunk ch-1 fully consumed
        if (ch + NBUF - 1 < C::NCH) stage_w(ch + NBUF - 1, (ch + NBUF - 1) % NBUF);
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
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < SM; ++i)
#pragma unroll
                        for (int j = 0; j < SN; ++j) Mma<T, MF>::run(acc[i][j], fa[tj & 1][g][i], fb[tj & 1][g][j]);
                __builtin_amdgcn_sched_barrier(0);
                if (ACC64 && ((tap & 1) == 1 || tap + 1 == C::NTAPS)) {
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the halo / weight images

    // ---- epilogue (same scheme as conv_igemm2: per-wave fp32 LDS tile -> 16-byte vectors) ----
    float* ep = (float*)(smem + wave * (64 * C::EP_LD * 4));
#pragma unroll
    for (int j = 0; j < SN; ++j) {
        const int colj = j * MF + (lane & (MF - 1));
        const float bi = a.bias ? a.bias[colj] : 0.f;
        const float sc = a.scale ? a.scale[colj] : 1.f;
        const float sf = a.shift ? a.shift[colj] : 0.f;
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                const int r = i * MF + ((MF == 32) ? ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) : ((lane >> 4) * 4 + e));
                float val;
                if (ACC64) val = (float)((dacc[i][j][e] + (double)bi) * (double)sc + (double)sf);
                else val = (acc[i][j][e] + bi) * sc + sf;
                ep[r * C::EP_LD + colj] = val;
            }
    }
    __syncthreads();

    const bool relu_pre = a.flags & LT_EPI_RELU_PRE, relu_post = a.flags & LT_EPI_RELU_POST;
    const bool has_res = a.res != nullptr;
    auto row_pix = [&](int r) -> size_t {   // r = row inside the workgroup tile
        const int tw = r % TW, th = (r / TW) % TH, td = r / (TW * TH);
        return (((size_t)n * a.D + d0 + td) * a.H + h0 + th) * a.W + w0 + tw;
    };
    constexpr int VECO = C::VEC;              // fp32: 4 channels, bf16: 8 channels per 16 bytes
    if ((a.Cout % VECO == 0) && (a.ldc % VECO == 0)) {
        constexpr int LPR = CP / VECO, RPP = 64 / LPR;
        const int cq = (lane % LPR) * VECO;
        if (cq < a.Cout) {
            constexpr int NIT = 64 / RPP;
            union Pack { uint4 u; float f[4]; unsigned short h[8]; };
            auto row = [&](int it, uint4 resv) {      // resv: this row's residual vector (zeros when there is none)
                const size_t off = row_pix(64 * wave + lane / LPR + it * RPP) * a.ldc + cq;
                const float* src = ep + (lane / LPR + it * RPP) * C::EP_LD + cq;
                Pack rv, ov;
                rv.u = resv;
#pragma unroll
                for (int e = 0; e < VECO; e += 4) {
                    const float4 q = *(const float4*)(src + e);
                    const float vq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float rr = sizeof(T) == 4 ? rv.f[(e + k) % 4] : bf16_to_f32(rv.h[(e + k) % 8]);
                        const float val = epi_act_h(vq[k], relu_pre, has_res, has_res ? rr : 0.f, relu_post);
                        if (sizeof(T) == 4) ov.f[(e + k) % 4] = val;
                        else ov.h[(e + k) % 8] = f32_to_bf16(val);
                    }
                }
                *(uint4*)((T*)a.y + off) = ov.u;
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
                Pack rv[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it)      // all residual loads first: independent HBM round trips
                    rv[it].u = *(const uint4*)((const T*)a.res + row_pix(64 * wave + lane / LPR + it * RPP) * a.ldc + cq);
#pragma unroll
                for (int it = 0; it < NIT; ++it) row(it, rv[it].u);
            }
        }
    } else {
        for (int idx = lane; idx < 64 * CP; idx += 64) {
            const int r = idx / CP, cc = idx - r * CP;
            if (cc >= a.Cout) continue;
            const size_t off = row_pix(64 * wave + r) * a.ldc + cc;
            const float rr = has_res ? elt<T>::ld((const T*)a.res + off) : 0.f;
            elt<T>::st((T*)a.y + off, epi_act_h(ep[r * C::EP_LD + cc], relu_pre, has_res, rr, relu_post));
        }
    }
}

template <typename T, int KS, int CIN, int CP, int TD, int TH, int TW, int TPC, int NBUF>
int launch_halo(const HaloArgs& a, hipStream_t s) {
    typedef HaloCfg<T, KS, CIN, CP, TD, TH, TW, TPC, NBUF> C;
    static_assert(C::LDS_BYTES <= 160 * 1024, "halo tile does not fit LDS");
    auto kern = conv3d_halo_kernel<T, KS, CIN, CP, TD, TH, TW, TPC, NBUF>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const long long nblk = (long long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, s, a);
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
    if (c.flags & (LT_EPI_STORE_F32 | LT_EPI_SIGMOID)) return 0;
    int ks = 0;
    if (p0.ntaps == 27 && c.pd == 1 && c.ph == 1 && c.pw == 1) ks = 3;
    else if (p0.ntaps == 343 && c.pd == 3 && c.ph == 3 && c.pw == 3) ks = 7;
    else return 0;
    if (c.D % 4 || c.H % 8 || c.W % 8) return 0;
    const long long nblk = (long long)c.N * (c.D / 4) * (c.H / 8) * (c.W / 8);
    if (nblk < 256 && !forced) return 0;   // too few workgroups: the 64x64 implicit-GEMM tile fills the chip better
    HaloArgs a;
    a.x = c.x; a.w = p0.w; a.y = c.y; a.res = c.res; a.bias = c.bias; a.scale = c.scale; a.shift = c.shift;
    a.N = c.N; a.D = c.D; a.H = c.H; a.W = c.W; a.Cout = c.Cout; a.ldc = c.ldc; a.k_pad = c.k_pad; a.flags = c.flags;
    a.tiles_d = c.D / 4; a.tiles_h = c.H / 8; a.tiles_w = c.W / 8;
    a.xcd_pin = (c.N % 8 == 0) ? 1 : 0;
    const bool bf = dtype == LT_BF16;
#define HALO_CASE(T_, KS_, CIN_, CP_, TPC_, NBUF_)                                               \
    if (ks == KS_ && c.Cin == CIN_ && cout_pad == CP_) {                                        \
        int rc = launch_halo<T_, KS_, CIN_, CP_, 4, 8, 8, TPC_, NBUF_>(a, s);                   \
        return rc == LT_OK ? 1 : rc;                                                            \
    }
    static const bool row_chunks = getenv("LT_HALO_ROW") != nullptr;   // A/B: 3-tap weight chunks -> 51 KB of LDS -> 3 workgroups per CU
    if (bf) {
        if (row_chunks) { HALO_CASE(bf16_t, 3, 32, 32, 3, 2) }
        HALO_CASE(bf16_t, 3, 32, 32, 9, 2)
        HALO_CASE(bf16_t, 3, 16, 32, 9, 2)
        HALO_CASE(bf16_t, 3, 64, 64, 3, 2)
        HALO_CASE(bf16_t, 3, 32, 64, 9, 2)
        HALO_CASE(bf16_t, 7, 32, 16, 7, 4)
    } else {
        HALO_CASE(float, 3, 32, 32, 3, 3)
        HALO_CASE(float, 3, 16, 32, 9, 2)
    }
#undef HALO_CASE
    return 0;
}

}  // namespace lt
