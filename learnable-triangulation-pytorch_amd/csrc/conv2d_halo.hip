// conv2d_halo_kernel: the 3x3 / stride 1 / pad 1 convolution 256 -> 256 of ResNet layer3's bottlenecks (reference mvn/models/pose_resnet.py:75-95, conv2 of
// the 35 stride-1 blocks of ResNet-152's third stage (the first block strides its 3x3): 20 % of the forward) with the input HALO of a tile resident in LDS -- the 2D sibling of
// conv3d_halo_wreg_kernel.
//
// Why (round 5): the implicit-GEMM kernel (conv_igemm7<2>) streams the im2col matrix through an LDS ring, i.e. every input pixel crosses L2 -> LDS nine
// times (540 MB from the fabric per launch for a 75 MB tensor: profiles/r05_hbm_traffic_pmc.json) behind 72 barriers per tile, and runs at 0.40 MFMA-busy.
// Here a workgroup owns TH rows x 24 columns of one image (24 x 24 maps: the 384 x 384 crops of the benchmark), DMAs the (TH + 2) x 26 pixel halo ONCE
// (512 bytes per pixel, 16-byte slots XOR-swizzled by (column & 7) | (row & 1) << 3 on the source side: every ds_read_b128 of a 4 x 8 pixel fragment is
// bank-conflict free for every tap), and then walks 9 taps x 16 K blocks with no barrier: per unit one weight fragment from global memory (fragment order
// of the transposed product, lt_conv_pack_weights_t32; wave = one 32-channel output block) and one pixel fragment per 4 x 8 block from LDS at the tap's
// offset (the taps come from the descriptor's tap table, read once per launch: one address register per tap and row group).  Transposed product with permuted
// weight rows: a lane ends up with two runs of 8 consecutive channels of ONE pixel and stores them as 16-byte vectors straight from the accumulators
// (BatchNorm + ReLU folded in; these layers have no residual).
// The same kernel runs the 4x4 / stride-2 / pad-1 transposed convolutions 256 -> 256 of the deconvolution head (pose_resnet.py:208-233): their four output
// parities are four phases of 2 x 2 taps over the same halo -- one launch, the halo loaded once, the weight ring running through the phase boundaries.
// Tile heights: 8 rows (one workgroup per CU) for throughput, 4 rows (two per CU) below half a round of 8-row tiles and for a ragged last round.
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"

using namespace lt;

namespace {

__device__ uint4 g_zero_page_2d[4];

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16p(const void* src, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
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
__device__ __forceinline__ void static_for_p(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for_p<I0 + 1, I1>(f);
    }
}

struct Halo2dPhase {
    const bf16_t* wfrag;     // lt_conv_pack_weights_t32 of this phase's [256][k_pad] weights (k = tap * 256 + ci)
    const int4* taps;        // lt_conv_phase.taps (device): (dd, dh, dw, element offset) per tap; input pixel of tap t for output pixel (h, w) = (h - pad_h + dh, w - pad_w + dw)
    int ooh, oow;            // output offset of the phase (a stride-2 transposed convolution: the output parity)
};

struct Halo2dArgs {
    const bf16_t* x;
    bf16_t* y;
    const float* bias;
    const float* scale;
    const float* shift;
    int N, H, W, OH, OW, osh, osw, ldc, flags, tiles_h, tiles_w, pad_h, pad_w;
    int tile0;               // first tile of this launch (a ragged last round runs as a second launch of half-height tiles)
    Halo2dPhase ph[4];
};

// TH = 8: six 4 x 8 pixel fragments per wave, 130 KB of halo, one workgroup per CU; TH = 4: three fragments, 78 KB, two workgroups per CU.
// NT taps per phase, NPH phases over the SAME halo: (9, 1) = the 3x3 convolution; (4, 4) = a 4x4 / stride-2 / pad-1 transposed convolution (the deconvolution
// head, pose_resnet.py:208-233: four output parities of 2 x 2 taps each -- the halo is loaded once for all four, where the implicit GEMM ran four launches).
template <int TH, int NT, int NPH>
__global__ __launch_bounds__(512, (TH == 8 ? 1 : 2)) void conv2d_halo_kernel(const Halo2dArgs a) {
    typedef bf16_t T;
    constexpr int CIN = 256, CP = 256, TW = 24, G = CIN / 16, NB = CP / 32, PW = TW + 2, HH = TH + 2, PXB = CIN * 2;
    constexpr int RG = TH / 4, CG = TW / 8, FR = RG * CG;
    constexpr int HV = HH * PW, NI = HV * PXB / 1024;
    static_assert(HV % 2 == 0 && PXB == 512 && NB == 8, "two pixels per DMA piece; eight waves = eight output blocks");
    auto swz = [](int hh_, int hw_) -> int { return (hw_ & 7) | ((hh_ & 1) << 3); };

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned long long zp_bits = (unsigned long long)(size_t)g_zero_page_2d;
    asm volatile("" : "+s"(zp_bits));
    const void* const zero_page = (const void*)(size_t)zp_bits;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int lin = blockIdx.x;
    {   // XCD b % 8 walks one contiguous run of tiles: the tiles of an image (they share halo rows) meet in one L2
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = lin & 7, j = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    lin += a.tile0;
    const int tpi = a.tiles_h * a.tiles_w;
    const int n = lin / tpi, rem = lin - n * tpi;
    const int h0 = (rem / a.tiles_w) * TH, w0 = (rem % a.tiles_w) * TW;
    const T* __restrict__ x = a.x + (size_t)n * a.H * a.W * CIN;

    // ---- halo DMA: piece i = halo pixels 2 i, 2 i + 1; lane l -> pixel 2 i + l / 32, physical slot l % 32 holds logical slot (l % 32) ^ swz ----
    for (int i = wave; i < NI; i += 8) {
        const int hv = 2 * i + (lane >> 5), ps = lane & 31;
        const int hh_ = hv / PW, hw_ = hv - hh_ * PW;
        const int ih = h0 - 1 + hh_, iw = w0 - 1 + hw_;
        const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        const void* src = ok ? (const void*)(x + ((size_t)ih * a.W + iw) * CIN + (ps ^ swz(hh_, hw_)) * 8) : zero_page;
        dma16p(src, lds0 + i * 1024);
    }

    // ---- roles: wave = output-channel block; fragment f = (row group i = f / CG, column group j = f % CG): pixel (4 i + vl / 8, 8 j + vl % 8) ----
    const int cb = wave;
    const int vl = lane & 31, hk = lane >> 5;
#ifndef LT_H2D_WD
#define LT_H2D_WD 7
#endif
    constexpr int NU = NT * G, WD = LT_H2D_WD, NS = WD + 1;   // weight fragments: requested WD units ahead, straight through the phases (NU % NS == 0)
    static_assert(NU % NS == 0, "the fragment ring must close at a phase boundary");
    auto wptr = [&](int p) -> const T* { return a.ph[p].wfrag + ((size_t)cb * 64 + lane) * 8; };
    auto load_w = [&](int p, int u) -> V16 {             // unit u = tap * G + g of phase p -> fragment ((u * NB + cb) * 64 + lane) * 16 bytes
        V16 v;
        v.u = *(const uint4*)(wptr(p) + (size_t)u * NB * 64 * 8);
        return v;
    };
    V16 wf[NS];
#pragma unroll
    for (int u = 0; u < WD; ++u) wf[u] = load_w(0, u);

    // epilogue constants: with several phases they are fetched once, in front of the tap loops (32 registers); the single-phase kernel fetches them behind
    // its tap loop instead (it has the registers of one more weight fragment in flight there)
    float esc[16], esf[16];
    auto load_consts = [&]() {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * cb + 16 * (e >> 3) + 8 * hk + (e & 7);
            const float bi = a.bias ? a.bias[c] : 0.f, sc = a.scale ? a.scale[c] : 1.f, sf = a.shift ? a.shift[c] : 0.f;
            esc[e] = sc; esf[e] = bi * sc + sf;
        }
    };
    if constexpr (NPH > 1) load_consts();
    const EpiFloors fl = epi_floors(a.flags);
    // every phase's taps, fetched ONCE and in front of the barrier, as wave-uniform halo offsets (row * PW + column of the tap for output pixel (0, 0)): inside
    // the phase loop they were vector loads with a vmcnt(0) each, queued behind the previous phase's stores (one exposed round trip per tap and phase)
    int tapo[NPH][NT];
    {
        int4 tv[NPH][NT];
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) tv[p][tp] = a.ph[p].taps[tp];        // all requests first: one wait, not one round trip per tap
        bool bad = false;
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                const int dh = tv[p][tp].y - a.pad_h, dw = tv[p][tp].z - a.pad_w;
                bad |= (tv[p][tp].x != 0) | (dh < -1) | (dh > 1) | (dw < -1) | (dw > 1);
                tapo[p][tp] = __builtin_amdgcn_readfirstlane((1 + dh) * 32 + (1 + dw));   // row offset * 32 + column offset (decoded below)
            }
        if (bad) __builtin_trap();                        // a tap outside the one-pixel halo: not this kernel's problem class (fail loudly)
    }

    // the halo pieces are the oldest vector-memory operations of this wave: wait for everything once, the barrier publishes the image
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    static_for_p<0, NPH>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        unsigned lp[NT][RG];                              // tap t, row group i: + j * 8 pixels as an immediate, K block g as XOR (g << 5)
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            const int th = 4 * i + (vl >> 3), tw = vl & 7;
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                const int row = th + (tapo[p][tp] >> 5), col = tw + (tapo[p][tp] & 31);
                lp[tp][i] = lds0 + (row * PW + col) * PXB + ((hk ^ swz(row, col)) << 4);
            }
        }
        f32x16 acc[FR];
#pragma unroll
        for (int f = 0; f < FR; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
#ifndef LT_H2D_XLOOK
#define LT_H2D_XLOOK 1
#endif
        constexpr int XL = LT_H2D_XLOOK, XS = XL + 1;       // pixel fragments are read XL units ahead of the MFMAs that use them (-DLT_H2D_XLOOK=n: A/B builds)
        V16 xa[XS][FR];
        auto load_x = [&](auto uc, V16 (&dst)[FR]) {
            constexpr int u = decltype(uc)::value;
            constexpr int tap = u / G, g = u % G;
            static_for_p<0, FR>([&](auto fc) {
                constexpr int f = decltype(fc)::value, i = f / CG, j = f % CG;
                dst[f].u = *(const uint4*)((lptr_t)(size_t)((lp[tap][i] ^ (g << 5)) + j * 8 * PXB));
            });
        };
        static_for_p<0, XL>([&](auto uc) { load_x(uc, xa[decltype(uc)::value % XS]); });
        static_for_p<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (u + WD < NU) wf[(u + WD) % NS] = load_w(p, u + WD);
            else if constexpr (p + 1 < NPH) wf[(u + WD) % NS] = load_w(p + 1, u + WD - NU);    // the next phase's first fragments, under this phase's tail
            if constexpr (u + XL < NU) load_x(std::integral_constant<int, u + XL>{}, xa[(u + XL) % XS]);
#pragma unroll
            for (int f = 0; f < FR; ++f)
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u % NS].h, xa[u % XS][f].h, acc[f], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });

        // ---- epilogue from the accumulators: lane (pixel, h) holds channels 32 cb + 8 h + e (e < 8) and 32 cb + 16 + 8 h + (e - 8) ----
        if constexpr (NPH == 1) load_consts();
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const int th = 4 * (f / CG) + (vl >> 3), tw = 8 * (f % CG) + (vl & 7);
            bf16_t* yo = a.y + (((size_t)n * a.OH + (h0 + th) * a.osh + a.ph[p].ooh) * a.OW + (w0 + tw) * a.osw + a.ph[p].oow) * a.ldc + 32 * cb + 8 * hk;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned o[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int e = 8 * q + 2 * d;
                    o[d] = pack_bf16x2(epi_apply(fmaf(acc[f][e], esc[e], esf[e]), fl, 0.f), epi_apply(fmaf(acc[f][e + 1], esc[e + 1], esf[e + 1]), fl, 0.f));
                }
                *(uint4*)(yo + 16 * q) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    });
}

// tiles [first, first + count) of the TH-row tiling of the whole tensor (count < 0: all of them)
template <int TH, int NT, int NPH>
int launch_halo2d(const Halo2dArgs& a0, hipStream_t s, long long first = 0, long long count = -1) {
    Halo2dArgs a = a0;
    a.tiles_h = a.H / TH;
    a.tiles_w = a.W / 24;
    const long long total = (long long)a.N * a.tiles_h * a.tiles_w;
    if (count < 0) count = total - first;
    if (count <= 0) return LT_OK;
    a.tile0 = (int)first;
    constexpr int lds = (TH + 2) * 26 * 512;
    static_assert(lds <= 160 * 1024, "halo fits LDS");
    auto kern = conv2d_halo_kernel<TH, NT, NPH>;
    LT_OPT_IN_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(512), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(2D halo)");
    return LT_OK;
}

}  // namespace

namespace lt {

// 1 = launched, 0 = not applicable (fall back), < 0 = error.  Takes: bf16, 256 -> 256 dense channels, maps whose width is a multiple of 24 and whose
// height is a multiple of 8, iteration space == input grid, every tap within one pixel of the output pixel, no residual, plain bf16 store, weights of every
// phase in the fragment order of the transposed product (weight_frag_layout 2); one phase of nine taps (3x3 / stride 1 / pad 1) or four phases of four
// taps with output stride 2 (4x4 / stride 2 / pad 1 transposed).
int conv2d_halo_try(int dtype, const ConvArgs& c, int cout_pad, int nphase, hipStream_t s) {
    if (dtype != LT_BF16 || (nphase != 1 && nphase != 4) || c.D != 1 || c.Do != 1 || c.OD != 1) return 0;
    if (c.sh != 1 || c.sw != 1 || c.H != c.Ho || c.W != c.Wo || c.W % 24 || c.H % 8) return 0;
    if (c.Cin != 256 || cout_pad != 256 || c.Cout != 256 || c.ldc % 8 || c.res || c.skip_x || c.x2) return 0;
    if (c.flags & (LT_EPI_STORE_F32 | LT_EPI_SIGMOID)) return 0;
    const int nt = nphase == 1 ? 9 : 4;
    // the taps live in device memory: their range is checked by the kernel (it traps); here the geometry that implies it -- a "same" 3x3 (pad 1) or the
    // 2 x 2-tap parities of a 4x4 / stride-2 / pad-1 transposed convolution (recorded with pad 0 and signed tap offsets)
    if (c.pd != 0) return 0;
    if (nphase == 1 && (c.osh != 1 || c.osw != 1 || c.OH != c.Ho || c.OW != c.Wo || c.ph != 1 || c.pw != 1)) return 0;
    if (nphase == 4 && (c.osh != 2 || c.osw != 2 || c.OH != 2 * c.Ho || c.OW != 2 * c.Wo || c.ph != 0 || c.pw != 0)) return 0;
    Halo2dArgs a;
    for (int p = 0; p < nphase; ++p) {
        const PhaseArg& ph = c.phase[p];
        if (!ph.wfrag_t || ph.ntaps != nt || ph.ood) return 0;
        a.ph[p].wfrag = (const bf16_t*)ph.wfrag_t; a.ph[p].taps = ph.taps; a.ph[p].ooh = ph.ooh; a.ph[p].oow = ph.oow;
    }
    a.x = (const bf16_t*)c.x; a.y = (bf16_t*)c.y;
    a.bias = c.bias; a.scale = c.scale; a.shift = c.shift;
    a.N = c.N; a.H = c.H; a.W = c.W; a.OH = c.OH; a.OW = c.OW; a.osh = c.osh; a.osw = c.osw; a.ldc = c.ldc; a.flags = c.flags; a.tiles_h = a.tiles_w = 0;
    a.pad_h = c.ph; a.pad_w = c.pw; a.tile0 = 0;
    // 8-row tiles (one workgroup per CU, every weight fragment feeds six MFMAs) from half a round of them on; below that the 4-row tiles (twice as many
    // workgroups) halve the time of the single round -- measured end to end (forward, samples/s, 8-row -> 4-row): 5 samples (60 tiles of 8 rows) 808 -> 871,
    // 10 samples (120) 1167 -> 1219, 16 samples (192) 1325 -> 1307, 64 samples (768) 1418 -> 1399 (LT_H2D_TH=4 / 8 forces one)
    const char* th = getenv("LT_H2D_TH");                       // A/B switches are read per launch CALL (tests flip them in-process); graph replays never get here
    const long long tiles8 = (long long)c.N * (c.H / 8) * (c.W / 24);
    const int n_cu = lt::device_cu_count8();
    const bool th4 = th ? th[0] == '4' : tiles8 <= n_cu / 2;   // half a round of 8-row tiles (128 on the 256-CU part)
    int rc;
    if (nphase != 1) rc = launch_halo2d<8, 4, 4>(a, s);
    else if (th4) rc = launch_halo2d<4, 9, 1>(a, s);
    else {
        // whole rounds of 8-row tiles (one workgroup per CU), and a ragged last round of at most half the CUs as 4-row tiles in a second launch: its
        // workgroups are half as long, so the tail costs half a round instead of a whole one (128 images = 384 tiles: 256 + 2 x 128)
        const long long rem = tiles8 % n_cu;
        // (the split point must be a whole row of tiles, so that 8-row tile t and the 4-row tiles 2 t, 2 t + 1 cover the same pixels)
        const bool split = !th && tiles8 > n_cu && rem > 0 && 2 * rem <= n_cu && (tiles8 - rem) % (c.W / 24) == 0 && !getenv("LT_H2D_NO_TAIL4");
        if (split) {
            rc = launch_halo2d<8, 9, 1>(a, s, 0, tiles8 - rem);
            if (rc == LT_OK) rc = launch_halo2d<4, 9, 1>(a, s, 2 * (tiles8 - rem), 2 * rem);
        } else rc = launch_halo2d<8, 9, 1>(a, s);
    }
    return rc == LT_OK ? 1 : rc;
}

}  // namespace lt
