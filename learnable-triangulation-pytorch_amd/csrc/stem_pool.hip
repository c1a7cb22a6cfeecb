// lt_stem_pool_fwd: the stem of the 2D backbone in one pass (bf16).
//
// PoseResNet.forward starts with conv1 (7x7, stride 2, pad 3, 3 -> 64) -> bn1 -> relu -> maxpool (3x3, stride 2, pad 1)
// (reference mvn/models/pose_resnet.py:293-297).  As two launches the 64-channel map at half resolution (604 MB at 128 images
// of 384x384) is written by the convolution and read back by the pool, and the implicit GEMM gathers its A operand in 16-byte
// pieces (Cin = 3 padded to 8 = one vector per tap): measured 0.56 + 0.19 ms.  Here a workgroup owns a 4 x 16 tile of POOLED
// pixels:
//   * the 23 x 72-pixel input patch under it is copied to LDS once (rows are contiguous 16-byte pixels -- or, x_layout 1, read
//     straight from the reference's fp32 (N, 3, H, W) images and rounded here: no separate layout pass), split by column
//     parity so that the stride-2 walk of the convolution reads consecutive 16-byte slots (no bank conflicts);
//   * the 9 x 33 convolution outputs the pool needs are 10 MFMA fragments of 32 pixels; the kernel window is padded to 7 x 8
//     taps (the 8th column has zero weights) so that one 32x32x16 MFMA consumes the tap pair (kh, 2kp), (kh, 2kp+1): lanes
//     0-31 hold the 8 channels of the even tap, lanes 32-63 of the odd one, and every patch offset is a compile-time immediate;
//   * the product is transposed (weights first) with the weight rows permuted as in conv3d_halo_col_kernel: a lane ends up with
//     two 16-byte channel runs of its own pixel, which go through affine + ReLU into an LDS tile [pixel][64 channels];
//   * the 3x3/2 max pool reads that tile (post-ReLU bf16 values are non-negative, so their bit patterns order like unsigned
//     integers: v_pk_max_u16; pixels outside the map hold 0 = the pool's -inf padding for non-negative data) and stores 16 bytes
//     per lane.
// Weights come straight from global memory, pre-packed in FRAGMENT order by lt_stem_pack_weights ([channel block][K step][lane]
// x 16 bytes, 56 KB, L2 resident): one fully coalesced 1 KB load per wave and K step (reading the lt_conv_fwd packing directly
// cost ~65 issue cycles per step, 64 scattered 16-byte rows, paid by the wave that issues the MFMAs: 0.48 -> 0.35 ms with the
// other fixes).  No LDS for them, so three workgroups fit a CU and their phases (patch copy / MFMA / pool) overlap.  rounding: the separate launches
// round the convolution output to bf16 before pooling; max commutes with that monotonic rounding, so the results are identical.
#include "conv_common.h"

using namespace lt;

namespace {

constexpr int SP_PH = 4, SP_PW = 16;                 // pooled tile
constexpr int SP_CH = 2 * SP_PH + 1, SP_CW = 2 * SP_PW + 1;   // convolution outputs under it: 9 x 33
constexpr int SP_NPIX = SP_CH * SP_CW;               // 297
constexpr int SP_NFRAG = (SP_NPIX + 31) / 32;        // 10 fragments of 32 pixels
constexpr int SP_IH = 2 * (SP_CH - 1) + 7;           // 23 input rows
constexpr int SP_IW = 2 * (SP_CW - 1) + 8;           // 72 input columns (7 taps + the zero-weight 8th)
constexpr int SP_HALF = SP_IW / 2;                   // slots per parity plane of a patch row
constexpr int SP_PATCH_B = SP_IH * SP_IW * 16;
constexpr int SP_OLD = 144;                          // out-tile pixel stride in bytes: 128 + 16, conflict-free 16-byte writes
constexpr int SP_OUT_B = SP_NPIX * SP_OLD;
constexpr int SP_LDS = SP_PATCH_B > SP_OUT_B ? SP_PATCH_B : SP_OUT_B;
static_assert(SP_NFRAG == 10, "two M halves of five fragments");

struct StemArgs {
    const void* x;         // N, H, W, 8 bf16 -- or N, 3, H, W fp32 (the reference's image tensor, converted in the patch copy)
    const bf16_t* w;       // fragment-packed: [2 channel blocks][28 K steps][64 lanes][8]
    bf16_t* y;             // N, Hp, Wp, 64
    const float* bias;
    const float* scale;
    const float* shift;
    int N, H, W, Hc, Wc, Hp, Wp;
    int tiles_y, tiles_x;
};

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

template <bool NCHW>
__global__ __launch_bounds__(256, 3) void stem_pool_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int py0 = ty * SP_PH, px0 = tx * SP_PW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;          // first convolution pixel of the tile (may be -1)
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // first input pixel of the patch

    // ---- input patch -> LDS: slot (row, parity, col / 2).  All of a thread's loads are issued before the first LDS write (as a
    // rolled loop every iteration waited out its own HBM round trip); no branches: out-of-image pixels read pixel 0 and are
    // cleared, the surplus items of the last round redo the last slot ----
    {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int NIT = (SP_IH * SP_IW + 255) / 256;
        u32x4 v[NIT];
        int slot[NIT];
        if (NCHW) {
            // fp32 planes of the reference's (N, 3, H, W) images: three coalesced 4-byte loads per pixel, rounded to bf16
            // (nearest even, exactly what lt_nchw_to_nhwc + the 8-channel path do) into channels 0-2 of the pixel's slot
            const float* xs = (const float*)a.x + (size_t)n * 3 * a.H * a.W;
            const size_t plane = (size_t)a.H * a.W;
            float f[NIT][3];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                int i = t + 256 * k;
                i = i < SP_IH * SP_IW ? i : SP_IH * SP_IW - 1;
                const int r = i / SP_IW, c = i - r * SP_IW;
                const int gy = iy0 + r, gx = ix0 + c;
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const size_t off = ok ? (size_t)gy * a.W + gx : 0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float ld = xs[ch * plane + off];
                    f[k][ch] = ok ? ld : 0.f;
                }
                slot[k] = ((r * 2 + (c & 1)) * SP_HALF + (c >> 1)) * 16;
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                v[k] = (u32x4)(0u);
                v[k][0] = pack_bf16x2(f[k][0], f[k][1]);
                v[k][1] = pack_bf16x2(f[k][2], 0.f);
            }
        } else {
            const bf16_t* xs = (const bf16_t*)a.x + (size_t)n * a.H * a.W * 8;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                int i = t + 256 * k;
                i = i < SP_IH * SP_IW ? i : SP_IH * SP_IW - 1;
                const int r = i / SP_IW, c = i - r * SP_IW;
                const int gy = iy0 + r, gx = ix0 + c;
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const u32x4 ld = *(const u32x4*)(xs + (ok ? ((size_t)gy * a.W + gx) * 8 : 0));
                v[k] = ok ? ld : (u32x4)(0u);
                slot[k] = ((r * 2 + (c & 1)) * SP_HALF + (c >> 1)) * 16;
            }
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) *(u32x4*)(smem + slot[k]) = v[k];
    }

    // ---- roles: wave = (M half mh, channel block nb); fragment f = 5 mh + i ----
    const int mh = wave >> 1, nb = wave & 1;
    const int pl = lane & 31, h = lane >> 5;
    unsigned abase[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int m = (5 * mh + i) * 32 + pl;
        m = m < SP_NPIX ? m : SP_NPIX - 1;               // padding rows of the last fragment recompute the last pixel
        const int cy = m / SP_CW, cx = m - cy * SP_CW;
        abase[i] = ((2 * cy * 2 + h) * SP_HALF + cx) * 16;   // row 2 cy, parity h, slot cx (+ kh rows, + kp slots as immediates)
    }
    // weight fragments of this lane: block nb, K step s -> 16 bytes at ((nb * 28 + s) * 64 + lane) * 16 (packed by stem_pack_kernel:
    // MFMA row r = lane & 31 carries channel chan(r), two 16-byte runs per lane as in conv3d_halo_col_kernel; the 8th tap is zero)
    const bf16_t* wrow = a.w + ((size_t)nb * 28 * 64 + lane) * 8;
    auto load_w = [&](int s) -> V16 {
        V16 v;
        v.u = *(const uint4*)(wrow + (size_t)s * 64 * 8);
        return v;
    };

    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    constexpr int WD = 4;                                // weight fragments requested WD K steps ahead (L2 latency)
    V16 wf[WD + 1];
#pragma unroll
    for (int s = 0; s < WD; ++s) wf[s] = load_w(s);
    __syncthreads();                                     // patch complete

    V16 fa[2][5];
#pragma unroll
    for (int i = 0; i < 5; ++i) fa[0][i].u = *(const uint4*)(smem + abase[i]);
#pragma unroll
    for (int s = 0; s < 28; ++s) {
        if (s + WD < 28) wf[(s + WD) % (WD + 1)] = load_w(s + WD);
        if (s + 1 < 28) {
            const int kh1 = (s + 1) >> 2, kp1 = (s + 1) & 3;
#pragma unroll
            for (int i = 0; i < 5; ++i) fa[(s + 1) & 1][i].u = *(const uint4*)(smem + abase[i] + (kh1 * 2 * SP_HALF + kp1) * 16);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s % (WD + 1)].h, fa[s & 1][i].h, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);               // keep the weight load WD steps ahead of its use (the scheduler sinks it to 1)
    }
    __syncthreads();                                     // every wave is done with the patch: the out tile takes its place

    // ---- affine + ReLU -> out tile [pixel][64 channels] (bf16), zero outside the map ----
    {
        float esc[16], esf[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * nb + 16 * (e >> 3) + 8 * h + (e & 7);
            const float bi = a.bias ? a.bias[c] : 0.f, sc = a.scale ? a.scale[c] : 1.f, sf = a.shift ? a.shift[c] : 0.f;
            esc[e] = sc; esf[e] = bi * sc + sf;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int m = (5 * mh + i) * 32 + pl;
            if (m < SP_NPIX) {
                const int cy = m / SP_CW, cx = m - cy * SP_CW;
                const bool ok = (unsigned)(cy0 + cy) < (unsigned)a.Hc && (unsigned)(cx0 + cx) < (unsigned)a.Wc;
                unsigned o[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const float v0 = ok ? fmaxf(fmaf(acc[i][2 * d], esc[2 * d], esf[2 * d]), 0.f) : 0.f;
                    const float v1 = ok ? fmaxf(fmaf(acc[i][2 * d + 1], esc[2 * d + 1], esf[2 * d + 1]), 0.f) : 0.f;
                    o[d] = pack_bf16x2(v0, v1);
                }
                unsigned char* dst = smem + m * SP_OLD + (32 * nb + 8 * h) * 2;
                *(uint4*)dst = make_uint4(o[0], o[1], o[2], o[3]);
                *(uint4*)(dst + 32) = make_uint4(o[4], o[5], o[6], o[7]);
            }
        }
    }
    __syncthreads();

    // ---- 3x3 / 2 max pool out of the tile: item = (pooled pixel, 8-channel chunk) ----
    bf16_t* ys = a.y + (size_t)n * a.Hp * a.Wp * 64;
    for (int it = t; it < SP_PH * SP_PW * 8; it += 256) {
        const int ch = it & 7, pp = it >> 3;
        const int lx = pp % SP_PW, ly = pp / SP_PW;
        const int py = py0 + ly, px = px0 + lx;
        if (py < a.Hp && px < a.Wp) {
            const unsigned char* src = smem + ((2 * ly) * SP_CW + 2 * lx) * SP_OLD + ch * 16;
            u16x8 m = *(const u16x8*)src;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    if (dy || dx) m = __builtin_elementwise_max(m, *(const u16x8*)(src + (dy * SP_CW + dx) * SP_OLD));
            *(u16x8*)(ys + ((size_t)py * a.Wp + px) * 64 + ch * 8) = m;
        }
    }
}

// lt_conv_fwd packing [64][k_pad] (k = (kh*7 + kw)*8 + ci) -> fragment order; one thread per 16-byte piece
__global__ void stem_pack_kernel(const bf16_t* __restrict__ w, int k_pad, bf16_t* __restrict__ out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;   // (nb * 28 + s) * 64 + lane
    if (g >= 2 * 28 * 64) return;
    const int lane = g & 63, s = (g >> 6) % 28, nb = g / (28 * 64);
    const int pl = lane & 31, h = lane >> 5, kh = s >> 2, kw = 2 * (s & 3) + h;
    const int chan = 32 * nb + 16 * (pl >> 4) + 8 * ((pl >> 2) & 1) + 4 * ((pl >> 3) & 1) + (pl & 3);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (kw < 7) v = *(const uint4*)(w + (size_t)chan * k_pad + (kh * 7 + kw) * 8);
    *(uint4*)(out + (size_t)g * 8) = v;
}

}  // namespace

extern "C" size_t lt_stem_packed_bytes(void) { return (size_t)2 * 28 * 64 * 16; }

extern "C" int lt_stem_pack_weights(const void* weight, int32_t k_pad, void* packed, void* stream) {
    LT_REQUIRE(weight && packed, LT_ERR_INVALID, "lt_stem_pack_weights: null argument");
    LT_REQUIRE(k_pad >= 49 * 8 && k_pad % 8 == 0, LT_ERR_INVALID, "lt_stem_pack_weights: k_pad %d (weights [64][k_pad >= 392], bf16)", k_pad);
    hipLaunchKernelGGL(stem_pack_kernel, dim3(2 * 28), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)weight, k_pad, (bf16_t*)packed);
    LT_CHECK_LAUNCH("lt_stem_pack_weights");
    return LT_OK;
}

extern "C" int lt_stem_pool_fwd(const lt_stem_desc* d, const void* x, void* y, void* stream) {
    LT_REQUIRE(d && x && y, LT_ERR_INVALID, "lt_stem_pool_fwd: null argument");
    LT_REQUIRE(d->dtype == LT_BF16, LT_ERR_UNSUPPORTED, "lt_stem_pool_fwd: bf16 only (fp32 plans record conv + max pool)");
    LT_REQUIRE(d->N >= 1 && d->H >= 1 && d->W >= 1, LT_ERR_INVALID, "lt_stem_pool_fwd: bad shape");
    LT_REQUIRE(d->x_layout == 0 || d->x_layout == 1, LT_ERR_INVALID, "lt_stem_pool_fwd: x_layout %d", d->x_layout);
    LT_REQUIRE(d->Cin == (d->x_layout ? 3 : 8) && d->Cout == 64, LT_ERR_UNSUPPORTED,
               "lt_stem_pool_fwd: %d -> %d channels (8 padded bf16 channels, or the 3 fp32 planes of x_layout 1, -> 64)", d->Cin, d->Cout);
    LT_REQUIRE(d->weight && ((uintptr_t)d->weight & 15) == 0, LT_ERR_INVALID, "lt_stem_pool_fwd: packed weights (lt_stem_pack_weights) missing or misaligned");
    StemArgs a;
    a.x = x; a.w = (const bf16_t*)d->weight; a.y = (bf16_t*)y;
    a.bias = d->bias; a.scale = d->scale; a.shift = d->shift;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.Hc = (d->H + 6 - 7) / 2 + 1; a.Wc = (d->W + 6 - 7) / 2 + 1;
    a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;
    a.tiles_y = (a.Hp + SP_PH - 1) / SP_PH; a.tiles_x = (a.Wp + SP_PW - 1) / SP_PW;
    const long long nblk = (long long)a.N * a.tiles_y * a.tiles_x;
    LT_REQUIRE(nblk < (1ll << 31), LT_ERR_UNSUPPORTED, "lt_stem_pool_fwd: too many tiles");
    if (d->x_layout) hipLaunchKernelGGL(stem_pool_kernel<true>, dim3((unsigned)nblk), dim3(256), SP_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stem_pool_kernel<false>, dim3((unsigned)nblk), dim3(256), SP_LDS, (hipStream_t)stream, a);
    LT_CHECK_LAUNCH("lt_stem_pool_fwd");
    return LT_OK;
}
