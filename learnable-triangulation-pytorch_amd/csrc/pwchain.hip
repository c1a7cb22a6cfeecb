// lt_pwchain_fwd: a chain of pointwise (1x1x1) convolutions evaluated per voxel in registers.
//
// V2VModel ends with Basic3DBlock(32,32,1) -> Basic3DBlock(32,32,1) -> Conv3d(32,J,1) (reference mvn/models/v2v.py:157-169).
// As three lt_conv_fwd launches each of them streams the whole 64^3 x 32-channel volume through HBM (read 268 MB + write
// 268 MB per launch at 16 samples; measured 0.27 + 0.27 + 0.35 ms); chained, the volume is read once and only the fp32
// logits are written (0.55 GB instead of 1.6 GB).
//
// Formulation: the TRANSPOSED product D[co][voxel] = W[co][ci] * X^T[ci][voxel] on the 32x32x16 bf16 MFMA.
//   * B operand (activations): lane (voxel = lane & 31, h = lane >> 5) holds X[voxel][16 kb + 8 h .. + 7] = one 16-byte
//     global load per K block, straight from the channels-last volume -- no LDS staging;
//   * A operand (weights): lane (co = lane & 31, h) holds W[co][16 kb + 8 h .. + 7], loaded once per wave;
//   * result: lane (voxel, h) holds channels c(e) = (e & 3) + 8 (e >> 2) + 4 h, e = 0..15, of ITS voxel.  After the affine
//     + ReLU those 16 values, rounded to bf16, are exactly the B operand of the next layer up to a permutation of the K
//     index (K slot s of block kb' <-> channel 16 kb' + 8 (s >> 2) + 4 h + (s & 3)); the contraction does not care about
//     the order as long as the weights use the same one, so layers >= 2 load their weights with that permutation and the
//     activations never leave the registers between layers;
//   * channels-last output: the fp32 logits of 64 voxels go through a wave-private LDS tile to become one contiguous
//     64*J*4-byte run of 16-byte stores (rows of J = 17 floats are not vector aligned on their own);
//   * planar output (desc.plane > 0, what the soft-argmax wants: one contiguous volume per joint): the same wave-private
//     tile, laid out [J][64 voxels]: every channel of the tile is a 256-byte run = 16 lanes x 16 bytes (storing the result
//     registers directly, 4 bytes per lane in 128-byte runs, measured 0.34 ms against 0.19 ms for the channels-last tile).
// Intermediate activations are rounded to bf16 exactly where the separate launches would store them.
#include "conv_common.h"

using namespace lt;

namespace {

struct PwArgs {
    const bf16_t* x;
    float* y;
    const bf16_t* w[LT_PWCHAIN_MAX];
    const float* bias[LT_PWCHAIN_MAX];
    const float* scale[LT_PWCHAIN_MAX];
    const float* shift[LT_PWCHAIN_MAX];
    int k_pad[LT_PWCHAIN_MAX];
    int relu[LT_PWCHAIN_MAX];
    long long ntile;   // tiles of 64 voxels
    long long plane;   // planar output: voxels per sample (a multiple of 64), else 0
    int cout_last;
};

template <int L, bool PLANAR>
__global__ __launch_bounds__(256) void pwchain_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vl = lane & 31, h = lane >> 5;
    const int J = a.cout_last;
    // epilogue constants live in LDS ([layer][bias|scale|shift][32]) and are re-read per tile: as registers they cost 48 VGPRs
    // per layer (260 in all for three layers = one wave per SIMD, and this kernel lives on loads in flight)
    float* cst = smem_f;
    float* ep = smem_f + L * 96 + wave * 64 * J;
    for (int i = threadIdx.x; i < L * 96; i += 256) {
        const int l = i / 96, k = (i / 32) % 3, c = i % 32;
        const float* src = k == 0 ? a.bias[l] : (k == 1 ? a.scale[l] : a.shift[l]);
        cst[i] = src ? src[c] : (k == 1 ? 1.f : 0.f);
    }
    __syncthreads();

    // ---- weights of every layer, once per wave ----
    V16 wf[L][2];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const bf16_t* wr = a.w[l] + (size_t)vl * a.k_pad[l];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (l == 0) {
                wf[l][kb].u = *(const uint4*)(wr + 16 * kb + 8 * h);
            } else {   // K slots in the order the previous layer's result registers hold the channels
                const uint2 lo = *(const uint2*)(wr + 16 * kb + 4 * h);
                const uint2 hi = *(const uint2*)(wr + 16 * kb + 8 + 4 * h);
                wf[l][kb].u = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }

    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    auto load_tile = [&](long long t, V16 (&xf)[2][2]) {
        const bf16_t* xr = a.x + ((size_t)t * 64 + vl) * 32 + 8 * h;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) xf[nt][kb].u = *(const uint4*)(xr + (size_t)nt * 32 * 32 + 16 * kb);
    };
    V16 xf[2][2], xn[2][2];
    if (gw < a.ntile) load_tile(gw, xf);
    for (long long t = gw; t < a.ntile; t += nw) {
        const bool more = t + nw < a.ntile;
        float* yp = nullptr;                             // planar: voxel 0 of tile t, channel 0
        if (PLANAR) {
            const long long smp = (t * 64) / a.plane;
            yp = a.y + smp * J * a.plane + (t * 64 - smp * a.plane);
        }
        if (more) load_tile(t + nw, xn);                 // next tile's 4 KB in flight under this tile's MFMAs
        f32x16 acc[2];
#pragma unroll
        for (int l = 0; l < L; ++l) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[l][kb].h, xf[nt][kb].h, acc[nt], 0, 0, 0);
            }
            const float floor_l = a.relu[l] ? 0.f : -__builtin_inff();   // max(v, -inf) == v
            // this lane's channels c(e) = (e & 3) + 8 (e >> 2) + 4 h: four runs of four consecutive channels
            float bi[16], sc[16], sf[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *(const float4*)(cst + l * 96 + 8 * g + 4 * h);
                const float4 s4 = *(const float4*)(cst + l * 96 + 32 + 8 * g + 4 * h);
                const float4 t4 = *(const float4*)(cst + l * 96 + 64 + 8 * g + 4 * h);
                bi[4 * g] = b4.x; bi[4 * g + 1] = b4.y; bi[4 * g + 2] = b4.z; bi[4 * g + 3] = b4.w;
                sc[4 * g] = s4.x; sc[4 * g + 1] = s4.y; sc[4 * g + 2] = s4.z; sc[4 * g + 3] = s4.w;
                sf[4 * g] = t4.x; sf[4 * g + 1] = t4.y; sf[4 * g + 2] = t4.z; sf[4 * g + 3] = t4.w;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float val[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    val[e] = fmaxf((acc[nt][e] + bi[e]) * sc[e] + sf[e], floor_l);
                }
                if (l + 1 < L) {                         // round to bf16: the next layer's B operand, in register order
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        unsigned u[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) u[i] = pack_bf16x2(val[8 * kb + 2 * i], val[8 * kb + 2 * i + 1]);
                        xf[nt][kb].u = make_uint4(u[0], u[1], u[2], u[3]);
                    }
                } else if (PLANAR) {                     // fp32 logits -> wave-private LDS tile [J][64 voxels]
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int c = (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (c < J) ep[c * 64 + 32 * nt + vl] = val[e];
                    }
                } else {                                 // fp32 logits -> wave-private LDS tile [64 voxels][J]
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int c = (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (c < J) ep[(32 * nt + vl) * J + c] = val[e];
                    }
                }
            }
        }
        // 64 * J floats = 16 J vectors of 16 bytes, contiguous in y (ldy == J, 64 voxels * J * 4 B is a multiple of 16)
        if (PLANAR) {                                    // channel c of the tile: 64 floats = lanes 16 c .. 16 c + 15 of the sweep
            for (int i = lane; i < 16 * J; i += 64)
                *(float4*)(yp + (long long)(i >> 4) * a.plane + 4 * (i & 15)) = *(const float4*)(ep + 4 * i);
        } else {
            float4* dst = (float4*)(a.y + (size_t)t * 64 * J);
            const float4* src = (const float4*)ep;
            for (int i = lane; i < 16 * J; i += 64) dst[i] = src[i];
        }
        if (more) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) xf[nt][kb] = xn[nt][kb];
        }
    }
}

template <int L, bool PLANAR>
int launch_pw(const PwArgs& a, hipStream_t s) {
    auto kern = pwchain_kernel<L, PLANAR>;
    const size_t lds = ((size_t)L * 96 + (size_t)4 * 64 * a.cout_last) * sizeof(float);
    long long blocks = (a.ntile + 3) / 4;
    if (blocks > 2048) blocks = 2048;                    // several tiles per wave: the weight/constant setup is amortised
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a);
    LT_CHECK_LAUNCH("lt_pwchain_fwd");
    return LT_OK;
}

}  // namespace

extern "C" int lt_pwchain_fwd(const lt_pwchain_desc* d, const void* x, void* y, void* stream) {
    LT_REQUIRE(d && x && y, LT_ERR_INVALID, "lt_pwchain_fwd: null argument");
    LT_REQUIRE(d->dtype == LT_BF16, LT_ERR_UNSUPPORTED, "lt_pwchain_fwd: bf16 activations only (fp32 runs the layers one by one)");
    LT_REQUIRE(d->nlayers >= 1 && d->nlayers <= LT_PWCHAIN_MAX, LT_ERR_INVALID, "lt_pwchain_fwd: nlayers %d", d->nlayers);
    LT_REQUIRE(d->cin == 32, LT_ERR_UNSUPPORTED, "lt_pwchain_fwd: input width %d (32 supported)", d->cin);
    LT_REQUIRE(d->rows > 0 && d->rows % 64 == 0, LT_ERR_UNSUPPORTED, "lt_pwchain_fwd: rows %lld not a multiple of 64", (long long)d->rows);
    PwArgs a;
    for (int l = 0; l < d->nlayers; ++l) {
        const bool last = l + 1 == d->nlayers;
        LT_REQUIRE(d->cout[l] >= 1 && d->cout[l] <= 32 && (last || d->cout[l] == 32), LT_ERR_UNSUPPORTED,
                   "lt_pwchain_fwd: layer %d width %d (inner layers must be 32 wide, the last <= 32)", l, d->cout[l]);
        LT_REQUIRE(d->k_pad[l] >= 32 && d->k_pad[l] % 8 == 0 && d->weight[l], LT_ERR_INVALID, "lt_pwchain_fwd: layer %d weights", l);
        const int extra = d->flags[l] & ~(LT_EPI_RELU_POST | LT_EPI_STORE_F32);
        LT_REQUIRE(extra == 0, LT_ERR_UNSUPPORTED, "lt_pwchain_fwd: layer %d flags 0x%x", l, d->flags[l]);
        LT_REQUIRE(((d->flags[l] & LT_EPI_STORE_F32) != 0) == last, LT_ERR_UNSUPPORTED,
                   "lt_pwchain_fwd: exactly the last layer stores fp32 (layer %d)", l);
        a.w[l] = (const bf16_t*)d->weight[l];
        a.bias[l] = d->bias[l]; a.scale[l] = d->scale[l]; a.shift[l] = d->shift[l];
        a.k_pad[l] = d->k_pad[l];
        a.relu[l] = (d->flags[l] & LT_EPI_RELU_POST) ? 1 : 0;
    }
    a.cout_last = d->cout[d->nlayers - 1];
    LT_REQUIRE(d->ldy == a.cout_last, LT_ERR_UNSUPPORTED, "lt_pwchain_fwd: ldy %d != width %d", d->ldy, a.cout_last);
    LT_REQUIRE(d->plane >= 0 && d->plane % 64 == 0 && (d->plane == 0 || d->rows % d->plane == 0), LT_ERR_INVALID,
               "lt_pwchain_fwd: plane %lld (planar output needs a multiple of 64 voxels per sample that divides rows)", (long long)d->plane);
    a.plane = d->plane;
    a.x = (const bf16_t*)x;
    a.y = (float*)y;
    a.ntile = d->rows / 64;
    hipStream_t s = (hipStream_t)stream;
    if (a.plane) {
        switch (d->nlayers) {
            case 1: return launch_pw<1, true>(a, s);
            case 2: return launch_pw<2, true>(a, s);
            default: return launch_pw<3, true>(a, s);
        }
    }
    switch (d->nlayers) {
        case 1: return launch_pw<1, false>(a, s);
        case 2: return launch_pw<2, false>(a, s);
        default: return launch_pw<3, false>(a, s);
    }
}
