// conv_pw: streaming kernel for the narrow single-tap layers of V2V (bf16): 1x1x1 convolutions (the Res3DBlock skip paths,
// reference mvn/models/v2v.py:20-46) and the 2x2x2 stride-2 transposed convolutions (Upsample3DBlock, v2v.py:60-75), whose
// eight output parities are eight single-tap phases over the same input voxels.
//
// These layers are pure HBM streams (16->32 at 64^3: read 268 MB, write 537 MB; deconv 64->32 at 32^3 -> 64^3: read 134 MB +
// 537 MB of skip connection, write 537 MB, at 32 samples) that the implicit GEMM ran at 1.9-2.1 TB/s: its K step is 64
// elements (a 16-channel voxel fills a quarter of a staged row, the rest comes from the zero page), every phase of the
// deconvolution was a separate pass over the input, and the tiles go through LDS twice.  Here, as in pwchain.hip:
//   * the product is transposed, D[co][voxel] = W[co][ci] X^T[ci][voxel]: activations are the B operand, 16 bytes per lane
//     and K block straight from the channels-last volume, read ONCE for all phases, the next tile's loads in flight under
//     this tile's arithmetic;
//   * the weights of all phases sit in LDS in MFMA fragment order (one conflict-free ds_read_b128 per fragment), with the rows
//     permuted so that a lane's 16 results are two runs of 8 consecutive channels (conv3d_halo_col_kernel): the epilogue
//     (affine, ReLU floors, residual) loads and stores 16 bytes per lane without any staging;
//   * a phase's output voxel is (o * out_stride + out_off) per axis, so the deconvolution's scatter is just an address.
#include "conv_common.h"

using namespace lt;

namespace {

struct PwPhase {
    const bf16_t* w;
    int ood, ooh, oow;
};

struct PwArgs {
    const bf16_t* x;
    bf16_t* y;
    const bf16_t* res;
    const float* bias;
    const float* scale;
    const float* shift;
    int Do, Ho, Wo, OD, OH, OW, osd, osh, osw;
    int ldc, k_pad, flags, nphase, plain;   // plain: output voxel index == input row (a 1x1x1 convolution)
    long long ntile;                        // tiles of 64 rows
    PwPhase phase[LT_CONV_MAX_PHASES];
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KB, int NB>
__global__ __launch_bounds__(256) void conv_pw_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vl = lane & 31, h = lane >> 5;
    constexpr int CIN = 16 * KB;

    // ---- weights of every phase -> LDS, fragment order: slot ((p * NB + nb) * KB + kb) * 64 + lane ----
    for (int g = threadIdx.x; g < a.nphase * NB * KB * 64; g += 256) {
        const int l = g & 63, kb = (g >> 6) % KB, nb = ((g >> 6) / KB) % NB, p = (g >> 6) / (KB * NB);
        const int r = l & 31, hh = l >> 5;
        const int chan = 32 * nb + 16 * (r >> 4) + 8 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1) + (r & 3);
        *(uint4*)(smem + (size_t)g * 16) = *(const uint4*)(a.phase[p].w + (size_t)chan * a.k_pad + 16 * kb + 8 * hh);
    }
    // this lane's 16 channels per block: 8 h + e (e < 8), 16 + 8 h + (e - 8); (acc + bias) * scale + shift as one fma
    float esc[NB][16], esf[NB][16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * nb + 16 * (e >> 3) + 8 * h + (e & 7);
            const float bi = a.bias ? a.bias[c] : 0.f, sc = a.scale ? a.scale[c] : 1.f, sf = a.shift ? a.shift[c] : 0.f;
            esc[nb][e] = sc; esf[nb][e] = bi * sc + sf;
        }
    const EpiFloors fl = epi_floors(a.flags);
    const bool has_res = a.res != nullptr;
    __syncthreads();

    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    auto load_tile = [&](long long t, V16 (&xf)[2][KB]) {
        const bf16_t* xr = a.x + ((size_t)t * 64 + vl) * CIN + 8 * h;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) xf[nt][kb].u = *(const uint4*)(xr + (size_t)nt * 32 * CIN + 16 * kb);
    };
    V16 xf[2][KB], xn[2][KB];
    if (gw < a.ntile) load_tile(gw, xf);
    for (long long t = gw; t < a.ntile; t += nw) {
        const bool more = t + nw < a.ntile;
        if (more) load_tile(t + nw, xn);
        // output voxel of this lane's row in each fragment, before the phase offset: element offset of (n, d * osd, h * osh, w * osw)
        size_t obase[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int m = (int)(t * 64) + 32 * nt + vl;       // rows < 2^31 (checked by lt_conv_fwd)
            if (a.plain) {
                obase[nt] = (size_t)m * a.ldc;
            } else {
                int r = m;
                const int ow = r % a.Wo; r /= a.Wo;
                const int oh = r % a.Ho; r /= a.Ho;
                const int od = r % a.Do;
                const int n = r / a.Do;
                obase[nt] = ((((size_t)n * a.OD + (size_t)od * a.osd) * a.OH + (size_t)oh * a.osh) * a.OW + (size_t)ow * a.osw) * a.ldc;
            }
        }
        for (int p = 0; p < a.nphase; ++p) {
            const size_t poff = (((size_t)a.phase[p].ood * a.OH + a.phase[p].ooh) * a.OW + a.phase[p].oow) * a.ldc + 8 * h;
            const unsigned char* wl = smem + ((size_t)p * NB * KB * 64 + lane) * 16;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                u32x4 rq[2][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int q = 0; q < 2; ++q) rq[nt][q] = (u32x4)(0u);
                if (has_res) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int q = 0; q < 2; ++q) rq[nt][q] = *(const u32x4*)(a.res + obase[nt] + poff + 32 * nb + 16 * q);
                }
                f32x16 acc[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    V16 wf;
                    wf.u = *(const uint4*)(wl + (size_t)(nb * KB + kb) * 64 * 16);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf.h, xf[nt][kb].h, acc[nt], 0, 0, 0);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    bf16_t* yo = a.y + obase[nt] + poff + 32 * nb;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        unsigned o[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const int e = 8 * q + 2 * d;
                            const unsigned rr = rq[nt][q][d];
                            // without a residual rr == 0: v + 0.0 (only the sign of a zero result can differ from v + -0.0)
                            const float v0 = epi_apply(fmaf(acc[nt][e], esc[nb][e], esf[nb][e]), fl, __uint_as_float(rr << 16));
                            const float v1 = epi_apply(fmaf(acc[nt][e + 1], esc[nb][e + 1], esf[nb][e + 1]), fl, __uint_as_float(rr & 0xffff0000u));
                            o[d] = pack_bf16x2(v0, v1);
                        }
                        *(uint4*)(yo + 16 * q) = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
        if (more) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) xf[nt][kb] = xn[nt][kb];
        }
    }
}

template <int KB, int NB>
int launch_pw(const PwArgs& a, hipStream_t s) {
    auto kern = conv_pw_kernel<KB, NB>;
    const size_t lds = (size_t)a.nphase * NB * KB * 64 * 16;
    LT_OPT_IN_LDS(kern, 64 * 1024);
    long long blocks = (a.ntile + 3) / 4;
    if (blocks > 1024) blocks = 1024;                    // several tiles per wave: the weight staging is amortised
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a);
    LT_CHECK_LAUNCH("lt_conv_fwd(pointwise stream)");
    return LT_OK;
}

}  // namespace

namespace lt {

// 1 = launched, 0 = not applicable (fall back), < 0 = error.  Takes: bf16, every phase a single tap, stride 1, no padding,
// phase grid == input grid (then the tap can only be the identity: any other would read outside the input), Cin in
// {16, 32, 64, 128}, Cout in {32, 64} dense in cout_pad, 16-byte aligned rows, M % 64 == 0, plain bf16 store.
int conv_pw_try(int dtype, const ConvArgs& c, int cout_pad, int nphase, hipStream_t s) {
    static const bool off = getenv("LT_CONV_NO_PW") != nullptr;   // A/B
    if (off || dtype != LT_BF16) return 0;
    if (c.sd != 1 || c.sh != 1 || c.sw != 1 || c.pd || c.ph || c.pw || c.D != c.Do || c.H != c.Ho || c.W != c.Wo) return 0;
    if (c.flags & (LT_EPI_STORE_F32 | LT_EPI_SIGMOID)) return 0;
    if (!(c.Cin == 16 || c.Cin == 32 || c.Cin == 64 || c.Cin == 128) || !(cout_pad == 32 || cout_pad == 64) || c.Cout != cout_pad) return 0;
    if (c.ldc % 8 || c.k_pad % 8 || c.M % 64) return 0;
    for (int p = 0; p < nphase; ++p)
        if (c.phase[p].ntaps != 1) return 0;
    const size_t lds = (size_t)nphase * cout_pad * c.Cin * 2;
    if (lds > 64 * 1024) return 0;
    PwArgs a;
    a.x = (const bf16_t*)c.x; a.y = (bf16_t*)c.y; a.res = (const bf16_t*)c.res;
    a.bias = c.bias; a.scale = c.scale; a.shift = c.shift;
    a.Do = c.Do; a.Ho = c.Ho; a.Wo = c.Wo; a.OD = c.OD; a.OH = c.OH; a.OW = c.OW; a.osd = c.osd; a.osh = c.osh; a.osw = c.osw;
    a.ldc = c.ldc; a.k_pad = c.k_pad; a.flags = c.flags; a.nphase = nphase;
    a.ntile = c.M / 64;
    for (int p = 0; p < nphase; ++p) {
        a.phase[p].w = (const bf16_t*)c.phase[p].w;
        a.phase[p].ood = c.phase[p].ood; a.phase[p].ooh = c.phase[p].ooh; a.phase[p].oow = c.phase[p].oow;
    }
    a.plain = (nphase == 1 && c.osd == 1 && c.osh == 1 && c.osw == 1 && c.OD == c.Do && c.OH == c.Ho && c.OW == c.Wo && !c.phase[0].ood &&
               !c.phase[0].ooh && !c.phase[0].oow)
                  ? 1
                  : 0;
    int rc;
    const int kb = c.Cin / 16, nb = cout_pad / 32;
#define PW_CASE(KB_, NB_) \
    if (kb == KB_ && nb == NB_) { rc = launch_pw<KB_, NB_>(a, s); return rc == LT_OK ? 1 : rc; }
    PW_CASE(1, 1) PW_CASE(2, 1) PW_CASE(4, 1) PW_CASE(8, 1) PW_CASE(1, 2) PW_CASE(2, 2) PW_CASE(4, 2) PW_CASE(8, 2)
#undef PW_CASE
    return 0;
}

}  // namespace lt
