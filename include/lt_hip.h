/*
 * lt_hip.h -- flat C ABI of liblt_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * volumetric-triangulation forward path of karfly/learnable-triangulation-pytorch.
 *
 * The reference has no FFI/operator registry: the boundary it offers is a set
 * of Python callables whose arithmetic is done by ATen kernels.  Each entry point below replaces the
 * ATen kernels behind one of those callables; the Python host in
 * learnable-triangulation-pytorch_amd/mvn/ keeps the callables' names and signatures and binds
 * these symbols with ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is enqueued on
 *     it and nothing synchronises, allocates or frees: every call is hipGraph-capturable;
 *   - return value: 0 = LT_OK, negative = error; lt_last_error() holds a message (thread local);
 *   - activations are channels-last: 2D maps N,H,W,C and volumes N,D,H,W,C, element type `dtype`
 *     (LT_F32, or LT_BF16 with fp32 accumulation); 2D tensors are the D == 1 case of 3D;
 *   - thread-safe for distinct streams.
 */
#ifndef LT_HIP_H
#define LT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_ABI_VERSION 1

enum { LT_F32 = 0, LT_BF16 = 1,
       LT_FP8 = 2 /* OCP e4m3 ("e4m3fn", torch.float8_e4m3fn; what gfx950's v_cvt_pk_fp8_f32 and fp8 MFMAs use), one byte per element: operands of
                     lt_conv_fwd only (BASELINE config 5: fp8 MFMA for the V2V 3D convolutions of the training step) */ };
enum { LT_OK = 0, LT_ERR_INVALID = -1, LT_ERR_UNSUPPORTED = -2, LT_ERR_LAUNCH = -3 };

/* view-aggregation modes of op.unproject_heatmaps (mvn/utils/op.py:149-164) */
enum { LT_AGG_SUM = 0, LT_AGG_MAX = 1, LT_AGG_SOFTMAX = 2, LT_AGG_CONF = 3,
       LT_AGG_CONF_NORM = 4 /* 'conf_norm': confidences divided by their sum over views first (triangulation.py:268-269) */ };

/* epilogue flags of lt_conv_fwd */
enum {
    LT_EPI_RELU_PRE = 1,  /* ReLU before the residual add  (Upsample3DBlock + skip, v2v.py:121-136) */
    LT_EPI_RELU_POST = 2, /* ReLU after the residual add   (Bottleneck / Res3DBlock, pose_resnet.py:92-93, v2v.py:42) */
    LT_EPI_STORE_F32 = 4, /* store fp32 even when dtype is bf16 (V2V logits feeding the soft-argmax) */
    LT_EPI_SIGMOID = 8,   /* v = 1/(1+exp(-v)) last (GlobalAveragePoolingHead, pose_resnet.py:160) */
    LT_EPI_RES_F32 = 64,  /* with LT_EPI_STORE_F32 on a bf16 convolution: the residual is fp32 too (the mixed-precision training step adds the input
                             gradient that is already there in the epilogue of the next input-gradient convolution) */
    LT_BN_FROZEN = 32,    /* lt_bn_act_bwd only: mean / var are FROZEN running statistics (a BatchNorm module in eval() inside a training step):
                             dy = gamma invstd g, without the batch-statistics terms; dgamma / dbeta as usual */
    LT_BN_Y_BF16 = 128,   /* lt_bn_act_fwd / lt_bn_act_bwd: the convolution output y is a bf16 tensor (the mixed-precision training step stores it in
                             16 bits; lt_bn_stats_fwd takes dtype = LT_BF16 for the same tensor); C % 4 == 0 */
    LT_ACT_BF16 = 256     /* lt_bn_act_fwd / lt_bn_act_bwd / lt_act_bwd: the 16-bit-activation training step (BASELINE config 5: "fp16 activations" -- bf16 here,
                             the 16-bit type whose MFMA products need no loss scaling): every activation-shaped tensor of the call other than y -- residual
                             and z (forward); dz, residual, dy, dres (backward) -- is a bf16 tensor behind the same pointer argument; the separate bf16
                             copies (z_bf16 / dy_bf16) must be NULL; C % 4 == 0 for the BatchNorm calls.  Statistics, gamma / beta and their gradients stay fp32. */
};

const char* lt_last_error(void);
int lt_abi_version(void);
/* fills CU count, LDS bytes per CU and the gcn arch name (e.g. "gfx950") of the current device */
int lt_device_info(int* cu_count, int* lds_per_cu, char* arch, int arch_len);

/* ---------------------------------------------------------------------------------------------
 * Generalised convolution = implicit GEMM on MFMA (fp32: v_mfma_f32_32x32x2_f32 / 16x16x4_f32,
 * exact fp32; bf16: v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16, fp32 accumulate).
 * Replaces F.conv2d / F.conv3d / F.conv_transpose2d / F.conv_transpose3d + eval BatchNorm + ReLU +
 * residual add as used by PoseResNet.forward (mvn/models/pose_resnet.py:293-318, :75-95),
 * process_features (mvn/models/triangulation.py:238-240) and V2VModel.forward (mvn/models/v2v.py:7-66,
 * :164-169).
 *
 *   for every phase p, every point o = (n, od, oh, ow) of the iteration space N x Do x Ho x Wo and
 *   every output channel co:
 *     acc = sum_{tap t, ci} x[n, od*sd - pd + dd_t, oh*sh - ph + dh_t, ow*sw - pw + dw_t, ci]
 *                           * w_p[co][t*Cin + ci]                      (out-of-range taps read 0)
 *     v   = (acc + bias[co]) * scale[co] + shift[co]    (conv bias, then eval BatchNorm as ATen evaluates it:
 *                                                         scale = weight/sqrt(var+eps), shift = bias_bn - mean*scale;
 *                                                         NULL = 0 / 1 / 0; arrays hold cout_pad floats)
 *     if RELU_PRE v = max(v,0);  if residual v += residual[same place as y];  if RELU_POST v = max(v,0)
 *     y[n, od*osd + ood_p, oh*osh + ooh_p, ow*osw + oow_p, co] = v
 *
 * dtype == LT_FP8: x and the weights are e4m3 bytes (per-tensor scaled by the caller: x ~ sx * x8, w ~ sw * w8, see lt_quant_fp8), the
 * products run on v_mfma_f32_*_fp8_fp8 with fp32 accumulation, the result is stored as fp32 (LT_EPI_STORE_F32, a residual is fp32 too:
 * LT_EPI_RES_F32) or -- without LT_EPI_STORE_F32, Cout % 8 == 0 -- as bf16 (y and the residual are bf16 tensors: the 16-bit-activation training
 * step); the caller passes sx * sw in `scale` and the convolution's bias in `shift` (bias = NULL).  Cin >= 16, k_pad % 128 == 0 (one 128-byte
 * K step = 128 elements); generic implicit-GEMM kernel only.
 *
 * A plain convolution is one phase with out_stride 1 / out_off 0; a stride-2 transposed convolution
 * is 2^nd phases (one per output parity) with out_stride 2.  A single one-tap phase with unit strides,
 * zero pads and equal input / iteration / output extents IS a pointwise (1x1) convolution -- its tap must
 * be (0,0,0,0) -- and takes a fast path without tap or bounds logic.
 * -------------------------------------------------------------------------------------------*/
#define LT_CONV_MAX_PHASES 8

typedef struct lt_conv_phase {
    const void* weight;  /* [cout_pad][k_pad] elements of `dtype`, k = tap*Cin + ci, zero padded      */
    const int32_t* taps; /* ntaps x 4 int32: dd, dh, dw, element offset ((dd*H + dh)*W + dw)*Cin     */
    int32_t ntaps;
    int32_t out_off[3];  /* ood, ooh, oow                                                            */
    const void* weight_frag; /* optional (NULL = none): the same bf16 weights in an MFMA fragment order, so that a kernel can read
                                its weight operand with coalesced loads instead of staging it through LDS */
    int32_t weight_frag_layout; /* 0 = none; 1 = lt_conv_pack_weights (Cout % 256 == 0 layers, 288 x 256 / 144 x 256 kernels on the 16x16x32 MFMA);
                                   2 = lt_conv_pack_weights_t32 (3x3x3 64->64 / 32->64 / 128->128 / 16->32 halo kernel; round 5: the 2D halo kernel for 256->256
                                       layers on maps of 24 n x 8 m pixels -- 3x3 / stride 1 / pad 1, or the four 2 x 2-tap phases of a 4x4 / stride-2 /
                                       pad-1 transposed convolution, every phase packed with its own ntaps; taps must lie within one pixel of the output pixel);
                                   3 = lt_conv_pack_weights32 (288 x 256 kernel on the 32x32x16 MFMA) */
} lt_conv_phase;

typedef struct lt_conv_desc {
    int32_t dtype;              /* LT_F32 | LT_BF16                                                  */
    int32_t N, D, H, W, Cin;    /* input tensor (Cin a power of two, >= 16 bytes per pixel)           */
    int32_t Do, Ho, Wo;         /* iteration space per sample                                        */
    int32_t stride[3], pad[3];  /* sd,sh,sw / pd,ph,pw                                               */
    int32_t OD, OH, OW;         /* output tensor spatial dims                                        */
    int32_t out_stride[3];      /* osd, osh, osw                                                     */
    int32_t Cout, ldc;          /* real output channels; output pixel stride in elements (>= Cout)   */
    int32_t cout_pad, k_pad;    /* padded weight dims: cout_pad % tile_n == 0, k_pad % (128 B) == 0  */
    int32_t nphase;
    int32_t flags;              /* LT_EPI_*                                                          */
    int32_t tile;               /* 0 = choose; else one of LT_TILE_* (tests / tuning)                */
    int32_t stages;             /* 0 = choose; 2 or 3 = LDS-DMA ring depth of the v2 kernels (tuning)       */
    lt_conv_phase phase[LT_CONV_MAX_PHASES];
} lt_conv_desc;

/* bf16 weights [cout_pad][k_pad] (cout_pad % 16 == 0, k_pad % 32 == 0) -> cout_pad * k_pad elements in fragment order
 * ([k_pad / 32][cout_pad / 16][64 lanes][8]: lane l holds column 16 t + (l & 15), K elements 32 s + 8 (l >> 4) .. + 7). */
int lt_conv_pack_weights(const void* weight, int32_t cout_pad, int32_t k_pad, void* packed, void* stream);
/* The fragment order of the TRANSPOSED product (weights as the first MFMA operand), used by the 3x3x3 64 -> 64 halo kernel:
 * [tap][cin / 16][cout_pad / 32][64 lanes][8]; lane (r = l & 31, h = l >> 5) holds output channel
 * 32 b + 16 (r >> 4) + 8 ((r >> 2) & 1) + 4 ((r >> 3) & 1) + (r & 3), K elements 16 g + 8 h .. + 7 of the tap.
 * lt_conv_phase.weight_frag_layout says which of the orders weight_frag holds; a kernel only uses the one it was written for. */
int lt_conv_pack_weights_t32(const void* weight, int32_t cout_pad, int32_t k_pad, int32_t cin, int32_t ntaps, void* packed,
                             void* stream);
/* B-fragment order of the 32x32x16 MFMA (layout 3, conv_igemm7): [k_pad / 32][cout_pad / 32][2 K halves][64 lanes][8]; lane l of
 * fragment (step, block, kk) holds column 32 block + (l & 31), K elements 32 step + 16 kk + 8 (l >> 5) .. + 7.  cout_pad % 32 == 0. */
int lt_conv_pack_weights32(const void* weight, int32_t cout_pad, int32_t k_pad, void* packed, void* stream);

enum { LT_TILE_AUTO = 0,
       /* v1: register-staged tiles (kept for A/B runs and as a cross-check) */
       LT_TILE_128x128 = 1, LT_TILE_128x64 = 2, LT_TILE_256x32 = 3, LT_TILE_256x16 = 4, LT_TILE_64x64 = 5,
       /* v2 (default): LDS-DMA staging + LDS-transposed 16-byte epilogue */
       LT_TILE2_128x128 = 11, LT_TILE2_128x64 = 12, LT_TILE2_256x32 = 13, LT_TILE2_256x16 = 14, LT_TILE2_64x64 = 15,
       /* stride-1 3^3 / 7^3 conv3d with the input halo tile resident in LDS (auto-selected where it applies) */
       LT_TILE_HALO = 20,
       /* 288-row tile, eight waves, three-stage ring (bf16, auto-selected for the wide ResNet levels) */
       LT_TILE3_288 = 30,
       LT_TILE_DIRECT = 99 /* scalar fp32 VALU kernel, debug cross-check only */ };

int lt_conv_fwd(const lt_conv_desc* desc, const void* x, const float* bias, const float* scale, const float* shift,
                const void* residual, void* y, void* stream);
/* lt_conv_fwd whose residual is COMPUTED instead of read: y = relu_post((acc + bias) * scale + shift + W_skip . skip.x[same voxel]) -- the second
 * convolution of a Res3DBlock whose skip connection is a 1x1x1 convolution + BatchNorm (mvn/models/v2v.py:20-42, the 16 -> 32 block at :76).  The caller
 * folds the skip branch's BatchNorm scale into W_skip (fp32 product, then one bf16 rounding) and its (bias * scale + shift) into `shift`; the skip
 * convolution's own launch, its Cout-channel output and the read of that tensor disappear.  Covered: bf16, 3x3x3 / stride 1 / pad 1, 32 -> 32 channels,
 * skip.cin == 16, shapes the column-walk halo kernel takes (D % 4 == 0 with D >= 8, H % 8 == 0, W % 8 == 0, N * (H / 8) * (W / 8) a multiple of 8 and
 * >= 256, no LT_EPI_RELU_PRE / LT_EPI_STORE_F32) -- anything else returns LT_ERR_UNSUPPORTED (there is no fallback: callers ask before they record).
 * skip.x: N,D,H,W,cin channels-last bf16;  skip.weight_frag: lt_conv_pack_weights_t32 of [32][16] (ntaps 1, cin 16). */
typedef struct lt_conv_skip {
    const void* x;
    int32_t cin;
    const void* weight_frag;
} lt_conv_skip;
int lt_conv_skip_fwd(const lt_conv_desc* desc, const void* x, const float* bias, const float* scale, const float* shift,
                     const lt_conv_skip* skip, void* y, void* stream);
/* A 1x1 convolution over the channel concatenation of TWO tensors, the second one optionally subsampled:
 *   acc[m][co] = sum_{ci < Cin} x[m][ci] w[co][ci]  +  sum_{cj < second.cin} x2[n, oh * s, ow * s, cj] w[co][Cin + cj]        (m = (n, oh, ow))
 * followed by lt_conv_fwd's epilogue.  It is the last 1x1 convolution of a Bottleneck block TOGETHER with the block's `downsample` branch
 * (mvn/models/pose_resnet.py:75-95: out = bn3(conv3(t2)); residual = downsample(x); out += residual; relu -- the first blocks of layer2-4, whose
 * downsample is a stride-2 1x1 convolution + BatchNorm of the block input): the caller folds both BatchNorm scales into the two weight blocks (fp32
 * product, one bf16 rounding), passes scale = NULL and shift = shift3 + shift_d, and the downsample's launch, its C-channel output and the read of
 * that tensor as the residual disappear.  desc: the pointwise convolution over x ([N][Ho][Wo][Cin], one phase, one tap) with k_pad = Cin + second.cin
 * and weights [cout_pad][k_pad] = [w3 | w_d], ALSO given in fragment layout 3 (lt_conv_pack_weights32); bf16, Cin % 32 == 0, second.cin % 32 == 0,
 * k_pad % 64 == 0, cout_pad % 256 == 0; second: x2 [N][H][W][cin] with H = stride * Ho, W = stride * Wo, stride 1 or 2.  Anything else:
 * LT_ERR_UNSUPPORTED (one kernel, no fallback). */
typedef struct lt_conv_cat2 {
    const void* x;
    int32_t cin, H, W, stride;
} lt_conv_cat2;
int lt_conv_cat2_fwd(const lt_conv_desc* desc, const void* x, const lt_conv_cat2* second, const float* bias, const float* scale, const float* shift,
                     const void* residual, void* y, void* stream);
/* weight padding rule (every tile's N divides it): cout_pad = 16 if Cout <= 16, 32 if <= 32, 64 if <= 64,
 * else Cout rounded up to a multiple of 128; bias/scale/shift arrays hold cout_pad floats. */
int lt_conv_cout_pad(int32_t cout);
/* Samples per launch of lt_conv_fwd / lt_conv_skip_fwd / lt_conv_cat2_fwd for tensors of `per_sample` elements per sample (the largest of input,
 * output, second source).  The kernels index ONE launch with 32-bit element offsets; the entry points accept any batch and walk batches beyond 2^31
 * elements per tensor in chunks of this many samples, every per-sample pointer advanced in 64 bits (BASELINE config 4 at 32 samples per GPU --
 * 32 x 128^3 voxels x 32 channels = 2^31 elements -- is one call; reference: mvn/models/triangulation.py:245-355 has no such limit).  Returns N
 * when the whole batch fits, a multiple of 8 where possible otherwise, 0 when a single sample is too large. */
int32_t lt_conv_chunk_samples(int32_t N, int64_t per_sample);

/* max pooling, channels-last, window k / stride s / zero-size padding p per dim (padding never wins):
 * F.max_pool2d(x,3,2,1) (pose_resnet.py:297) and F.max_pool3d(x,2,2) (v2v.py:51) */
int lt_maxpool_fwd(int32_t dtype, const void* x, void* y, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C,
                   const int32_t k[3], const int32_t s[3], const int32_t p[3], void* stream);

/* x.mean over all pixels per channel (GlobalAveragePoolingHead.forward, pose_resnet.py:168-170):
 * x N,HW,C channels-last -> y N,C */
int lt_global_avgpool(int32_t dtype, const void* x, void* y, int32_t N, int32_t HW, int32_t C, void* stream);

/* images N,C,H,W fp32 (the reference's input layout, datasets/utils.py:45-52) -> N,H,W,c_pad `dtype`,
 * channels >= C zero filled */
int lt_nchw_to_nhwc(int32_t dtype, const float* x, void* y, int32_t N, int32_t C, int32_t HW, int32_t c_pad, void* stream);
/* channels-last `dtype` (pixel stride ld) -> N,C,HW fp32 (API outputs) */
int lt_nhwc_to_nchw_f32(int32_t dtype, const void* x, float* y, int32_t N, int32_t C, int32_t HW, int32_t ld, void* stream);

/* The reduction half of a split-K convolution: lt_conv_fwd run with S tap-group PHASES (identity epilogue, LT_EPI_STORE_F32, phase p
 * writing depth slice p of a [N][S * Do][Ho][Wo][C] fp32 tensor) leaves S partial sums per output; this adds them in the order p = 0 ..
 * S-1 (fp32) and applies the convolution's real epilogue -- (sum + bias) * scale + shift, LT_EPI_RELU_PRE, residual add, LT_EPI_RELU_POST
 * -- like the conv kernels do.  partial: [N][S][rows_per_sample][C] fp32; residual / y: [N][rows_per_sample][C] of `dtype`.  Used for
 * V2V's 3^3 128 -> 128 layers at the 8^3 / 4^3 / 2^3 levels (mvn/models/v2v.py:78-90), whose 54-step K loop is latency-bound. */
int lt_splitk_reduce(int32_t dtype, const float* partial, int32_t S, int64_t N, int64_t rows_per_sample, int32_t C, const float* bias,
                     const float* scale, const float* shift, const void* residual, void* y, int32_t flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Voxel-grid construction: mvn/models/triangulation.py:298-339 + volumetric.rotate_coord_volume
 * (mvn/utils/volumetric.py:102-114), fp32 in the reference's operation order:
 *   X = R_b * ((pos_b + step*idx) - center_b) + center_b        idx = (i,j,k), 'ij' meshgrid
 * pos/center: B x 3 fp32, rot: B x 9 fp32 row-major.  coords: B,V,V,V,3 fp32.
 * cmu_transfer != 0 applies permute(0,2,1,3) + flip(axis 1) (triangulation.py:336-339).
 * -------------------------------------------------------------------------------------------*/
int lt_coord_volumes(const float* pos, const float* center, const float* rot, float step, int32_t B, int32_t V,
                     int32_t cmu_transfer, float* coords, void* stream);
/* volumetric.rotate_coord_volume (mvn/utils/volumetric.py:102-114) for an arbitrary point set:
 * y[i] = rot (3x3 row-major, device) * x[i], i < n; x, y: n x 3 fp32 */
int lt_rotate_points(const float* x, const float* rot, float* y, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * op.unproject_heatmaps (mvn/utils/op.py:99-166) fused with
 * multiview.project_3d_points_to_image_plane_without_distortion (mvn/utils/multiview.py:89-110):
 * per voxel and view: project, z<=0 mask, perspective divide, the reference's (h,w)-swapped
 * normalisation, bilinear sample (zeros padding, align_corners=True), then aggregate over views.
 *   feats : B,NV,h,w,C channels-last `dtype`     proj: B,NV,3,4 fp32
 *   coords: B,v0,v1,v2,3 fp32 (any grid; arbitrary point sets use v0 = v1 = 1)
 *   conf  : B,NV,C fp32 (LT_AGG_CONF*)           out : B,v0,v1,v2,C channels-last `dtype`
 * One 16-byte channel vector per lane (a voxel's C channels = consecutive lanes -> every bilinear tap
 * and every output voxel is one contiguous run); 4x4x16 voxel bricks per workgroup keep the projected
 * footprint compact in L1/L2; with B % 8 == 0 sample b is pinned to XCD b % 8 (private L2).
 * -------------------------------------------------------------------------------------------*/
int lt_unproject_fwd(int32_t dtype, const void* feats, const float* proj, const float* coords, const float* conf,
                     void* out, int32_t B, int32_t NV, int32_t C, int32_t h, int32_t w, int32_t v0, int32_t v1,
                     int32_t v2, int32_t agg, void* stream);
/* The same over the cuboid grid of the model (triangulation.py:298-339) described by lt_coord_volumes' arguments instead of a
 * coordinate tensor: the voxel centres are computed in registers (bit-identical to lt_coord_volumes) and WRITTEN to coords_out
 * (B,V,V,V,3 fp32, a returned tensor of the forward) on the way -- the 12 bytes per voxel are never read back.  Fused for bf16,
 * C = 32, 4 or 8 views, view softmax, V % 16 == 0; any other configuration runs lt_coord_volumes + lt_unproject_fwd inside. */
int lt_unproject_grid_fwd(int32_t dtype, const void* feats, const float* proj, const float* pos, const float* center, const float* rot,
                          float step, int32_t cmu_transfer, float* coords_out, const float* conf, void* out, int32_t B, int32_t NV,
                          int32_t C, int32_t h, int32_t w, int32_t V, int32_t agg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * op.integrate_tensor_3d_with_coordinates (mvn/utils/op.py:84-96):
 *   p = softmax_over_voxels(mult * logits[b,j,:])  (or relu(mult*logits) when softmax == 0)
 *   kp[b,j,:] = sum_vox p * coords[b,vox,:]
 * logits: fp32, element (b,j,vox) at b*J*nvox + vox*ld + j when channels_last (ld >= J) else
 * b*J*nvox.. (b*J + j)*nvox + vox.  probs: B,J,nvox fp32 (always joint-major, the API layout) or
 * NULL.  workspace: lt_softargmax3d_workspace() bytes.
 * -------------------------------------------------------------------------------------------*/
size_t lt_softargmax3d_workspace(int32_t B, int32_t J, int64_t nvox);
int lt_softargmax3d_fwd(const float* logits, const float* coords, float mult, int32_t softmax, int32_t channels_last,
                        int32_t ld, float* kp, float* probs, int32_t B, int32_t J, int64_t nvox, void* workspace,
                        void* stream);

/* op.integrate_tensor_2d (mvn/utils/op.py:11-47) on N,J,h,w fp32 heatmaps (joint-major): 2D
 * soft-argmax; writes coords N,J,2 (x,y) and the normalised heatmaps (or NULL). */
int lt_softargmax2d_fwd(const float* heatmaps, float mult, int32_t softmax, float* coords, float* probs, int32_t NJ,
                        int32_t h, int32_t w, void* stream);
/* Backward of the two ops above and of lt_triangulate_dlt (training of AlgebraicTriangulationNet, train.py:189-236): what autograd derives in the
 * reference through softmax / sums (op.py:23-45) and through torch.svd (multiview.py:163).  lt_softargmax2d_bwd: probs / coords are the forward's
 * outputs, grad_coords N*J,2 -> grad_heatmaps (both modes; a gradient on the returned heatmaps is not supported).  lt_triangulate_dlt_bwd:
 * grad_out B,J,3 -> grad_points B,NV,J,2 and grad_conf B,NV,J (may be NULL); fp64 inside, the forward's Jacobi eigen-decomposition recomputed. */
int lt_softargmax2d_bwd(const float* probs, const float* coords, const float* grad_coords, float mult, int32_t softmax, float* grad_heatmaps,
                        int32_t NJ, int32_t h, int32_t w, void* stream);
int lt_triangulate_dlt_bwd(const float* proj, const float* points, const float* conf, const float* grad_out, float* grad_points, float* grad_conf,
                           int32_t B, int32_t NV, int32_t J, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-step pieces outside the convolutions: what torch.autograd derives in the
 * reference for op.unproject_heatmaps and op.integrate_tensor_3d_with_coordinates, the fused VolumetricCELoss
 * (mvn/models/loss.py:52-80; the reference walks samples and joints in Python with a .cpu() per sample), and the
 * batch statistics of training-mode BatchNorm.  All gradients are fp32.
 *
 * lt_unproject_bwd: grad_out B,v0,v1,v2,C (channels-last, fp32; the coordinate volume's grid shape) -> grad_feats B,NV,h,w,C fp32 (written
 *   completely, no pre-zeroing), grad_conf B,NV,C or NULL.  Autograd semantics of op.py:113-162: nothing flows to the grid / projections;
 *   depth <= 0 samples pass no gradient but their zero takes part in the view softmax, d out/d x_v = w_v (1 + x_v - out); 'max': the first
 *   maximal view; LT_AGG_CONF: g * conf_v and grad_conf_v = sum over voxels of g * x_v; LT_AGG_CONF_NORM: conf is the RAW head output,
 *   normalised over the views as in the forward (triangulation.py:268-269), grad_conf is the gradient of the raw values.  NV <= 8,
 *   C % 4 == 0.  A deterministic GATHER (bitwise repeatable; no global atomics): workspace from lt_unproject_bwd_workspace (bytes for the
 *   whole batch; with less -- at least one sample's share -- the batch is walked in chunks).  C > 64 or not a power of two (and
 *   LT_UNPROJ_BWD_ATOMICS=1): the round-2 scatter by float atomics (no workspace, not repeatable, no conf_norm).
 * lt_softargmax3d_bwd: probs B,J,nvox and kp B,J,3 are the forward's outputs; grad_kp B,J,3; an optional SPARSE gradient on the
 *   returned probabilities (one voxel per (b,j): gp_idx / gp_val, what lt_volumetric_ce_fwd produces) -> grad_logits
 *   (planar B,J,nvox, or channels-last B,nvox,J when channels_last != 0):  mult * p_i * (a_i - sum_j p_j a_j), a_i = g_kp . X_i + gp_i.
 * lt_volumetric_ce_fwd: per (b,j) the voxel nearest to keypoints_gt (first minimum, like torch.argmin), terms[b,j] =
 *   validity * -log(p + 1e-6), idx[b,j], grad_val[b,j] = d(sum(terms) / (B J)) / d probs[b,j,idx].  The loss is sum(terms) / (B J).
 * lt_bn_stats_fwd: x rows x C channels-last -> per-channel mean and BIASED variance (fp64 accumulation); running statistics
 *   (may be NULL) are updated the way torch does: (1 - momentum) * running + momentum * stat, variance unbiased.
 * -------------------------------------------------------------------------------------------*/
size_t lt_unproject_bwd_workspace(int32_t B, int32_t NV, int32_t C, int32_t v0, int32_t v1, int32_t v2);
int lt_unproject_bwd(int32_t dtype, const void* feats, const float* proj, const float* coords, const float* conf, const float* grad_out,
                     float* grad_feats, float* grad_conf, int32_t B, int32_t NV, int32_t C, int32_t h, int32_t w, int32_t v0, int32_t v1,
                     int32_t v2, int32_t agg, void* workspace, size_t workspace_bytes, void* stream);
int lt_softargmax3d_bwd(const float* probs, const float* coords, const float* kp, const float* grad_kp, const int32_t* gp_idx,
                        const float* gp_val, float multiplier, int32_t softmax, int32_t channels_last, float* grad_logits, int32_t B,
                        int32_t J, int64_t nvox, void* stream);
/* the same with a DENSE gradient on the returned probabilities as well (gp_dense: B,J,nvox fp32 or NULL -- any loss on the volumes, what the reference's
 * autograd accepts at train.py:222-230): a_i += gp_dense_i; workspace_bj: B * J floats (sum_i p_i gp_dense_i per joint, softmax mode). */
int lt_softargmax3d_bwd_dense(const float* probs, const float* coords, const float* kp, const float* grad_kp, const int32_t* gp_idx,
                              const float* gp_val, const float* gp_dense, float* workspace_bj, float multiplier, int32_t softmax,
                              int32_t channels_last, float* grad_logits, int32_t B, int32_t J, int64_t nvox, void* stream);
int lt_volumetric_ce_fwd(const float* coords, const float* probs, const float* keypoints_gt, const float* validity, float* terms,
                         int32_t* idx, float* grad_val, int32_t B, int32_t J, int64_t nvox, void* stream);
size_t lt_bn_stats_workspace(int64_t rows, int32_t C);
int lt_bn_stats_fwd(int32_t dtype, const void* x, int64_t rows, int32_t C, float* mean, float* var, float* running_mean,
                    float* running_var, float momentum, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step around the convolutions (fp32, channels-last rows x C): the reference's modules in training mode
 * (train.py:233-243: zero_grad / backward / step) -- see lt_train.py for the tape that strings them together.
 * lt_bn_act_fwd : z = act(gamma (y - mean) / sqrt(var + eps) + beta, residual) with lt_conv_fwd's LT_EPI_RELU_* flags; mean / var
 *   from lt_bn_stats_fwd (batch statistics).  C % 4 == 0.
 * lt_bn_act_bwd : its autograd: g = dz * relu mask; dbeta = sum g; dgamma = sum g x^; dy = gamma invstd (g - dbeta/n - x^ dgamma/n)
 *   (LT_BN_FROZEN in flags: mean / var are running statistics that do not depend on the batch: dy = gamma invstd g);
 *   dres (may be NULL) = the residual input's gradient, added to the buffer when accumulate_res.  workspace: lt_bn_act_bwd_workspace.
 * lt_act_bwd    : layers without BatchNorm: dy = dz * mask(z, flags) (+ dres); LT_EPI_SIGMOID: dy = dz * z * (1 - z).  LT_EPI_RELU_PRE together
 *   with a residual is refused (LT_ERR_UNSUPPORTED): the sign of v cannot be rebuilt from z = relu(v) + res.
 * lt_channel_sum: out[c] (+)= sum over rows of x[row][c] (bias gradients), fp64 accumulation.
 * lt_maxpool_bwd: dx (pre-zeroed / accumulated) += dy at the first maximal element of every window; overlapping windows (k > s) are walked in
 *   ceil(k / s)^3 classes of mutually disjoint windows, one launch each: no atomics, bitwise repeatable.
 * lt_conv_wgrad : dw[co][tap * Cin + ci] (=|+=) sum_m dy[m][co] * x[m @ tap][ci] over the GEMM rows m = (n, od, oh, ow) of the forward
 *   convolution described by (N, D, H, W, Cin, Do, Ho, Wo, stride, pad, taps); exact-fp32 MFMA, no atomics (the pixel range is cut into
 *   slabs whose partial sums are added in a fixed order: workspace of lt_conv_wgrad_workspace(rows = N*Do*Ho*Wo, cout_pad, k_pad) bytes,
 *   may be 0).  For a transposed convolution swap the roles (dy := the layer's INPUT at its own resolution, x := the output gradient,
 *   stride 2): dw[ci][tap * Cout + co].
 * lt_adam_step  : torch.optim.Adam's single-tensor update (bias-corrected, eps outside the sqrt).
 * -------------------------------------------------------------------------------------------*/
int lt_bn_act_fwd(const void* y /* fp32, or bf16 with LT_BN_Y_BF16 */, const float* mean, const float* var, const float* gamma, const float* beta, const void* residual, void* z,
                  void* z_bf16 /* optional: a bf16 copy of z on the way (mixed-precision training), or NULL */, int64_t rows, int32_t C, float eps,
                  int32_t flags, void* stream);
size_t lt_bn_act_bwd_workspace(int64_t rows, int32_t C);
int lt_bn_act_bwd(const void* dz, const void* y /* fp32, or bf16 with LT_BN_Y_BF16 */, const void* residual, const float* mean, const float* var, const float* gamma, const float* beta,
                  void* dy, void* dy_bf16 /* optional bf16 copy of dy, or NULL */, float* dgamma, float* dbeta, void* dres, int32_t accumulate_res,
                  int64_t rows, int32_t C, float eps, int32_t flags, void* workspace, void* stream);
int lt_act_bwd(const void* dz, const void* z, const void* residual, void* dy, void* dres, int32_t accumulate_res, int64_t total, int32_t flags,
               void* stream);
size_t lt_channel_sum_workspace(int64_t rows, int32_t C);
int lt_channel_sum(const float* x, int64_t rows, int32_t C, float* out, int32_t accumulate, void* workspace, void* stream);
/* the same over a tensor of element type `dtype` (LT_F32 | LT_BF16: the 16-bit-activation training step); same workspace */
int lt_channel_sum_dt(int32_t dtype, const void* x, int64_t rows, int32_t C, float* out, int32_t accumulate, void* workspace, void* stream);
int lt_maxpool_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, const int32_t k[3],
                   const int32_t s[3], const int32_t p[3], void* stream);
/* x, dy, dx of element type `dtype` (LT_F32 | LT_BF16); the (up to ceil(k / s)^3) contributions an input collects are added in the class order, in `dtype` */
int lt_maxpool_bwd_dt(int32_t dtype, const void* x, const void* dy, void* dx, int32_t N, int32_t D, int32_t H, int32_t W, int32_t C, const int32_t k[3],
                      const int32_t s[3], const int32_t p[3], void* stream);
size_t lt_conv_wgrad_workspace(int64_t rows, int32_t cout_pad, int32_t k_pad);
int lt_conv_wgrad(const float* dy, const float* x, const int32_t* taps, float* dw, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin, int32_t Do,
                  int32_t Ho, int32_t Wo, const int32_t stride[3], const int32_t pad[3], int32_t Cout, int32_t ldy, int32_t cout_pad, int32_t k_pad,
                  int32_t ntaps, int32_t accumulate, void* workspace, void* stream);
/* The same weight gradient on the bf16 MFMA (mixed-precision training, BASELINE config 5): the reduction index is the pixel and the 16-bit MFMA
 * wants 8 consecutive K values per lane, so the operands are IMAGE-OCTET packed -- the one axis no tap ever shifts:
 *   lt_pack_n8_bf16: src [N][P][ld >= C] fp32 (P pixels per image) -> dst [ceil(N / 8)][P][C] elements of 8 bf16 (16 bytes) = the values of images
 *   8g .. 8g+7 at that pixel and channel, zero for images past N; dst is 16-byte aligned, lt_pack_n8_bf16_bytes(N, P, C) bytes.
 *   lt_conv_wgrad_bf16: arguments as lt_conv_wgrad with dy16 = pack of dy [N][Do*Ho*Wo][ldy] and x16 = pack of x [N][D*H*W][Cin]; fp32
 *   accumulation, fixed summation order (slabs + ordered reduce: bitwise repeatable); workspace of lt_conv_wgrad_bf16_workspace(
 *   ceil(N / 8) * Do*Ho*Wo, cout_pad, k_pad) bytes.  Result = lt_conv_wgrad of the bf16-rounded operands up to fp32 summation order. */
size_t lt_pack_n8_bf16_bytes(int32_t N, int64_t P, int32_t C);
int lt_pack_n8_bf16(const float* src, void* dst, int32_t N, int64_t P, int32_t C, int32_t ld, void* stream);
/* the same from a bf16 tensor [N][P][ld] (the operand copy the mixed-precision convolutions read): C, ld multiples of 8, 16-byte aligned */
int lt_pack_n8_from_bf16(const void* src_bf16, void* dst, int32_t N, int64_t P, int32_t C, int32_t ld, void* stream);
size_t lt_conv_wgrad_bf16_workspace(int64_t octet_rows, int32_t cout_pad, int32_t k_pad);
int lt_conv_wgrad_bf16(const void* dy16, const void* x16, const int32_t* taps, float* dw, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin, int32_t Do,
                       int32_t Ho, int32_t Wo, const int32_t stride[3], const int32_t pad[3], int32_t Cout, int32_t ldy, int32_t cout_pad, int32_t k_pad,
                       int32_t ntaps, int32_t accumulate, void* workspace, void* stream);
/* The same GEMM straight from the channels-last bf16 tensors, without the packs (the 16-bit-activation step: the activations and their gradients
 * ARE the bf16 operands): dy16 [N][Do*Ho*Wo][ldy], x16 [N][D][H][W][ldx], 16-byte aligned; every lane of the 16x16x32 MFMA reads 4 or 8 consecutive
 * channels of the eight images of its pixel and transposes them in registers (v_perm_b32) into that many operands.  lt_conv_wgrad_bf16_nhwc_ok
 * -> 1 when the shape is covered: Cin a power of two, Cout / ldy multiples of 8 (4 when cout_pad <= 64), Cin / ldx multiples of 4 (8 when
 * cout_pad <= 64), k_pad a multiple of 4, 32-bit element offsets -- and NOT one of the V2V shapes lt_conv_wgrad_bf16 has LDS kernels for (3^3 / stride 1
 * / pad 1 bricks, the 7^3 front layer): those stay on the packed path.  Same result contract, same workspace as lt_conv_wgrad_bf16
 * (reference: autograd of nn.Conv2d / nn.ConvTranspose2d in pose_resnet.py inside train.py:217-243's backward). */
int lt_conv_wgrad_bf16_nhwc_ok(int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin, int32_t ldx, int32_t Do, int32_t Ho, int32_t Wo, const int32_t stride[3],
                               const int32_t pad[3], int32_t Cout, int32_t ldy, int32_t cout_pad, int32_t k_pad, int32_t ntaps);
int lt_conv_wgrad_bf16_nhwc(const void* dy16, const void* x16, const int32_t* taps, float* dw, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Cin, int32_t ldx,
                            int32_t Do, int32_t Ho, int32_t Wo, const int32_t stride[3], const int32_t pad[3], int32_t Cout, int32_t ldy, int32_t cout_pad,
                            int32_t k_pad, int32_t ntaps, int32_t accumulate, void* workspace, void* stream);
/* dst[i] = idx[i] >= 0 ? src[idx[i]] : 0 -- the layout changes between a Parameter's own layout and the GEMM layouts of its layer
 * (forward weights, input-gradient weights, weight-gradient blocks), with index maps built once when a training plan is recorded */
int lt_gather_f32(const float* src, const int32_t* idx, float* dst, int64_t n, void* stream);
/* the same with the result rounded to bf16 (the live weights of the mixed-precision step in the layout its bf16 convolutions read) */
int lt_gather_f32_bf16(const float* src, const int32_t* idx, void* dst_bf16, int64_t n, void* stream);
/* small helpers of the training tape, so that no torch op sits between the launches: y += x; dst[r][c] = c < C ? src[r][c] : 0 (dY of
 * the 17-joint layer widened to the power-of-two channel count lt_conv_fwd wants on its input); zero fill (scatter targets) */
int lt_add_f32(float* y, const float* x, int64_t n, void* stream);
int lt_pad_channels_f32(const float* src, float* dst, int64_t rows, int32_t C, int32_t c_pad, void* stream);
/* the same with a change of element type on the way (src_dtype / dst_dtype: LT_F32 | LT_BF16, round to nearest even; c_pad == C: a plain cast): the fp32 <-> bf16
 * boundaries of the 16-bit-activation training step (loss gradient in, unprojection backward in / out); 16-byte aligned pointers */
int lt_convert_pad(int32_t src_dtype, const void* src, int32_t dst_dtype, void* dst, int64_t rows, int32_t C, int32_t c_pad, void* stream);
int lt_zero(void* p, int64_t nbytes, void* stream);
/* *ptrs[i] += delta for n int64 scalars in device memory (ptrs: device array of n pointers): BatchNorm's num_batches_tracked counters of a
 * whole network in one launch (torch: one `num_batches_tracked += 1` kernel per layer, pose_resnet.py / v2v.py BatchNorm modules in train()) */
int lt_add_i64_multi(const void* ptrs, int32_t n, int64_t delta, void* stream);
/* backward of lt_global_avgpool (GlobalAveragePoolingHead's mean over the map, pose_resnet.py:166-168): dx[n][p][c] (=|+=) dy[n][c] / HW */
int lt_global_avgpool_bwd(const float* dy, float* dx, int32_t N, int32_t HW, int32_t C, int32_t accumulate, void* stream);
/* dy / dx of element type `dtype` (LT_F32 | LT_BF16: the 16-bit-activation training step) */
int lt_global_avgpool_bwd_dt(int32_t dtype, const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, int32_t accumulate, void* stream);
/* fp32 -> bf16, round to nearest even (operands of the mixed-precision training convolutions); 16-byte aligned pointers */
int lt_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
/* ---- fp8 (e4m3) operands for lt_conv_fwd(dtype = LT_FP8): per-tensor amax scaling, scale = amax / 448 (the largest e4m3 value), q = rne(x / scale).
 * lt_amax_f32     : *amax = max(*amax, max_i |x[i]|)  (zero *amax first, e.g. with lt_zero; an exact, order-independent maximum of non-negative
 *                   floats taken with integer atomics on their bit patterns: bitwise repeatable);
 * lt_quant_fp8    : q[i] = e4m3(x[i] / scale) with scale = *amax > 0 ? *amax / 448 : 1; writes the scale to *scale_out (device) -- the caller
 *                   multiplies the two operands' scales into lt_conv_fwd's `scale` array with lt_scale_product;
 * lt_gather_f32_fp8: the same for weights in one pass with the layout change of lt_gather_f32: q[i] = idx[i] >= 0 ? e4m3(src[idx[i]] / scale) : 0;
 * lt_scale_product: dst[i] = *a * *b for i < n. */
int lt_amax_f32(const float* x, int64_t n, float* amax, void* stream);
int lt_quant_fp8(const float* x, void* q, int64_t n, const float* amax, float* scale_out, void* stream);
/* the same two over a tensor of element type `dtype` (LT_F32 | LT_BF16: the bf16 activations / gradients of the 16-bit-activation training step) */
int lt_amax_dt(int32_t dtype, const void* x, int64_t n, float* amax, void* stream);
int lt_quant_fp8_dt(int32_t dtype, const void* x, void* q, int64_t n, const float* amax, float* scale_out, void* stream);
int lt_gather_f32_fp8(const float* src, const int32_t* idx, void* q, int64_t n, const float* amax, float* scale_out, void* stream);
int lt_scale_product(float* dst, int32_t n, const float* a, const float* b, void* stream);
/* many gathers in one launch.  jobs (device memory): njobs records of
 *   { const float* src; const int32_t* idx; float* dst; int64_t n; int32_t first_block; int32_t out_bf16; }   (40 bytes)
 * with first_block = running sum of ceil(n / 1024) over the preceding jobs; total_blocks = that sum over all jobs; out_bf16 != 0: dst is a bf16
 * array (the values are rounded to nearest even). */
int lt_gather_f32_multi(const void* jobs, int32_t njobs, int32_t total_blocks, void* stream);
/* the same update for MANY tensors in one launch.  jobs (device memory): njobs records of
 *   { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int64_t n; float lr; int32_t first_block; }   (48 bytes)
 * with first_block = running sum of ceil(n / 1024) over the preceding jobs; total_blocks = that sum over all jobs. */
int lt_adam_step_multi(const void* jobs, int32_t njobs, int32_t total_blocks, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                       void* stream);
int lt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int32_t step, void* stream);

/* multiview.triangulate_batch_of_points (mvn/utils/multiview.py:141-183): confidence-weighted DLT.
 * proj B,NV,3,4; points B,NV,J,2; conf B,NV,J or NULL; out B,J,3.  Smallest right singular vector
 * of the (2NV x 4) system by Jacobi eigen-iteration on A^T A in fp64. */
int lt_triangulate_dlt(const float* proj, const float* points, const float* conf, float* out, int32_t B, int32_t NV,
                       int32_t J, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Chain of up to LT_PWCHAIN_MAX pointwise (1x1x1) convolutions evaluated per voxel without the
 * intermediate activations leaving the registers:  y = L_n(...L_1(x)),
 * L_i(v) = act_i((W_i v + bias_i) * scale_i + shift_i)  (same epilogue constants as lt_conv_fwd).
 * Replaces the tail of V2VModel.forward (mvn/models/v2v.py:157-169: back_layers[1:] =
 * Basic3DBlock(32,32,1) x 2, then output_layer = Conv3d(32, J, 1)): one pass over the volume
 * instead of three.  bf16 activations, 32 input channels, inner widths 32, last width <= 32;
 * intermediate activations are rounded to bf16 exactly where separate launches would store them;
 * the last layer stores fp32: plane == 0 -> rows of ldy == cout[last] floats (channels-last);
 * plane > 0 -> planar, y[((row / plane) * cout[last] + c) * plane + row % plane] with plane = voxels
 * per sample (a multiple of 64 that divides rows) -- the (N, J, V, V, V) layout of the reference's
 * volumes, which is what lt_softargmax3d_fwd streams fastest.  x: rows x 32 bf16 (channels-last
 * volume), rows % 64 == 0.  weight[i]: the lt_conv_fwd packing [32][k_pad[i]] (k = input channel).
 * -------------------------------------------------------------------------------------------*/
#define LT_PWCHAIN_MAX 3
typedef struct {
    int32_t dtype;                       /* LT_BF16 */
    int32_t nlayers;                     /* 1..LT_PWCHAIN_MAX */
    int64_t rows;                        /* voxels */
    int32_t cin;                         /* 32 */
    int32_t ldy;                         /* == cout[nlayers-1] */
    int32_t cout[LT_PWCHAIN_MAX];
    int32_t k_pad[LT_PWCHAIN_MAX];
    int32_t flags[LT_PWCHAIN_MAX];       /* LT_EPI_RELU_POST; LT_EPI_STORE_F32 on the last layer (required) */
    const void* weight[LT_PWCHAIN_MAX];
    const float* bias[LT_PWCHAIN_MAX];   /* 32 entries each (padded), or NULL */
    const float* scale[LT_PWCHAIN_MAX];
    const float* shift[LT_PWCHAIN_MAX];
    int64_t plane;                       /* 0: channels-last output; > 0: planar output, voxels per sample */
} lt_pwchain_desc;
int lt_pwchain_fwd(const lt_pwchain_desc* desc, const void* x, void* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The stem of the 2D backbone in one pass: conv 7x7 / stride 2 / pad 3 -> (acc + bias) * scale +
 * shift (eval BatchNorm folded) -> ReLU -> max pool 3x3 / stride 2 / pad 1.  Replaces the first four
 * lines of PoseResNet.forward (mvn/models/pose_resnet.py:293-297: conv1, bn1, relu, maxpool): the
 * half-resolution 64-channel map never goes to HBM.  bf16 only (fp32 plans record lt_conv_fwd +
 * lt_maxpool_fwd); results are identical to those two launches (max commutes with the bf16 rounding).
 * x: N,H,W,8 channels-last (3 image channels zero-padded to 8); y: N,Hp,Wp,64 with Hc = (H-1)/2 + 1,
 * Hp = (Hc-1)/2 + 1.  weight: lt_stem_packed_bytes() bytes filled ONCE per model by
 * lt_stem_pack_weights from the lt_conv_fwd packing [64][k_pad] (k = (kh*7 + kw)*8 + ci, bf16): the
 * kernel's MFMA fragment order, so that a wave reads 1 KB contiguous per K step.
 * -------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t dtype;                       /* LT_BF16 */
    int32_t N, H, W;
    int32_t Cin, Cout;                   /* 8, 64 */
    const void* weight;                  /* packed by lt_stem_pack_weights, 16-byte aligned */
    const float* bias;                   /* 64 entries each, or NULL */
    const float* scale;
    const float* shift;
    int32_t x_layout;                    /* 0: x = N,H,W,8 bf16 (Cin = 8); 1: x = N,3,H,W fp32, the reference's image
                                            tensor (Cin = 3), rounded to bf16 on the fly like lt_nchw_to_nhwc would */
} lt_stem_desc;
size_t lt_stem_packed_bytes(void);
int lt_stem_pack_weights(const void* weight, int32_t k_pad, void* packed, void* stream);
int lt_stem_pool_fwd(const lt_stem_desc* desc, const void* x, void* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A whole identity Bottleneck block in one launch: Bottleneck.forward of the reference
 * (mvn/models/pose_resnet.py:75-95 with downsample == None and stride 1):
 *   t1 = relu(bn1(conv1x1(x)));  t2 = relu(bn2(conv3x3(t1), pad 1));  y = relu(bn3(conv1x1(t2)) + x)
 * with eval-mode BatchNorm folded like lt_conv_fwd's epilogue ((acc + bias) * scale + shift).  The two
 * intermediate tensors never leave LDS (they are rounded to bf16 exactly where the three separate launches
 * would store them); x is read once (plus the 1-pixel halo of an 8 x 16 pixel tile) and y written once.
 * x, y: N,H,W,C channels-last bf16, y != x;  (C, P) = (256, 64) or (512, 128) (ResNet layer1 / layer2);
 * H % 8 == 0, W % 16 == 0.  weight[i]: lt_conv_pack_weights_t32 of the lt_conv_fwd packing of layer i
 * (i = 0: [P][C], ntaps 1, cin C;  1: [P][9 P], ntaps 9, cin P;  2: [C][P], ntaps 1, cin P);
 * bias[i] may be NULL; scale / shift hold the layer's output-channel count of floats.
 * -------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t dtype;                       /* LT_BF16 */
    int32_t N, H, W;
    int32_t C, P;                        /* block width, bottleneck width (C == 4 P) */
    const void* weight[3];
    const float* bias[3];
    const float* scale[3];
    const float* shift[3];
} lt_bneck_desc;
int lt_bottleneck_fwd(const lt_bneck_desc* desc, const void* x, void* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The FIRST Bottleneck of ResNet layer1 -- the one with a `downsample` branch and stride 1 -- in one launch
 * (mvn/models/pose_resnet.py:75-95 with downsample = conv1x1 + BatchNorm, built at :196-206):
 *   t1 = relu(bn1(conv1x1(x)));  t2 = relu(bn2(conv3x3(t1), pad 1));  y = relu(bn3(conv1x1(t2)) + bn_d(conv1x1_d(x)))
 * Like lt_bottleneck_fwd the two inner tensors stay in LDS; here the residual is not read but COMPUTED from the
 * tile of x that is in LDS already (a second K = Cin GEMM of the output phase), and it is added in fp32 -- the four
 * separate launches round the downsample branch to bf16 first.  x: N,H,W,Cin, y: N,H,W,C channels-last bf16;
 * (Cin, P, C) = (64, 64, 256);  H % 8 == 0, W % 16 == 0.  weight[i]: lt_conv_pack_weights_t32 of the lt_conv_fwd
 * packing of conv1 [P][Cin], conv2 [P][9 P], conv3 [C][P], downsample [C][Cin];  scale / shift [i]: the folded
 * BatchNorm of that layer (ResNet convolutions carry no bias).
 * -------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t dtype;                       /* LT_BF16 */
    int32_t N, H, W;
    int32_t Cin, P, C;                   /* input width, bottleneck width, block width */
    const void* weight[4];               /* conv1, conv2, conv3, downsample */
    const float* scale[4];
    const float* shift[4];
} lt_bneck_ds_desc;
int lt_bottleneck_ds_fwd(const lt_bneck_ds_desc* desc, const void* x, void* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The 1x1 EXPAND of one identity Bottleneck block and the 1x1 REDUCE of the next one in one launch
 * (mvn/models/pose_resnet.py:75-95, the seam between two consecutive blocks of a ResNet stage):
 *   y  = relu(bn3(conv1x1_expand(t2)) + residual)          C channels, written once
 *   t1 = relu(bn1'(conv1x1_reduce'(y)))                    P channels, the next block's first activation
 * The reduce consumes y from LDS while the tile is there: the C-channel tensor is not read back from memory.
 * y is rounded to bf16 exactly where the expand's own launch would store it (the reduce sees those values).
 * t2 [M][P], residual / y [M][C], t1 [M][P]: channels-last bf16, M = N * H * W GEMM rows (any M: tiles of
 * 96 / 64 / 32 rows, chosen by M against the device's CU count -- 96 from 7/8 of a round of 96-row tiles on,
 * 64 from 7/16; LT_XR_NPB=3|2|1 forces one -- and a ragged last tile as a second one-workgroup launch with
 * row checks);  (C, P) = (1024, 256): the identity blocks of ResNet layer3.
 * weight[0]: lt_conv_pack_weights_t32 of the expand's lt_conv_fwd packing [C][P] (ntaps 1, cin P);
 * weight[1]: the same of the reduce's [P][C] (ntaps 1, cin C);  scale / shift [i]: the folded BatchNorm of
 * layer i (C / P floats), applied as acc * scale + shift (ResNet convolutions carry no bias).
 * Outputs must not alias the inputs.
 * -------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t dtype;                       /* LT_BF16 */
    int32_t C, P;                        /* block width, bottleneck width (C == 4 P) */
    int64_t M;                           /* GEMM rows */
    const void* weight[2];
    const float* scale[2];
    const float* shift[2];
    const float* consts;                 /* optional (may be NULL): scale[0] | shift[0] | scale[1] | shift[1] back to back (2 C + 2 P floats,
                                            16-byte aligned): the kernel then fetches the four tables with one LDS-DMA instead of four dependent loads */
} lt_xr_desc;
int lt_expand_reduce_fwd(const lt_xr_desc* desc, const void* t2, const void* residual, void* y, void* t1_next, void* stream);

/* ---------------------------------------------------------------------------------------------
 * hipGraph + event helpers (the forward is ~230 launches: replay it as one graph)
 * -------------------------------------------------------------------------------------------*/
int lt_graph_begin(void* stream);
int lt_graph_end(void* stream, void** graph_exec_out);
int lt_graph_launch(void* graph_exec, void* stream);
int lt_graph_destroy(void* graph_exec);
int lt_event_create(void** ev_out);
int lt_event_record(void* ev, void* stream);
int lt_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* synchronises on ev_stop */
int lt_event_destroy(void* ev);

/* ---------------------------------------------------------------------------------------------
 * PLAN-LEVEL entry points ("whole-network plan entry points that own pre-packed weights and a hipGraph"; round 6).
 * Everything the Python host does between the reference's module API and the kernel-level calls above lives behind these four symbols, so that a
 * host in any language can run the timed forward: layer -> kernel selection with every batch threshold (fused stem, whole-bottleneck launches, the
 * layer3 seam kernel, conv_cat2, the 2D / 3D halo kernels by weight layout, split-K of the tiny V2V levels, the pointwise tail), the eval-BatchNorm fold
 * as ATen evaluates it, weight packing (GEMM layout, MFMA fragment orders, parity phases of the stride-2 transposed convolutions), buffer reuse, the
 * fp64 camera algebra of the reference's forward, and the captured hipGraph.
 *
 * lt_plan_create_vol  = VolumetricTriangulationNet.__init__ + load_state_dict + the first forward's plan recording
 *                       (mvn/models/triangulation.py:204-242, mvn/models/pose_resnet.py:177-377, mvn/models/v2v.py:142-180).
 *   weights: the reference's state_dict -- names as torch writes them ("backbone.layer3.0.conv1.weight", "volume_net.front_layers.0.block.0.bias", ...;
 *   a leading "module." is stripped like train.py:408-410), HOST fp32 arrays in torch's layout; keys the volumetric path does not use
 *   (backbone.final_layer.*, num_batches_tracked) are ignored; a missing key is LT_ERR_INVALID with its name.  The arrays are read during this call only.
 *   One plan = one input shape (B, NV, H, W) and one element type: LT_F32 (exact-fp32 MFMA: the kernel set that meets the 1e-4 joint tolerance) or
 *   LT_BF16 (the throughput set).  Allocates device memory (hipMalloc) that lt_plan_destroy frees.
 * lt_plan_forward_vol = VolumetricTriangulationNet.forward (triangulation.py:245-355) for inference (eval-mode BatchNorm).
 *   images: DEVICE (B, NV, 3, H, W) fp32, the reference's input tensor.  Cameras: HOST fp64 K (B, NV, 3, 3) at IMAGE resolution, R (B, NV, 3, 3),
 *   t (B, NV, 3) -- batch['cameras'][view][sample] of the reference (datasets/utils.py:26), sample-major here; the call rescales K to the heatmap
 *   resolution and forms K [R | t] in fp64 like Camera.update_after_resize / .projection (multiview.py:33-52).  base_points: HOST (B, 3) fp64, the
 *   pelvis the cuboid is centred on (triangulation.py:284-296: batch['pred_keypoints_3d'][b][6] for kind 'mpii').  rot: HOST (B, 9) fp64 row-major
 *   cuboid rotations (triangulation.py:318-328) or NULL = identity (what eval mode draws).
 *   Outputs, DEVICE fp32, written on `stream`: keypoints_3d (B, J, 3) [required]; volumes (B, J, V, V, V) softmaxed / ReLU'd (NULL: kept in a plan
 *   buffer); features (B, NV, 32, h, w) or NULL; coord_volumes (B, V, V, V, 3) or NULL; vol_confidences (B, NV, 32) RAW head outputs or NULL (only for
 *   LT_AGG_CONF / LT_AGG_CONF_NORM plans; the reference returns them divided by their sum over views for 'conf_norm').  cuboids / base_points of the
 *   reference's 7-tuple are host values the caller already has.  Asynchronous; the first call captures the hipGraph (use_graph), later calls replay it.
 *   stream == NULL with use_graph: the legacy default stream cannot be captured, so the forward runs on a stream the plan owns, ordered behind the default
 *   stream's earlier work and in front of its later work by events.  Not thread-safe per plan (one forward of a plan at a time).
 * -------------------------------------------------------------------------------------------*/
typedef struct lt_plan lt_plan;
typedef struct lt_named_tensor {
    const char* name;        /* state_dict key */
    const float* data;       /* HOST fp32, contiguous, torch's layout (Conv: [Cout][Cin][k..]; ConvTranspose: [Cin][Cout][k..]; Linear: [out][in]) */
    int32_t ndim;            /* 1 .. 5 */
    int64_t shape[5];
} lt_named_tensor;
typedef struct lt_vol_plan_config {
    int32_t dtype;                        /* LT_F32 | LT_BF16 */
    int32_t num_layers;                   /* config.model.backbone.num_layers: 18 | 34 | 50 | 101 | 152 */
    int32_t style_caffe;                  /* config.model.backbone.style == "caffe" */
    int32_t num_joints;                   /* 17 */
    int32_t B, NV, H, W;                  /* samples, views per sample, image size */
    int32_t volume_size;                  /* config.model.volume_size (64) */
    double cuboid_side;                   /* config.model.cuboid_side (2500.0 mm) */
    double volume_multiplier;             /* config.model.volume_multiplier */
    int32_t volume_softmax;               /* config.model.volume_softmax */
    int32_t aggregation;                  /* LT_AGG_* of config.model.volume_aggregation_method */
    int32_t transfer_cmu_to_human36m;     /* config.model.transfer_cmu_to_human36m (triangulation.py:336-339) */
    int32_t use_graph;                    /* capture the forward into a hipGraph at the first call and replay it */
} lt_vol_plan_config;
typedef struct lt_plan_info_t {
    int32_t launches;                     /* ops of one forward (pre + captured + tail) */
    int32_t heatmap_h, heatmap_w;         /* h, w of the feature maps (H / 4, W / 4) */
    double flops;                         /* 2 * MAC of the recorded convolutions */
    int64_t bytes_allocated;              /* device memory owned by the plan */
    int32_t n_expand_reduce, n_bottleneck, n_bottleneck_ds, n_conv_cat2, n_conv2d_halo, n_pwchain, n_stem_pool, n_splitk, n_conv_skip;   /* which fused kernels it recorded */
    int32_t graph_captured;
    const float* logits;                  /* V2V logits (fp32; planar (B, J, V, V, V) when logits_planar, else channels-last (B, V, V, V, J)): valid after a forward, for tests */
    int32_t logits_planar;
} lt_plan_info_t;
int lt_plan_create_vol(const lt_vol_plan_config* cfg, const lt_named_tensor* weights, int32_t nweights, lt_plan** plan_out);
int lt_plan_forward_vol(lt_plan* plan, const float* images, const double* K_host, const double* R_host, const double* t_host, const double* base_points_host,
                        const double* rot_host, float* keypoints_3d, float* volumes, float* features, float* coord_volumes, float* vol_confidences, void* stream);
int lt_plan_info(const lt_plan* plan, lt_plan_info_t* info);
void lt_plan_destroy(lt_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* LT_HIP_H */
