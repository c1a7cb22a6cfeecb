"""Host-side execution engine: turns the reference-shaped nn.Module trees (parameter containers only)
into a flat list of liblt_hip launches over preallocated channels-last buffers, captured once per
input shape into a hipGraph and replayed.

Everything numerical happens in liblt_hip.so.  This module only
  * re-packs weights once per plan (k = tap*Cin + ci, zero padded), folds eval-mode BatchNorm and the
    conv bias into per-channel scale/shift, and splits stride-2 transposed convs into parity phases;
  * allocates activations (torch is the device allocator) with size-keyed reuse;
  * issues the C-ABI calls on an explicit stream and owns the hipGraph.
"""
import ctypes as C
import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

import lt_hip as H

BN_EPS = 1e-5


def cout_pad_of(cout):
    """lt_conv_cout_pad (include/lt_hip.h) restated so that specs can be built without the library."""
    if cout <= 16:
        return 16
    if cout <= 32:
        return 32
    if cout <= 64:
        return 64
    return (cout + 127) // 128 * 128


def conv_chunk_samples(N, per_sample):
    """lt_conv_chunk_samples (include/lt_hip.h) restated: samples per launch of a convolution over tensors of ``per_sample`` elements per sample -- all of
    N while N * per_sample < 2^31, else equal chunks (multiples of 8 where possible) walked by the C entry point with 64-bit base pointers."""
    lim = (1 << 31) - 1
    if N < 1 or per_sample < 1 or per_sample > lim:
        return 0
    if N * per_sample <= lim:
        return N
    nmax = lim // per_sample
    if nmax >= 8:
        nmax &= ~7
    chunks = -(-N // nmax)
    nc = -(-N // chunks)
    if nc > 8 and (nc & 7) and ((nc + 7) & ~7) <= nmax:
        nc = (nc + 7) & ~7
    return nc


def k_step_of(dtype):
    return 128 // torch.empty((), dtype=dtype).element_size()  # elements per 128-byte K step: 32 fp32 / 64 bf16 / 128 fp8


def min_cin_of(dtype):
    return 16 // torch.empty((), dtype=dtype).element_size()   # one 16-byte vector per pixel


@dataclass
class ConvPhaseSpec:
    weight: torch.Tensor            # fp32 [cout_pad, k_pad]
    taps: torch.Tensor              # int32 [ntaps, 4] = dd, dh, dw, element offset
    out_off: Tuple[int, int, int]


@dataclass
class ConvSpec:
    N: int; D: int; H: int; W: int; Cin: int
    Do: int; Ho: int; Wo: int
    stride: Tuple[int, int, int]; pad: Tuple[int, int, int]
    OD: int; OH: int; OW: int
    out_stride: Tuple[int, int, int]
    Cout: int; cout_pad: int; k_pad: int
    flags: int
    bias: torch.Tensor              # fp32 [cout_pad]   y = (acc + bias) * scale + shift
    scale: torch.Tensor             # fp32 [cout_pad]
    shift: torch.Tensor             # fp32 [cout_pad]
    phases: List[ConvPhaseSpec] = field(default_factory=list)


class BnParams(tuple):
    """(weight, bias, running_mean, running_var) of a BatchNorm module -- unpacks like the plain 4-tuple the plan code takes -- that also
    remembers whether the module is in training mode (``.training``): the training tape normalises with batch statistics there and with the
    FROZEN running statistics for a module left in eval() (fine-tuning with a frozen backbone)."""

    def __new__(cls, bn):
        self = super().__new__(cls, (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        self.training = bool(bn.training)
        return self


def fold_bn(cout, bias, bn, cout_pad):
    """Epilogue constants of y = (acc + bias) * scale + shift, as fp32 arrays of cout_pad entries.

    Eval-mode BatchNorm is folded exactly the way ATen evaluates it (batch_norm_cpu_collect_linear_and_constant_terms):
    fp32 invstd = 1/sqrt(var + eps), scale = invstd * weight, shift = bias_bn - mean * scale -- so the per-channel
    constants carry the same roundings as the reference's.  bn = (gamma, beta, running_mean, running_var) or None.
    """
    bi = torch.zeros(cout_pad, dtype=torch.float32)
    sc = torch.ones(cout_pad, dtype=torch.float32)
    sh = torch.zeros(cout_pad, dtype=torch.float32)
    if bias is not None:
        bi[:cout] = bias.detach().float().cpu()
    if bn is not None:
        # numpy fp32, not torch: ATen's eval BatchNorm computes 1 / std::sqrt(var + eps) per channel with the scalar (correctly rounded) square root, and so does
        # the C host of the plan-level ABI (csrc/plan.hip: fold_bn); torch's VECTORISED CPU sqrt differs from the IEEE result in ~0.5 % of the elements
        # (measured, round 6), which made the two hosts' plans differ in the last bit of some scales.  numpy's float32 sqrt / divide / multiply are IEEE.
        g, b, m, v = (t.detach().float().cpu().numpy() for t in bn)
        invstd = (np.float32(1.0) / np.sqrt(v + np.float32(BN_EPS))).astype(np.float32)
        alpha = (invstd * g).astype(np.float32)
        sc[:cout] = torch.from_numpy(alpha)
        sh[:cout] = torch.from_numpy((b - (m * alpha).astype(np.float32)).astype(np.float32))
    return bi, sc, sh


def _pad_k(wk, cout_pad, k_pad):
    out = torch.zeros(cout_pad, k_pad, dtype=torch.float32)
    out[:wk.shape[0], :wk.shape[1]] = wk
    return out


def make_conv_spec(weight, bias, bn, in_shape, stride, pad, dtype, transposed=False, flags=0, output_padding=0):
    """Build the lt_conv_fwd description of one (transposed) convolution layer.

    weight: Conv{2,3}d [Cout,Cin,*k] or ConvTranspose{2,3}d [Cin,Cout,*k] (stride 2 only);
    in_shape: (N, D, H, W, Cin_buffer) of the channels-last input (D = 1 for 2D; Cin_buffer >= Cin,
    extra input channels get zero weights); stride/pad: ints or 3-tuples (d,h,w).
    """
    w = weight.detach().float().cpu()
    if w.dim() == 4:
        w = w.unsqueeze(2)
    N, D, Hh, W, cin_buf = in_shape
    st = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
    pd = (pad,) * 3 if isinstance(pad, int) else tuple(pad)
    if weight.dim() == 4:
        st = (1, st[1], st[2]); pd = (0, pd[1], pd[2])
    kstep = k_step_of(dtype)
    if not transposed:
        cout, cin, kd, kh, kw = w.shape
        assert cin <= cin_buf
        if cin < cin_buf:
            w = torch.cat([w, torch.zeros(cout, cin_buf - cin, kd, kh, kw)], dim=1)
        Do = (D + 2 * pd[0] - kd) // st[0] + 1
        Ho = (Hh + 2 * pd[1] - kh) // st[1] + 1
        Wo = (W + 2 * pd[2] - kw) // st[2] + 1
        cp = cout_pad_of(cout)
        K = kd * kh * kw * cin_buf
        k_pad = (K + kstep - 1) // kstep * kstep
        wk = w.permute(0, 2, 3, 4, 1).reshape(cout, K)
        taps = [(a, b, c, ((a * Hh + b) * W + c) * cin_buf) for a in range(kd) for b in range(kh) for c in range(kw)]
        bi, sc, sh = fold_bn(cout, bias, bn, cp)
        spec = ConvSpec(N, D, Hh, W, cin_buf, Do, Ho, Wo, st, pd, Do, Ho, Wo, (1, 1, 1), cout, cp, k_pad, flags, bi, sc, sh)
        spec.phases.append(ConvPhaseSpec(_pad_k(wk, cp, k_pad), torch.tensor(taps, dtype=torch.int32).reshape(-1, 4), (0, 0, 0)))
        return spec
    # ---- stride-2 transposed conv: one phase per output parity --------------------------------
    cin, cout, kd, kh, kw = w.shape
    assert cin == cin_buf, "transposed conv input may not be channel padded"
    nd3 = weight.dim() == 5
    assert all(s == 2 for s in (st if nd3 else st[1:])), "only stride-2 transposed convolutions"
    ks = (kd, kh, kw)
    dims = (D, Hh, W)
    outs = []
    for i in range(3):
        if not nd3 and i == 0:
            outs.append(1)
        else:
            o = (dims[i] - 1) * 2 - 2 * pd[i] + ks[i] + output_padding
            assert o == 2 * dims[i], "transposed conv must exactly double the size (k=4,p=1 / k=2,p=0; with output_padding=1: k=3,p=1 / k=1,p=0 -- the input gradients of the stride-2 convolutions)"
            outs.append(o)
    cp = cout_pad_of(cout)

    def dim_phases(i):
        if not nd3 and i == 0:
            return [(0, [(0, 0)])]
        res = []
        for phi in (0, 1):  # o = 2q + phi = 2*i_in - p + kk  ->  kk = phi + p (mod 2), i_in = q + (phi + p - kk)/2
            res.append((phi, [(kk, (phi + pd[i] - kk) // 2) for kk in range(ks[i]) if (phi + pd[i] - kk) % 2 == 0]))
        return res

    phase_list = []
    ntaps_max = 0
    for pa, ta in dim_phases(0):
        for pb, tb in dim_phases(1):
            for pc, tc in dim_phases(2):
                taps, cols = [], []
                for (ka, da) in ta:
                    for (kb, db) in tb:
                        for (kc, dc) in tc:
                            taps.append((da, db, dc, ((da * Hh + db) * W + dc) * cin))
                            cols.append(w[:, :, ka, kb, kc].t())  # [cout, cin]
                if not taps:      # an output parity no tap reaches (1x1 / stride 2: the odd positions): one tap with zero weights writes the zeros
                    taps, cols = [(0, 0, 0, 0)], [torch.zeros(cout, cin)]
                wk = torch.cat(cols, dim=1)
                phase_list.append((wk, taps, (pa, pb, pc)))
                ntaps_max = max(ntaps_max, len(taps))
    K = ntaps_max * cin
    k_pad = (K + kstep - 1) // kstep * kstep
    bi, sc, sh = fold_bn(cout, bias, bn, cp)
    ostr = (2 if nd3 else 1, 2, 2)
    spec = ConvSpec(N, D, Hh, W, cin, D, Hh, W, (1, 1, 1), (0, 0, 0), outs[0], outs[1], outs[2], ostr, cout, cp, k_pad, flags, bi, sc, sh)
    for wk, taps, off in phase_list:
        spec.phases.append(ConvPhaseSpec(_pad_k(wk, cp, k_pad), torch.tensor(taps, dtype=torch.int32).reshape(-1, 4), off))
    assert len(spec.phases) <= H.MAX_PHASES
    return spec


class Act:
    """A channels-last activation: tensor [N, D, H, W, C] (D == 1 for 2D maps)."""
    __slots__ = ("t", "pooled")

    def __init__(self, t):
        self.t = t
        self.pooled = True

    @property
    def shape(self):
        return tuple(self.t.shape)


class PlanBuilder:
    """Records liblt_hip launches; buffers are reused by exact byte size once released."""

    def __init__(self, device, dtype, tile_override=0, dry_run=False, stages=0):
        """dry_run=True records the plan over CPU tensors WITHOUT being able to execute it (Plan.run raises): the CPU
        test-suite uses it to check the recorded wiring / buffer reuse with an interpreter that lives in tests/."""
        self.device = torch.device(device)
        self.dry_run = dry_run
        if self.device.type != "cuda" and not dry_run:
            raise RuntimeError("liblt_hip plans run on the GPU only (device=%s); there is no CPU fallback" % device)
        if not dry_run:
            H.lib()
        self.dtype = dtype
        self.code = H.dtype_code(dtype)
        # element type of outputs / residuals that are not fp32: the plan's own, except that fp8 (an OPERAND type of lt_conv_fwd only) stores bf16
        self.out_dtype = torch.bfloat16 if dtype == torch.float8_e4m3fn else dtype
        self.ops = []          # (callable, args)
        self.keep = []         # tensors / ctypes objects that must outlive the plan
        self.pool = {}         # nbytes -> [tensor]
        self.tile_override = tile_override
        self.stages = stages
        # lt_train.TrainTape sets this on its bf16 builder: the weights of its convolutions are re-gathered from the live Parameters every step, so no
        # copy in MFMA fragment order may be packed at build time (a kernel reading one would compute with the build-time values), and no split-K
        # pair (two ops per convolution) is recorded
        self.live_weights = False
        # round 6: with live weights, STILL pack the fragment-order copies (from whatever the build-time tensor holds) and hand each one out as
        # info["wfrag"] = [(phase, fragment tensor, pack(src_ptr, dst_ptr))]: the tape composes the pack's permutation with its own index map and gathers
        # the live Parameter straight into the fragment layout every step, so the training forward / input gradients run the kernels of the inference
        # forward (conv2d_halo, conv_igemm6/7, conv3d_halo_wreg) instead of the generic tiles
        self.live_frag = False
        self.flops = 0         # 2*MAC of the recorded convolutions
        self.bytes_alloc = 0
        self.ntail = 0         # trailing ops kept out of the captured graph (PlanBuilder.custom(tail=True))
        self.npre = 0          # leading ops kept out of the captured graph (they read a caller-owned tensor: stem_pool(image_cell=...))

    # ---- memory ---------------------------------------------------------------------------
    def alloc(self, shape, dtype=None):
        dtype = dtype or self.dtype
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        lst = self.pool.get((n, dtype))
        if lst:
            t = lst.pop().view(*shape)
        else:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.bytes_alloc += n
        self.keep.append(t)
        return Act(t)

    def release(self, act):
        """Hand a dead activation's storage back (stream order makes the reuse safe)."""
        if act is None or not act.pooled:
            return
        t = act.t
        n = t.numel() * t.element_size()
        self.pool.setdefault((n, t.dtype), []).append(t.reshape(-1))

    def const(self, t, dtype=None):
        t = t.to(device=self.device, dtype=dtype or t.dtype).contiguous()
        self.keep.append(t)
        return t

    # ---- ops ------------------------------------------------------------------------------
    def _add(self, fn, kind="op", label="", flops=0, nbytes=0, info=None):
        meta = {"kind": kind, "label": label, "flops": flops, "bytes": nbytes}
        if self.dry_run:
            meta["info"] = info or {}
        self.last_info = info or {}        # lt_train.TrainTape reads the device buffers of the op it has just recorded
        self.ops.append((fn, meta))

    def can_conv_skip(self, x_shape, weight, skip_shape, skip_weight):
        """True when lt_conv_skip_fwd covers this convolution + computed residual: the second 3x3x3 convolution (32 -> 32) of a Res3DBlock whose skip
        connection is a 1x1x1 convolution of a 16-channel tensor (v2v.py:20-42, :76), bf16 plan over build-time weights, on a shape the column-walk halo
        kernel takes -- the conditions of conv3d_halo_try, mirrored (the C side has no fallback and fails loudly).  LT_NO_CONV_SKIP=1: off (A/B)."""
        if (self.dtype != torch.bfloat16 or self.out_dtype != torch.bfloat16 or self.live_weights or self.tile_override or
                any(os.environ.get(k) for k in ("LT_NO_CONV_SKIP", "LT_HALO_NO_COL", "LT_HALO_NO_PERSIST", "LT_CONV_NO_HALO"))):
            return False
        if tuple(weight.shape) != (32, 32, 3, 3, 3) or tuple(skip_weight.shape) != (32, 16, 1, 1, 1):
            return False
        N, D, Hh, W, Cin = x_shape
        if Cin != 32 or tuple(skip_shape) != (N, D, Hh, W, 16) or D % 4 or Hh % 8 or W % 8 or D // 4 < 2:
            return False
        # batches beyond 2^31 elements run as sample chunks inside the C entry point (round 6): EVERY chunk has to satisfy the kernel's conditions
        nc = conv_chunk_samples(N, D * Hh * W * 32)
        if nc < 1:
            return False
        for n in {nc, N - (N - 1) // nc * nc}:          # the full chunks and the last one
            nblk, cols = n * (D // 4) * (Hh // 8) * (W // 8), n * (Hh // 8) * (W // 8)
            if not (nblk >= 1024 and nblk % 8 == 0 and cols % 8 == 0 and cols >= 256):
                return False
        return True

    def conv(self, x, weight, bias=None, bn=None, stride=1, pad=0, transposed=False, relu=False, relu_pre=False,
             residual=None, out_f32=False, out=None, sigmoid=False, output_padding=0, residual_f32=False, skip=None):
        """x: Act.  Returns the output Act [N, OD, OH, OW, Cout].  ``residual_f32`` (bf16 plans, with ``out_f32``): the residual is an fp32 tensor
        (LT_EPI_RES_F32: the training tape's input-gradient accumulation).  ``skip`` = (Act, weight, bias, bn) of a 1x1x1 skip convolution whose output
        is this convolution's residual (lt_conv_skip_fwd; only where ``can_conv_skip`` says so): the residual is computed inside the launch."""
        flags = ((H.EPI_RELU_POST if relu else 0) | (H.EPI_RELU_PRE if relu_pre else 0) | (H.EPI_STORE_F32 if out_f32 else 0)
                 | (H.EPI_SIGMOID if sigmoid else 0) | (H.EPI_RES_F32 if residual_f32 else 0))
        assert not residual_f32 or (out_f32 and residual is not None and self.dtype in (torch.bfloat16, torch.float8_e4m3fn))
        spec = make_conv_spec(weight, bias, bn, x.shape, stride, pad, self.dtype, transposed, flags, output_padding)
        skip_info = None
        if skip is not None:
            sx, sw, sb, sbn = skip
            assert residual is None and not relu_pre and not out_f32 and out is None and self.can_conv_skip(x.shape, weight, sx.shape, sw)
            ss = make_conv_spec(sw, sb, sbn, sx.shape, 1, 0, self.dtype, False, 0)
            assert ss.cout_pad == spec.cout_pad == 32
            # the skip branch's BatchNorm: scale into its weights (fp32 product, ONE bf16 rounding), (bias * scale + shift) into this convolution's shift
            wfold = (ss.phases[0].weight * ss.scale[:, None]).to(self.dtype)
            spec.shift = spec.shift + ss.bias * ss.scale + ss.shift
            wsk = self.const(wfold, self.dtype)
            wfr = torch.empty(32 * 16, dtype=self.dtype, device=wsk.device)
            if not self.dry_run:
                H.check(H.lib().lt_conv_pack_weights_t32(wsk.data_ptr(), 32, ss.k_pad, 16, 1, wfr.data_ptr(), H.cur_stream()), "lt_conv_pack_weights_t32")
            sk = H.ConvSkip()
            sk.x, sk.cin, sk.weight_frag = sx.t.data_ptr(), 16, wfr.data_ptr()
            self.keep += [wfr, sx.t, sk]
            skip_info = {"x": sx, "w": wfold[:, :16].float(), "desc": sk}
        S = self.splitk_slices(spec, weight, transposed, out_f32, sigmoid, out)
        if S > 1:
            return self._conv_splitk(x, weight, spec, S, residual)
        y = out or self.alloc((spec.N, spec.OD, spec.OH, spec.OW, spec.Cout), torch.float32 if out_f32 else self.out_dtype)
        self.keep.append(x.t)   # the launch closure holds raw pointers only
        if residual is not None:
            self.keep.append(residual.t)
        if residual is not None:
            assert residual.shape == y.shape and residual.t.dtype == (torch.float32 if residual_f32 else self.out_dtype), (residual.shape, y.shape)
        d = H.ConvDesc()
        d.dtype = self.code
        d.N, d.D, d.H, d.W, d.Cin = spec.N, spec.D, spec.H, spec.W, spec.Cin
        d.Do, d.Ho, d.Wo = spec.Do, spec.Ho, spec.Wo
        d.stride = H.i3(spec.stride); d.pad = H.i3(spec.pad)
        d.OD, d.OH, d.OW = spec.OD, spec.OH, spec.OW
        d.out_stride = H.i3(spec.out_stride)
        d.Cout, d.ldc, d.cout_pad, d.k_pad = spec.Cout, spec.Cout, spec.cout_pad, spec.k_pad
        d.nphase, d.flags, d.tile, d.stages = len(spec.phases), spec.flags, self.tile_override, self.stages
        wdevs, wfrags = [], []
        for i, ph in enumerate(spec.phases):
            wdev = self.const(ph.weight, self.dtype)
            wdevs.append(wdev)
            tdev = self.const(ph.taps)
            d.phase[i].weight = wdev.data_ptr(); d.phase[i].taps = tdev.data_ptr()
            d.phase[i].ntaps = ph.taps.shape[0]; d.phase[i].out_off = H.i3(ph.out_off)
            # wide bf16 layers also get their weights in MFMA fragment order (B operand read straight from global memory by
            # the 288 x 256 kernel); packed once, here
            if self.live_weights and not self.live_frag:
                pass
            elif (self.dtype == torch.bfloat16 and not self.dry_run and x.shape[-1] == 256 and spec.D == 1 and spec.W % 24 == 0 and spec.H % 8 == 0
                  and residual is None and not out_f32 and not sigmoid and not self.tile_override and os.environ.get("LT_CONV_NO_H2D") != "1"
                  and spec.Cout == spec.cout_pad == 256 and d.ldc % 8 == 0 and not any(os.environ.get(k) for k in ("LT_CONV_V1",))
                  and (spec.N * (spec.H // 8) * (spec.W // 24) >= 60 or os.environ.get("LT_H2D_ANY_SIZE") == "1")
                  # conv2d_halo_try's predicate, mirrored (the dispatcher fails loudly on a layout-2 2D layer the kernel declines): a "same" 3x3 / stride 1 /
                  # pad 1, or the four 2 x 2-tap parities of a 4x4 / stride 2 / pad 1 / output_padding 0 transposed convolution that doubles the map
                  and ((not transposed and tuple(weight.shape) == (256, 256, 3, 3) and spec.stride == (1, 1, 1) and spec.pad == (0, 1, 1) and spec.W == 24
                        and (spec.OH, spec.OW) == (spec.Ho, spec.Wo) == (spec.H, spec.W) and spec.out_stride == (1, 1, 1)) or
                       (transposed and tuple(weight.shape) == (256, 256, 4, 4) and len(spec.phases) == 4 and os.environ.get("LT_DECONV_NO_H2D") != "1"
                        and output_padding == 0 and spec.out_stride == (1, 2, 2) and (spec.OH, spec.OW) == (2 * spec.H, 2 * spec.W)
                        and all(int(p.taps.shape[0]) == 4 for p in spec.phases)))):
                # ResNet layer3's 3x3 256 -> 256 on 24-wide maps and the 4x4 / stride-2 transposed convolutions 256 -> 256 of the head (four parities of
                # 2 x 2 taps), from 60 tiles of 8 x 24 pixels on (= 5 samples of 4 views; measured with the threshold off: 799.9 -> 811.3 samples/s at 5 samples,
                # 1145 -> 1172 at 10, 1406 -> 1428 at 32 -- a tile is a ~40 us serial chain, so a handful of them loses to the small implicit-GEMM tiles):
                # fragments of the transposed product for conv2d_halo_kernel (input halo resident in LDS; LT_CONV_NO_H2D=1 keeps conv_igemm7,
                # LT_DECONV_NO_H2D=1 only for the transposed ones)
                wfr = torch.empty_like(wdev)
                pack = lambda sp, dp, a=(spec.cout_pad, spec.k_pad, 256, int(ph.taps.shape[0])): H.check(
                    H.lib().lt_conv_pack_weights_t32(sp, a[0], a[1], a[2], a[3], dp, H.cur_stream()), "lt_conv_pack_weights_t32")
                pack(wdev.data_ptr(), wfr.data_ptr())
                self.keep.append(wfr)
                wfrags.append((i, wfr, pack))
                d.phase[i].weight_frag, d.phase[i].weight_frag_layout = wfr.data_ptr(), 2
            elif self.dtype == torch.bfloat16 and not self.dry_run and spec.cout_pad % 256 == 0 and spec.k_pad % 64 == 0:
                wfr = torch.empty_like(wdev)
                # the 288-row layers get their weights in the fragment order of the 32x32x16 MFMA (conv_igemm7: +1 % end to end over
                # conv_igemm6, 3x3 256->256 90.8 -> 87.4 us, 1x1 1024->256 50.9 -> 48.2 us inside the forward; LT_CONV_NO_V7=1 when the
                # plan is built keeps conv_igemm6); the short-K pointwise layers stay on the 144-row variant of conv_igemm6 and its
                # 16x16x32 order (measured: conv_igemm7 103.6 vs 85.6 us on 256->1024)
                short_pw = all(k == 1 for k in weight.shape[2:]) and spec.k_pad <= 256 and not transposed
                if os.environ.get("LT_CONV_NO_V7") != "1" and not short_pw:
                    pack = lambda sp, dp, a=(spec.cout_pad, spec.k_pad): H.check(H.lib().lt_conv_pack_weights32(sp, a[0], a[1], dp, H.cur_stream()),
                                                                                 "lt_conv_pack_weights32")
                    layout = 3
                else:
                    pack = lambda sp, dp, a=(spec.cout_pad, spec.k_pad): H.check(H.lib().lt_conv_pack_weights(sp, a[0], a[1], dp, H.cur_stream()),
                                                                                 "lt_conv_pack_weights")
                    layout = 1
                pack(wdev.data_ptr(), wfr.data_ptr())
                self.keep.append(wfr)
                wfrags.append((i, wfr, pack))
                d.phase[i].weight_frag, d.phase[i].weight_frag_layout = wfr.data_ptr(), layout
            elif (self.dtype == torch.bfloat16 and not self.dry_run and not transposed and x.shape[-1] == weight.shape[1]
                  and tuple(weight.shape) in ((64, 64, 3, 3, 3), (64, 32, 3, 3, 3), (128, 128, 3, 3, 3), (32, 16, 3, 3, 3))
                  and spec.stride == (1, 1, 1) and spec.pad == (1, 1, 1)):
                # 3x3x3 64 -> 64, 32 -> 64, 128 -> 128, 16 -> 32 (V2V): fragments of the transposed product for conv3d_halo_wreg_kernel
                wfr = torch.empty_like(wdev)
                pack = lambda sp, dp, a=(spec.cout_pad, spec.k_pad, int(weight.shape[1])): H.check(
                    H.lib().lt_conv_pack_weights_t32(sp, a[0], a[1], a[2], 27, dp, H.cur_stream()), "lt_conv_pack_weights_t32")
                pack(wdev.data_ptr(), wfr.data_ptr())
                self.keep.append(wfr)
                wfrags.append((i, wfr, pack))
                d.phase[i].weight_frag, d.phase[i].weight_frag_layout = wfr.data_ptr(), 2
        bi, sc, sh = self.const(spec.bias), self.const(spec.scale), self.const(spec.shift)
        self.keep.append(d)
        macs = spec.N * spec.Do * spec.Ho * spec.Wo * spec.Cout * sum(int(p.taps.shape[0]) for p in spec.phases) * (
            weight.shape[1] if not transposed else weight.shape[0])
        self.flops += 2 * macs
        lib = None if self.dry_run else H.lib()
        ksz = "x".join(str(k) for k in weight.shape[2:])
        label = "%s%s %d->%d @%s" % ("deconv" if transposed else "conv", ksz, spec.Cin, spec.Cout,
                                     "x".join(str(v) for v in (spec.N, spec.Do, spec.Ho, spec.Wo)))
        esz = torch.empty((), dtype=self.dtype).element_size()
        nbytes = (x.t.numel() + y.t.numel() + (residual.t.numel() if residual is not None else 0)) * esz + \
            sum(p.weight.numel() for p in spec.phases) * esz
        info = {"spec": spec, "x": x, "y": y, "res": residual, "wdev": wdevs, "bias_dev": bi, "scale_dev": sc, "shift_dev": sh, "desc": d, "wfrag": wfrags}
        if skip_info is not None:
            macs += spec.N * spec.Do * spec.Ho * spec.Wo * spec.Cout * 16
            self.flops += 2 * spec.N * spec.Do * spec.Ho * spec.Wo * spec.Cout * 16
            nbytes += skip_info["x"].t.numel() * esz
            info["skip"] = skip_info
            self._add(lambda s, d=d, xp=x.t.data_ptr(), bip=bi.data_ptr(), scp=sc.data_ptr(), shp=sh.data_ptr(), sk=skip_info["desc"], yp=y.t.data_ptr():
                      H.check(lib.lt_conv_skip_fwd(C.byref(d), xp, bip, scp, shp, C.byref(sk), yp, s), "lt_conv_skip_fwd"),
                      "conv", label + " + skip conv1x1x1 16->32", 2 * macs, nbytes, info)
            return y
        self._add(lambda s, d=d, xp=x.t.data_ptr(), bip=bi.data_ptr(), scp=sc.data_ptr(), shp=sh.data_ptr(),
                  rp=H.ptr(residual.t) if residual is not None else None, yp=y.t.data_ptr():
                  H.check(lib.lt_conv_fwd(C.byref(d), xp, bip, scp, shp, rp, yp, s), "lt_conv_fwd"),
                  "conv", label, 2 * macs, nbytes, info)
        return y

    # ---- last 1x1 convolution of a Bottleneck + its downsample branch as ONE pointwise convolution over two sources ---------------------------------
    def can_conv_cat2(self, t2_shape, w_expand, x_shape, w_down, stride_down):
        """True when lt_conv_cat2_fwd covers  relu(bn3(conv1x1(t2)) + bn_d(conv1x1_d(x), stride s)): bf16 plan over build-time weights, 2D maps, both
        channel counts multiples of 32 with a sum that is a multiple of 64, the block width a multiple of 256, s in (1, 2) and x's map exactly s times
        t2's (LT_NO_CONV_CAT2=1: off; needs conv_igemm7, so LT_CONV_NO_V7=1 turns it off too)."""
        if (self.dtype != torch.bfloat16 or self.out_dtype != torch.bfloat16 or self.live_weights or self.tile_override or
                os.environ.get("LT_NO_CONV_CAT2") == "1" or os.environ.get("LT_CONV_NO_V7") == "1" or os.environ.get("LT_CONV_NO_V3") == "1"):
            return False
        N, D, Ho, Wo, P = t2_shape
        if D != 1 or w_expand.dim() != 4 or w_down.dim() != 4 or tuple(w_expand.shape[2:]) != (1, 1) or tuple(w_down.shape[2:]) != (1, 1):
            return False
        Cc, Cin2 = w_expand.shape[0], w_down.shape[1]
        if w_expand.shape[1] != P or w_down.shape[0] != Cc or stride_down not in (1, 2) or tuple(x_shape) != (N, 1, Ho * stride_down, Wo * stride_down, Cin2):
            return False
        if P % 32 or Cin2 % 32 or (P + Cin2) % 64 or Cc % 256 or P & (P - 1):
            return False
        # one kernel with 288 x 256 tiles, one workgroup per CU: below ~200 tiles (lt_conv_fwd's own rule for that tile) the separate launches on
        # smaller tiles fill the chip better (LT_CAT2_ANY_SIZE=1: always -- tests)
        if -(-(N * Ho * Wo) // 288) * (Cc // 256) < 200 and os.environ.get("LT_CAT2_ANY_SIZE") != "1":
            return False
        return N * Ho * Wo * Cc < 2 ** 31 and N * x_shape[2] * x_shape[3] * Cin2 < 2 ** 31

    def conv_cat2(self, t2, w_expand, bn_expand, x, w_down, bn_down, stride_down):
        """relu(bn3(conv1x1(t2)) + bn_d(conv1x1_d(x), stride s)) as ONE launch (lt_conv_cat2_fwd): a pointwise convolution over the channel concatenation
        [t2 | x at the strided pixels] with weights [s3 * w3 | s_d * w_d] (the two BatchNorm scales folded into the weights: fp32 product, one bf16
        rounding) and shift = shift3 + shift_d.  The downsample's own launch, its output and the read of it as the residual disappear."""
        assert self.can_conv_cat2(t2.shape, w_expand, x.shape, w_down, stride_down)
        N, _, Ho, Wo, P = t2.shape
        Cc, Cin2 = w_expand.shape[0], w_down.shape[1]
        s3 = make_conv_spec(w_expand, None, bn_expand, t2.shape, 1, 0, self.dtype, False, H.EPI_RELU_POST)
        sd = make_conv_spec(w_down, None, bn_down, x.shape, stride_down, 0, self.dtype, False, 0)
        assert s3.cout_pad == sd.cout_pad == Cc and (s3.Do, s3.Ho, s3.Wo) == (sd.Do, sd.Ho, sd.Wo) == (1, Ho, Wo)
        wcat = torch.cat([s3.phases[0].weight[:, :P] * s3.scale[:, None], sd.phases[0].weight[:, :Cin2] * sd.scale[:, None]], dim=1).to(self.dtype)
        shift = (s3.bias * s3.scale + s3.shift) + (sd.bias * sd.scale + sd.shift)
        kp = P + Cin2
        spec = ConvSpec(N, 1, Ho, Wo, kp, 1, Ho, Wo, (1, 1, 1), (0, 0, 0), 1, Ho, Wo, (1, 1, 1), Cc, Cc, kp, H.EPI_RELU_POST,
                        torch.zeros(Cc), torch.ones(Cc), shift)          # the convolution over the concatenated tensor (what the plan interpreter evaluates)
        spec.phases.append(ConvPhaseSpec(wcat.float(), s3.phases[0].taps, (0, 0, 0)))
        y = self.alloc((N, 1, Ho, Wo, Cc))
        self.keep += [t2.t, x.t]
        d = H.ConvDesc()
        d.dtype = self.code
        d.N, d.D, d.H, d.W, d.Cin = N, 1, Ho, Wo, P
        d.Do, d.Ho, d.Wo = 1, Ho, Wo
        d.stride = H.i3((1, 1, 1)); d.pad = H.i3((0, 0, 0))
        d.OD, d.OH, d.OW = 1, Ho, Wo
        d.out_stride = H.i3((1, 1, 1))
        d.Cout, d.ldc, d.cout_pad, d.k_pad = Cc, Cc, Cc, kp
        d.nphase, d.flags, d.tile, d.stages = 1, spec.flags, 0, 0
        wdev = self.const(wcat, self.dtype)
        tdev = self.const(s3.phases[0].taps)
        wfr = torch.empty_like(wdev)
        lib = None if self.dry_run else H.lib()
        if not self.dry_run:
            H.check(lib.lt_conv_pack_weights32(wdev.data_ptr(), Cc, kp, wfr.data_ptr(), H.cur_stream()), "lt_conv_pack_weights32")
        d.phase[0].weight, d.phase[0].taps, d.phase[0].ntaps, d.phase[0].out_off = wdev.data_ptr(), tdev.data_ptr(), 1, H.i3((0, 0, 0))
        d.phase[0].weight_frag, d.phase[0].weight_frag_layout = wfr.data_ptr(), 3
        c2 = H.ConvCat2()
        c2.x, c2.cin, c2.H, c2.W, c2.stride = x.t.data_ptr(), Cin2, x.shape[2], x.shape[3], stride_down
        bi, sh = self.const(spec.bias), self.const(spec.shift)
        self.keep += [wfr, d, c2]
        macs = N * Ho * Wo * Cc * kp
        self.flops += 2 * macs
        esz = t2.t.element_size()
        nbytes = (t2.t.numel() + N * Ho * Wo * Cin2 + y.t.numel() + wcat.numel()) * esz
        label = "conv1x1 %d+%d->%d @%s (expand + stride-%d downsample)" % (P, Cin2, Cc, "x".join(str(v) for v in (N, 1, Ho, Wo)), stride_down)
        self._add(lambda s, d=d, xp=t2.t.data_ptr(), c2=c2, bip=bi.data_ptr(), shp=sh.data_ptr(), yp=y.t.data_ptr():
                  H.check(lib.lt_conv_cat2_fwd(C.byref(d), xp, C.byref(c2), bip, None, shp, None, yp, s), "lt_conv_cat2_fwd"),
                  "conv", label, 2 * macs, nbytes, {"cat2": True, "spec": spec, "x": t2, "x2": x, "stride2": stride_down, "y": y})
        return y

    # ---- split-K for the tiny levels of V2V ---------------------------------------------------------------------------------------
    def splitk_slices(self, spec, weight, transposed, out_f32, sigmoid, out):
        """Number of tap groups S the reduction of this convolution is cut into (1 = not split).  Taken for bf16 plans' 3 x 3 x 3 / stride 1
        convolutions with >= 128 input channels on volumes of at most 8^3 voxels -- V2V's 128 -> 128 layers at the 8^3 / 4^3 / 2^3 levels
        (v2v.py:78-90), 18 launches of ~30 us each whatever the batch: K = 3456 is a 54-step latency chain for the one or few workgroups
        the few output rows give.  S is chosen so that tiles x S fills the chip (<= 8: lt_conv_fwd's phase limit)."""
        if (self.dtype != torch.bfloat16 or self.live_weights or transposed or out_f32 or sigmoid or out is not None or os.environ.get("LT_CONV_NO_SPLITK") == "1"
                or self.tile_override):
            return 1
        if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or spec.stride != (1, 1, 1) or spec.pad != (1, 1, 1):
            return 1
        if spec.Cin < 128 or spec.Cin % 64 or spec.Cout % 4 or spec.Cout != spec.cout_pad or spec.D * spec.H * spec.W > 512:
            return 1
        rows = spec.N * spec.Do * spec.Ho * spec.Wo
        bm = 128 if rows >= 8192 else 64
        tiles = -(-rows // bm) * (spec.cout_pad // bm)
        return max(1, min(8, 256 // tiles))

    def _conv_splitk(self, x, weight, spec, S, residual):
        """The convolution as S tap-group phases of ONE lt_conv_fwd launch (fp32 partial sums, depth-stacked) + lt_splitk_reduce (the
        real epilogue).  Returns the output Act."""
        ph0 = spec.phases[0]
        ntaps, Cin = ph0.taps.shape[0], spec.Cin
        bounds = [ntaps * i // S for i in range(S + 1)]            # contiguous tap groups, sizes differ by at most one
        kstep = k_step_of(self.dtype)
        kp = (max(bounds[i + 1] - bounds[i] for i in range(S)) * Cin + kstep - 1) // kstep * kstep
        ident = (torch.zeros(spec.cout_pad), torch.ones(spec.cout_pad), torch.zeros(spec.cout_pad))
        pspec = ConvSpec(spec.N, spec.D, spec.H, spec.W, Cin, spec.Do, spec.Ho, spec.Wo, spec.stride, spec.pad, S * spec.Do, spec.Ho, spec.Wo, (1, 1, 1),
                         spec.Cout, spec.cout_pad, kp, H.EPI_STORE_F32, *ident)
        for i in range(S):
            t0, t1 = bounds[i], bounds[i + 1]
            pspec.phases.append(ConvPhaseSpec(_pad_k(ph0.weight[:, t0 * Cin:t1 * Cin], spec.cout_pad, kp), ph0.taps[t0:t1].contiguous(), (i * spec.Do, 0, 0)))
        part = self.alloc((spec.N, S * spec.Do, spec.Ho, spec.Wo, spec.Cout), torch.float32)
        y = self.alloc((spec.N, spec.Do, spec.Ho, spec.Wo, spec.Cout))
        self.keep.append(x.t)
        if residual is not None:
            assert residual.shape == y.shape and residual.t.dtype == self.dtype, (residual.shape, y.shape)
            self.keep.append(residual.t)
        rows = spec.N * spec.Do * spec.Ho * spec.Wo
        d = H.ConvDesc()
        d.dtype = self.code
        d.N, d.D, d.H, d.W, d.Cin = spec.N, spec.D, spec.H, spec.W, Cin
        d.Do, d.Ho, d.Wo = spec.Do, spec.Ho, spec.Wo
        d.stride = H.i3(spec.stride); d.pad = H.i3(spec.pad)
        d.OD, d.OH, d.OW = S * spec.Do, spec.Ho, spec.Wo
        d.out_stride = H.i3((1, 1, 1))
        d.Cout, d.ldc, d.cout_pad, d.k_pad = spec.Cout, spec.Cout, spec.cout_pad, kp
        d.nphase, d.flags, d.stages = S, H.EPI_STORE_F32, self.stages
        d.tile = H.TILE2_128x128 if rows >= 8192 else H.TILE2_64x64        # the generic implicit GEMM: its grid.y runs the phases side by side
        wdevs = []
        for i, ph in enumerate(pspec.phases):
            wdev, tdev = self.const(ph.weight, self.dtype), self.const(ph.taps)
            wdevs.append(wdev)
            d.phase[i].weight = wdev.data_ptr(); d.phase[i].taps = tdev.data_ptr()
            d.phase[i].ntaps = ph.taps.shape[0]; d.phase[i].out_off = H.i3(ph.out_off)
        ibi, isc, ish = (self.const(t) for t in ident)
        bi, sc, sh = self.const(spec.bias), self.const(spec.scale), self.const(spec.shift)
        self.keep.append(d)
        macs = rows * spec.Cout * ntaps * weight.shape[1]
        self.flops += 2 * macs
        lib = None if self.dry_run else H.lib()
        label = "conv3x3x3 %d->%d @%s" % (Cin, spec.Cout, "x".join(str(v) for v in (spec.N, spec.Do, spec.Ho, spec.Wo)))
        esz = x.t.element_size()
        self._add(lambda s, d=d, xp=x.t.data_ptr(), bip=ibi.data_ptr(), scp=isc.data_ptr(), shp=ish.data_ptr(), yp=part.t.data_ptr():
                  H.check(lib.lt_conv_fwd(C.byref(d), xp, bip, scp, shp, None, yp, s), "lt_conv_fwd(split-K)"),
                  "conv", label + " split-K x%d" % S, 2 * macs, x.t.numel() * esz + part.t.numel() * 4 + sum(p.weight.numel() for p in pspec.phases) * esz,
                  {"spec": pspec, "x": x, "y": part, "res": None, "wdev": wdevs, "bias_dev": ibi})
        R = spec.Do * spec.Ho * spec.Wo
        self._add(lambda s, pp=part.t.data_ptr(), bip=bi.data_ptr(), scp=sc.data_ptr(), shp=sh.data_ptr(),
                  rp=H.ptr(residual.t) if residual is not None else None, yp=y.t.data_ptr(), a=(S, spec.N, R, spec.Cout, spec.flags):
                  H.check(lib.lt_splitk_reduce(self.code, pp, a[0], a[1], a[2], a[3], bip, scp, shp, rp, yp, a[4], s), "lt_splitk_reduce"),
                  "conv", label + " split-K reduce", 0, part.t.numel() * 4 + (y.t.numel() + (residual.t.numel() if residual is not None else 0)) * esz,
                  {"splitk_reduce": True, "spec": spec, "S": S, "part": part, "res": residual, "y": y})
        self.release(part)
        return y

    def can_chain_pointwise(self, x, layers):
        """True when lt_pwchain_fwd covers this chain: bf16 plan, 32 input channels, 1x1x1 kernels, inner widths 32, last
        width <= 32 stored as fp32, voxel count a multiple of 64.  layers: [(weight, bias, bn, relu), ...]."""
        if self.dtype != torch.bfloat16 or not (1 <= len(layers) <= H.PWCHAIN_MAX) or x.shape[-1] != 32:
            return False
        if int(np.prod(x.shape[:-1])) % 64:
            return False
        cin = 32
        for i, (w, _, _, _) in enumerate(layers):
            if tuple(w.shape[2:]) != (1,) * (w.dim() - 2) or w.shape[1] != cin or w.shape[0] > 32:
                return False
            if i + 1 < len(layers) and w.shape[0] != 32:
                return False
            cin = w.shape[0]
        return True

    def pwchain(self, x, layers, planar=False):
        """Chain of pointwise convolutions in one pass over the volume (lt_pwchain_fwd); the last layer's output is fp32.
        layers: [(weight [Cout,Cin,1,1,1], bias, bn-tuple-or-None, relu), ...].  Returns the output Act [N,D,H,W,Cout_last];
        planar=True stores it as (N, Cout, D, H, W) -- the Act's tensor is then the channels-last VIEW of that storage
        (not contiguous; `.t.permute(0, 4, 1, 2, 3)` is), which is the layout lt_softargmax3d_fwd streams fastest."""
        assert self.can_chain_pointwise(x, layers)
        N, D, Hh, W, _ = x.shape
        d = H.PwChainDesc()
        d.dtype, d.nlayers, d.rows, d.cin = self.code, len(layers), N * D * Hh * W, 32
        specs = []
        shape = x.shape
        flops = 0
        for i, (w, bias, bn, relu) in enumerate(layers):
            last = i + 1 == len(layers)
            flags = (H.EPI_RELU_POST if relu else 0) | (H.EPI_STORE_F32 if last else 0)
            spec = make_conv_spec(w, bias, bn, shape, 1, 0, self.dtype, False, flags)
            specs.append(spec)
            wdev = self.const(spec.phases[0].weight, self.dtype)
            bi, sc, sh = self.const(spec.bias), self.const(spec.scale), self.const(spec.shift)
            d.cout[i], d.k_pad[i], d.flags[i] = spec.Cout, spec.k_pad, flags
            d.weight[i], d.bias[i], d.scale[i], d.shift[i] = wdev.data_ptr(), bi.data_ptr(), sc.data_ptr(), sh.data_ptr()
            flops += 2 * d.rows * spec.Cout * w.shape[1]
            shape = (N, D, Hh, W, spec.Cout)
        d.ldy = specs[-1].Cout
        planar = bool(planar) and (D * Hh * W) % 64 == 0
        if planar:
            d.plane = D * Hh * W
            y = Act(self.alloc((N, shape[-1], D, Hh, W), torch.float32).t.permute(0, 2, 3, 4, 1))
            y.pooled = False
        else:
            y = self.alloc(shape, torch.float32)
        self.keep.append(x.t)
        self.keep.append(d)
        self.flops += flops
        lib = None if self.dry_run else H.lib()
        label = "pwchain " + "->".join(str(c) for c in [32] + [sp.Cout for sp in specs]) + " @" + "x".join(str(v) for v in (N, D, Hh, W))
        nbytes = x.t.numel() * x.t.element_size() + y.t.numel() * 4
        self._add(lambda s, d=d, xp=x.t.data_ptr(), yp=y.t.data_ptr(): H.check(lib.lt_pwchain_fwd(C.byref(d), xp, yp, s), "lt_pwchain_fwd"),
                  "pwchain", label, flops, nbytes, {"specs": specs, "x": x, "y": y})
        return y

    # ---- whole identity Bottleneck block in one launch (ResNet layer1 / layer2) -----------------------------------------------------
    def can_bottleneck(self, x, convs, strides):
        """True when lt_bottleneck_fwd covers the block: bf16 plan over build-time weights, 2D map, three stride-1 convolutions
        1x1 C->P, 3x3 P->P, 1x1 P->C with (C, P) = (256, 64) or (512, 128), no downsample (the caller checks), H % 8 == 0 and W % 16 == 0."""
        if self.dtype != torch.bfloat16 or self.live_weights or self.tile_override or os.environ.get("LT_NO_BNECK") == "1":
            return False
        N, D, Hh, W, Cc = x.shape
        if D != 1 or len(convs) != 3 or any(s != 1 for s in strides):
            return False
        P = convs[0].shape[0]
        if (Cc, P) not in ((256, 64), (512, 128)):
            return False
        if (tuple(convs[0].shape) != (P, Cc, 1, 1) or tuple(convs[1].shape) != (P, P, 3, 3) or tuple(convs[2].shape) != (Cc, P, 1, 1)):
            return False
        return Hh % 8 == 0 and W % 16 == 0 and N * Hh * W * Cc < 2 ** 31

    def bottleneck(self, x, convs, bns):
        """relu(bn3(conv1x1(relu(bn2(conv3x3(relu(bn1(conv1x1(x)))))))) + x) in ONE launch (lt_bottleneck_fwd: the two bottleneck-width tensors
        stay in LDS).  convs: the three Conv2d weights, bns: their BatchNorm tuples.  Returns the output Act (a new buffer: the kernel
        cannot run in place)."""
        assert self.can_bottleneck(x, convs, (1, 1, 1))
        N, _, Hh, W, Cc = x.shape
        P = convs[0].shape[0]
        specs = []
        shape = x.shape
        for i, (w, bn) in enumerate(zip(convs, bns)):
            spec = make_conv_spec(w, None, bn, shape, 1, 1 if i == 1 else 0, self.dtype, False, H.EPI_RELU_POST)
            specs.append(spec)
            shape = (N, 1, Hh, W, spec.Cout)
        y = self.alloc((N, 1, Hh, W, Cc))
        d = H.BneckDesc()
        d.dtype, d.N, d.H, d.W, d.C, d.P = self.code, N, Hh, W, Cc, P
        lib = None if self.dry_run else H.lib()
        flops = 0
        for i, spec in enumerate(specs):
            wdev = self.const(spec.phases[0].weight, self.dtype)
            assert spec.cout_pad == spec.Cout and spec.k_pad == spec.phases[0].taps.shape[0] * spec.Cin, (spec.cout_pad, spec.k_pad)
            sc, sh = self.const(spec.scale), self.const(spec.shift)
            wfr = torch.empty_like(wdev)
            if not self.dry_run:
                H.check(lib.lt_conv_pack_weights_t32(wdev.data_ptr(), spec.cout_pad, spec.k_pad, spec.Cin, int(spec.phases[0].taps.shape[0]),
                                                     wfr.data_ptr(), H.cur_stream()), "lt_conv_pack_weights_t32")
            self.keep.append(wfr)
            d.weight[i], d.bias[i], d.scale[i], d.shift[i] = wfr.data_ptr(), None, sc.data_ptr(), sh.data_ptr()
            flops += 2 * N * Hh * W * spec.Cout * spec.phases[0].taps.shape[0] * spec.Cin
        self.keep.append(x.t)
        self.keep.append(d)
        self.flops += flops
        esz = x.t.element_size()
        nbytes = (x.t.numel() + y.t.numel()) * esz + sum(sp.phases[0].weight.numel() for sp in specs) * esz
        label = "bneck %d->%d->%d @%s" % (Cc, P, Cc, "x".join(str(v) for v in (N, 1, Hh, W)))
        self._add(lambda s, d=d, xp=x.t.data_ptr(), yp=y.t.data_ptr(): H.check(lib.lt_bottleneck_fwd(C.byref(d), xp, yp, s), "lt_bottleneck_fwd"),
                  "conv", label, flops, nbytes, {"bneck": True, "specs": specs, "x": x, "y": y})
        return y

    # ---- the first Bottleneck of ResNet layer1 (downsample branch, stride 1) in one launch --------------------------------------------------
    def can_bottleneck_ds(self, x, convs, strides, w_down, stride_down):
        """True when lt_bottleneck_ds_fwd covers the block: bf16 plan over build-time weights, 2D map, stride-1 convolutions 1x1 64->64, 3x3 64->64,
        1x1 64->256 and a stride-1 1x1 64->256 downsample, H % 8 == 0 and W % 16 == 0 (LT_NO_BNECK_DS=1 or LT_NO_BNECK=1: off -- the A/B switches)."""
        if (self.dtype != torch.bfloat16 or self.live_weights or self.tile_override or os.environ.get("LT_NO_BNECK") == "1" or
                os.environ.get("LT_NO_BNECK_DS") == "1"):
            return False
        N, D, Hh, W, Cin = x.shape
        if D != 1 or len(convs) != 3 or any(s != 1 for s in strides) or stride_down != 1:
            return False
        P, Cc = convs[0].shape[0], convs[2].shape[0]
        if (Cin, P, Cc) != (64, 64, 256):
            return False
        if (tuple(convs[0].shape) != (P, Cin, 1, 1) or tuple(convs[1].shape) != (P, P, 3, 3) or tuple(convs[2].shape) != (Cc, P, 1, 1) or
                tuple(w_down.shape) != (Cc, Cin, 1, 1)):
            return False
        return Hh % 8 == 0 and W % 16 == 0 and N * Hh * W * Cc < 2 ** 31

    def bottleneck_ds(self, x, convs, bns, w_down, bn_down):
        """relu(bn3(conv1x1(relu(bn2(conv3x3(relu(bn1(conv1x1(x)))))))) + bn_d(conv1x1_d(x))) in ONE launch (lt_bottleneck_ds_fwd): the two bottleneck-width
        tensors stay in LDS and the downsample branch is computed from the tile of x that is there already.  Returns the output Act."""
        assert self.can_bottleneck_ds(x, convs, (1, 1, 1), w_down, 1)
        N, _, Hh, W, Cin = x.shape
        P, Cc = convs[0].shape[0], convs[2].shape[0]
        specs = []
        shape = x.shape
        for i, (w, bn) in enumerate(zip(convs, bns)):
            spec = make_conv_spec(w, None, bn, shape, 1, 1 if i == 1 else 0, self.dtype, False, H.EPI_RELU_POST)
            specs.append(spec)
            shape = (N, 1, Hh, W, spec.Cout)
        specs.append(make_conv_spec(w_down, None, bn_down, x.shape, 1, 0, self.dtype, False, 0))
        y = self.alloc((N, 1, Hh, W, Cc))
        d = H.BneckDsDesc()
        d.dtype, d.N, d.H, d.W, d.Cin, d.P, d.C = self.code, N, Hh, W, Cin, P, Cc
        lib = None if self.dry_run else H.lib()
        flops = 0
        for i, spec in enumerate(specs):
            wdev = self.const(spec.phases[0].weight, self.dtype)
            assert spec.cout_pad == spec.Cout and spec.k_pad == spec.phases[0].taps.shape[0] * spec.Cin, (spec.cout_pad, spec.k_pad)
            assert not bool(spec.bias.any()), "ResNet convolutions carry no bias"
            sc, sh = self.const(spec.scale), self.const(spec.shift)
            wfr = torch.empty_like(wdev)
            if not self.dry_run:
                H.check(lib.lt_conv_pack_weights_t32(wdev.data_ptr(), spec.cout_pad, spec.k_pad, spec.Cin, int(spec.phases[0].taps.shape[0]),
                                                     wfr.data_ptr(), H.cur_stream()), "lt_conv_pack_weights_t32")
            self.keep.append(wfr)
            d.weight[i], d.scale[i], d.shift[i] = wfr.data_ptr(), sc.data_ptr(), sh.data_ptr()
            flops += 2 * N * Hh * W * spec.Cout * spec.phases[0].taps.shape[0] * spec.Cin
        self.keep.append(x.t)
        self.keep.append(d)
        self.flops += flops
        esz = x.t.element_size()
        nbytes = (x.t.numel() + y.t.numel()) * esz + sum(sp.phases[0].weight.numel() for sp in specs) * esz
        label = "bneck-ds %d->%d->%d @%s" % (Cin, P, Cc, "x".join(str(v) for v in (N, 1, Hh, W)))
        self._add(lambda s, d=d, xp=x.t.data_ptr(), yp=y.t.data_ptr(): H.check(lib.lt_bottleneck_ds_fwd(C.byref(d), xp, yp, s), "lt_bottleneck_ds_fwd"),
                  "conv", label, flops, nbytes, {"bneck_ds": True, "specs": specs, "x": x, "y": y})
        return y

    # ---- the seam between two identity Bottleneck blocks in one launch (ResNet layer3): expand of block i + reduce of block i + 1 -----------
    def can_expand_reduce(self, t2, res, w_expand, w_reduce):
        """True when lt_expand_reduce_fwd covers the seam: bf16 plan over build-time weights, 2D maps, 1x1 P -> C expand with a C-channel residual and
        1x1 C -> P reduce with (C, P) = (1024, 256) (LT_NO_XR=1: off -- the A/B switch)."""
        if self.dtype != torch.bfloat16 or self.live_weights or self.tile_override or os.environ.get("LT_NO_XR") == "1":
            return False
        if t2.shape[1] != 1 or res.shape[1] != 1 or tuple(t2.shape[:4]) != tuple(res.shape[:4]):
            return False
        P, Cc = t2.shape[-1], res.shape[-1]
        if (Cc, P) != (1024, 256):
            return False
        # one tile per workgroup and one workgroup per CU.  With 96-row tiles only, 1 / 2 samples (24 / 48 tiles) lost 8 % / 2.5 % end to end to the two
        # launches (144-row tiles x 4 column tiles, two workgroups per CU) and the builder fused from 64 tiles on; the launcher now picks 64- and 32-row tiles
        # for small row counts (measured, forward samples/s, 96 / 64 / 32-row tiles / two launches: 1 sample 277 / 286 / 297 / 301, 2 samples 446 / 461 / 470 /
        # 454, 5 samples 857 / 876 / 836 / 827, 10 samples 1186 / 1135 / 1122 / 1104), so the seam is fused from 2 samples = 36 tiles of 96 rows on
        if t2.shape[0] * t2.shape[2] * t2.shape[3] < 36 * 96 and os.environ.get("LT_XR_ANY_SIZE") != "1":
            return False
        return tuple(w_expand.shape) == (Cc, P, 1, 1) and tuple(w_reduce.shape) == (P, Cc, 1, 1)

    def expand_reduce(self, t2, res, w_expand, bn_expand, w_reduce, bn_reduce):
        """y = relu(bn3(conv1x1(t2)) + res) and t1' = relu(bn1'(conv1x1'(y))) in ONE launch (lt_expand_reduce_fwd: the reduce consumes y from LDS).
        Returns (y, t1') as new Acts."""
        assert self.can_expand_reduce(t2, res, w_expand, w_reduce)
        N, _, Hh, W, P = t2.shape
        Cc = res.shape[-1]
        s3 = make_conv_spec(w_expand, None, bn_expand, t2.shape, 1, 0, self.dtype, False, H.EPI_RELU_POST)
        s1 = make_conv_spec(w_reduce, None, bn_reduce, res.shape, 1, 0, self.dtype, False, H.EPI_RELU_POST)
        y = self.alloc((N, 1, Hh, W, Cc))
        t1 = self.alloc((N, 1, Hh, W, P))
        d = H.XrDesc()
        d.dtype, d.C, d.P, d.M = self.code, Cc, P, N * Hh * W
        lib = None if self.dry_run else H.lib()
        flops = 0
        for i, spec in enumerate((s3, s1)):
            assert spec.cout_pad == spec.Cout and spec.k_pad == spec.Cin and not bool(spec.bias.any()), (spec.cout_pad, spec.k_pad)
            wdev = self.const(spec.phases[0].weight, self.dtype)
            sc, sh = self.const(spec.scale), self.const(spec.shift)
            wfr = torch.empty_like(wdev)
            if not self.dry_run:
                H.check(lib.lt_conv_pack_weights_t32(wdev.data_ptr(), spec.cout_pad, spec.k_pad, spec.Cin, 1, wfr.data_ptr(), H.cur_stream()), "lt_conv_pack_weights_t32")
            self.keep += [wfr, sc, sh]
            d.weight[i], d.scale[i], d.shift[i] = wfr.data_ptr(), sc.data_ptr(), sh.data_ptr()
            flops += 2 * N * Hh * W * spec.Cout * spec.Cin
        packed = self.const(torch.cat([s3.scale, s3.shift, s1.scale, s1.shift]))          # the four tables back to back: one LDS-DMA in the kernel's prologue
        d.consts = packed.data_ptr()
        self.keep += [t2.t, res.t, d, packed]
        self.flops += flops
        esz = t2.t.element_size()
        nbytes = (t2.t.numel() + res.t.numel() + y.t.numel() + t1.t.numel() + 2 * Cc * P) * esz
        label = "xr expand %d->%d + reduce %d->%d @%s" % (P, Cc, Cc, P, "x".join(str(v) for v in (N, 1, Hh, W)))
        self._add(lambda s, d=d, a=t2.t.data_ptr(), r=res.t.data_ptr(), yp=y.t.data_ptr(), tp=t1.t.data_ptr():
                  H.check(lib.lt_expand_reduce_fwd(C.byref(d), a, r, yp, tp, s), "lt_expand_reduce_fwd"),
                  "conv", label, flops, nbytes, {"xr": True, "specs": [s3, s1], "x": t2, "res": res, "y": y, "t1": t1})
        return y, t1

    def can_stem_pool(self, x, weight, stride, pad, pool):
        """True when lt_stem_pool_fwd covers conv -> BN -> ReLU -> max pool: bf16 plan, 2D map with 8 (padded) channels,
        7x7 / stride 2 / pad 3 convolution to 64 channels, 3x3 / stride 2 / pad 1 pool."""
        return (self.dtype == torch.bfloat16 and x.shape[1] == 1 and x.shape[-1] == 8 and weight.dim() == 4
                and tuple(weight.shape[2:]) == (7, 7) and weight.shape[0] == 64 and weight.shape[1] <= 8
                and stride == 2 and pad == 3 and tuple(pool) == (3, 2, 1))

    def stem_pool(self, x, weight, bn, image_cell=None):
        """conv 7x7/2 (no bias) + eval BN + ReLU + max pool 3x3/2 in one pass (lt_stem_pool_fwd).  x: Act [N,1,H,W,8];
        returns Act [N,1,Hp,Wp,64].  image_cell: a dict whose "ptr" the caller sets, before every Plan.run, to the data
        pointer of its contiguous fp32 (N,3,H,W) images: the kernel then reads THAT tensor (rounding to bf16 on the fly) instead
        of x, the op stays outside the captured graph (its input address changes per call), and the separate layout pass into x
        is not needed.  Must be the first op of the plan; ignored by dry-run plans (their interpreter reads x)."""
        assert self.can_stem_pool(x, weight, 2, 3, (3, 2, 1))
        if self.dry_run or weight.shape[1] != 3:
            image_cell = None
        assert image_cell is None or not self.ops, "an op reading the caller's tensor must be the first of the plan"
        N, _, Hh, W, _ = x.shape
        spec = make_conv_spec(weight, None, bn, x.shape, 2, 3, self.dtype, False, H.EPI_RELU_POST)
        Hp, Wp = (spec.Ho - 1) // 2 + 1, (spec.Wo - 1) // 2 + 1
        y = self.alloc((N, 1, Hp, Wp, 64))
        wdev = self.const(spec.phases[0].weight, self.dtype)
        bi, sc, sh = self.const(spec.bias), self.const(spec.scale), self.const(spec.shift)
        lib = None if self.dry_run else H.lib()
        # the kernel reads its weights in MFMA fragment order: packed once, here (on the plan's device, legacy stream)
        wpk = torch.empty(57344 if self.dry_run else lib.lt_stem_packed_bytes(), dtype=torch.uint8, device=self.device)
        if not self.dry_run:
            H.check(lib.lt_stem_pack_weights(wdev.data_ptr(), spec.k_pad, wpk.data_ptr(), H.cur_stream()), "lt_stem_pack_weights")
            torch.cuda.current_stream().synchronize()
        d = H.StemDesc()
        d.dtype, d.N, d.H, d.W, d.Cin, d.Cout = self.code, N, Hh, W, (3 if image_cell is not None else 8), 64
        d.weight, d.bias, d.scale, d.shift = wpk.data_ptr(), bi.data_ptr(), sc.data_ptr(), sh.data_ptr()
        d.x_layout = 1 if image_cell is not None else 0
        self.keep.append(x.t)
        self.keep.append(d)
        self.keep.append(wpk)
        flops = 2 * N * spec.Ho * spec.Wo * 64 * 49 * weight.shape[1]
        self.flops += flops
        esz = x.t.element_size()
        if image_cell is not None:
            def launch(s, d=d, yp=y.t.data_ptr(), cell=image_cell):
                if not cell.get("ptr"):
                    raise RuntimeError("stem_pool: the plan reads the caller's images; set image_cell['ptr'] before Plan.run")
                H.check(lib.lt_stem_pool_fwd(C.byref(d), cell["ptr"], yp, s), "lt_stem_pool_fwd")
            nbytes = N * 3 * Hh * W * 4 + y.t.numel() * esz + spec.phases[0].weight.numel() * esz
            self.npre += 1
        else:
            def launch(s, d=d, xp=x.t.data_ptr(), yp=y.t.data_ptr()):
                H.check(lib.lt_stem_pool_fwd(C.byref(d), xp, yp, s), "lt_stem_pool_fwd")
            nbytes = (x.t.numel() + y.t.numel()) * esz + spec.phases[0].weight.numel() * esz
        self._add(launch, "stem", "stem conv7x7/2+pool3x3/2 %d->64 @%s" % (d.Cin, "x".join(str(v) for v in (N, Hh, W))), flops, nbytes,
                  {"spec": spec, "x": x, "y": y})
        return y

    def maxpool(self, x, k, s, p, nd):
        N, D, Hh, W, Cc = x.shape
        kk = (1, k, k) if nd == 2 else (k, k, k)
        ss = (1, s, s) if nd == 2 else (s, s, s)
        pp = (0, p, p) if nd == 2 else (p, p, p)
        od = [(dim + 2 * pp[i] - kk[i]) // ss[i] + 1 for i, dim in enumerate((D, Hh, W))]
        y = self.alloc((N, od[0], od[1], od[2], Cc))
        self.keep.append(x.t)
        lib = None if self.dry_run else H.lib()
        self._add(lambda st, xp=x.t.data_ptr(), yp=y.t.data_ptr(), a=(N, D, Hh, W, Cc), kk=H.i3(kk), ss=H.i3(ss), pp=H.i3(pp):
                  H.check(lib.lt_maxpool_fwd(self.code, xp, yp, a[0], a[1], a[2], a[3], a[4], kk, ss, pp, st), "lt_maxpool_fwd"),
                  "maxpool", "maxpool%dd k%d @%s" % (nd, k, "x".join(map(str, x.shape))), 0,
                  (x.t.numel() + y.t.numel()) * x.t.element_size(), {"x": x, "y": y, "k": kk, "s": ss, "p": pp})
        return y

    def global_avgpool(self, x):
        """x: Act [N,1,H,W,C] -> Act [1,1,1,N,C] (a one-row 'image' of N pixels: feeds 1x1 convs = linears)."""
        N, D, Hh, W, Cc = x.shape
        y = self.alloc((1, 1, 1, N, Cc))
        self.keep.append(x.t)
        lib = None if self.dry_run else H.lib()
        self._add(lambda st, xp=x.t.data_ptr(), yp=y.t.data_ptr(), a=(N, D * Hh * W, Cc):
                  H.check(lib.lt_global_avgpool(self.code, xp, yp, a[0], a[1], a[2], st), "lt_global_avgpool"), "avgpool", "global_avgpool",
                  info={"x": x, "y": y})
        return y

    def custom(self, fn, kind="op", label="", flops=0, nbytes=0, info=None, tail=False):
        """fn(stream) -> None: any other liblt_hip launch; nbytes = its ALGORITHMIC HBM bytes (roofline numerator).
        tail=True: fn(stream, outs=None) is one of the LAST ops of the plan and writes the tensors the caller returns; it stays
        outside the captured graph so that Plan.run can point it at freshly allocated outputs (no clone of the results)."""
        assert tail or not self.ntail, "tail ops must be recorded last"
        self._add(fn, kind, label or kind, flops, nbytes, info)
        self.ntail += 1 if tail else 0

    def finish(self):
        return Plan(self.ops, self.keep, self.device, self.flops, self.bytes_alloc, self.dry_run, self.ntail, self.npre)

    def finish_after(self, result):
        """finish() for a plan whose last recorded Act is its result (kept on the plan as ``.result``)."""
        plan = self.finish()
        plan.result = result
        return plan


class Plan:
    def __init__(self, ops, keep, device, flops, bytes_alloc, dry_run=False, ntail=0, npre=0):
        self.ops, self.keep, self.device = ops, keep, device
        self.flops, self.bytes_alloc = flops, bytes_alloc
        self.graph = None
        self.dry_run = dry_run
        # ops[:npre] read a caller-owned tensor and ops[nhead:] write the caller's result tensors: both are launched eagerly
        # around the captured graph of ops[npre:nhead]
        self.npre = npre
        self.nhead = len(ops) - ntail

    def run_eager(self, stream):
        if self.dry_run:
            raise RuntimeError("a dry-run plan cannot execute: liblt_hip runs on the GPU only")
        for fn, _ in self.ops:
            fn(stream)

    def run_profiled(self, stream, reps=3):
        """Eager launches with a hipEvent pair around EVERY op on `stream`; returns one record per op with the mean
        duration in ms (kind, label, flops, bytes, ms).  Used by bench.py for the per-kernel roofline numbers."""
        n = len(self.ops)
        tot = [0.0] * n
        for _ in range(reps):
            evs = [(H.Event(), H.Event()) for _ in range(n)]
            for (fn, _), (e0, e1) in zip(self.ops, evs):
                e0.record(stream)
                fn(stream)
                e1.record(stream)
            for i, (e0, e1) in enumerate(evs):
                tot[i] += e0.elapsed_ms(e1)
        return [dict(meta, ms=tot[i] / reps) for i, (_, meta) in enumerate(self.ops)]

    def capture(self, stream):
        g = H.Graph()
        g.capture(stream, lambda: [fn(stream) for fn, _ in self.ops[self.npre:self.nhead]])
        self.graph = g

    def run(self, stream, outs=None):
        """Launches the pre ops, replays the captured graph (or launches eagerly), then the tail ops; outs: passed to every
        tail op (the tensors they should write instead of their recorded outputs), None = the recorded ones."""
        if self.dry_run:
            raise RuntimeError("a dry-run plan cannot execute: liblt_hip runs on the GPU only")
        for fn, _ in self.ops[:self.npre]:
            fn(stream)
        if self.graph is not None:
            self.graph.launch(stream)
        else:
            for fn, _ in self.ops[self.npre:self.nhead]:
                fn(stream)
        for fn, _ in self.ops[self.nhead:]:
            fn(stream, outs)


class PlanCache(nn.Module):
    """Shared plan cache: one recorded + graph-captured launch list per (shape, dtype, device, baked scalars).

    A plan bakes folded / packed COPIES of every weight (BN fold, bf16 fragment packs, a hipGraph with their addresses), so
    every entry also remembers a fingerprint of the weights it was built from -- ``data_ptr`` and the autograd version counter
    of every parameter and buffer of the whole tree -- and is rebuilt when that changes: ``load_state_dict`` on the net OR any
    child, ``optimizer.step()``, ``param.copy_()``, ``.to()``.  (Writes through ``param.data`` bypass the version counter:
    call ``invalidate_plans()`` after those.)  The cache is a small LRU: a ragged last batch does not pin a second set of
    multi-GB buffers forever."""

    max_plans = 3

    def _init_plan_cache(self):
        self._plans = OrderedDict()
        self.register_load_state_dict_post_hook(lambda m, k: m._plans.clear())

    def invalidate_plans(self):
        """Drops every recorded plan.  Needed only after writes the fingerprint cannot see (``param.data`` edits)."""
        self._plans.clear()

    def weights_fingerprint(self):
        """Changes whenever a parameter / buffer of the tree is re-bound, moved or written in place through autograd-visible
        ops (every such write bumps ``Tensor._version``)."""
        mods = self.__dict__.get("_fp_modules")
        if mods is None:       # the module LIST is cached (walking the tree costs 2 ms per call on ResNet-152 + V2V; this loop 0.45 ms)
            mods = self.__dict__["_fp_modules"] = list(self.modules())
        fp = 0
        for mod in mods:
            for t in mod._parameters.values():
                if t is not None:
                    fp = (fp * 1000003 + t.data_ptr() + 7919 * t._version) & 0x1FFFFFFFFFFFFFFF
            for t in mod._buffers.values():
                if t is not None:
                    fp = (fp * 1000003 + t.data_ptr() + 7919 * t._version) & 0x1FFFFFFFFFFFFFFF
        return fp

    def __setattr__(self, name, value):
        if isinstance(value, nn.Module):          # a re-bound child: walk the tree again
            self.__dict__.pop("_fp_modules", None)
        super().__setattr__(name, value)

    def _plan_for(self, key, build):
        """LRU lookup; ``build()`` records a new plan.  Stale entries (weights changed since they were recorded) are rebuilt."""
        fp = self.weights_fingerprint()
        P = self._plans.get(key)
        if P is not None and P["fingerprint"] != fp:
            del self._plans[key]
            P = None
        if P is None:
            while len(self._plans) >= max(1, self.max_plans):
                self._plans.popitem(last=False)          # least recently used: frees its buffers and its hipGraph
            P = build()
            P["fingerprint"] = fp
            self._plans[key] = P
        else:
            self._plans.move_to_end(key)
        return P
