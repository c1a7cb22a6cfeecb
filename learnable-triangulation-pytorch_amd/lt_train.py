"""Training-mode execution of the reference-shaped networks (BASELINE config 5): the modules'
``record()`` methods -- the same code that records the inference plan -- run against a ``TrainTape`` that EXECUTES every layer
as it is recorded, in training mode (convolution without a folded BatchNorm -> batch statistics -> normalise + ReLU + residual),
keeps every launch as a closure over fixed buffers, and records the backward the same way on the first ``run_backward()``; later steps
REPLAY the two launch lists (DESIGN.md "Training step").  The backward of a layer:

    activation / BatchNorm backward   lt_bn_act_bwd / lt_act_bwd   (csrc/train.hip)
    bias gradient                     lt_channel_sum
    weight gradient                   lt_conv_wgrad               (exact-fp32 MFMA, no atomics) on a SIDE stream, behind an event
    input gradient                    lt_conv_fwd over dY with the weights transposed + flipped (stride 1), as a parity-phase transposed
                                      convolution (stride-2 layers: k=3/p=1 or k=1/p=0 with output_padding 1) or as the strided convolution a
                                      transposed layer is the adjoint of; an already existing gradient of the input rides in as the
                                      epilogue's residual, so sums over consumers cost no extra pass
    max pool                          lt_maxpool_bwd

The weights are live (lt_gather_f32 from the Parameters into each layer's GEMM layout, index maps built at record time), the
parameter gradients land in one flat arena (contiguous buckets for lt_dist.GradReducer).  fp32 throughout by default (the reference
trains in fp32); ``mixed=True`` takes the convolutions and their input gradients to the bf16 MFMA (bf16 operand copies, fp32
accumulation and storage).  All arithmetic is liblt_hip's, torch only owns the memory.  What the reference does with
``total_loss.backward()`` (train.py:233-236) arrives here through the autograd Function that wraps a network's training forward
(mvn/models/triangulation.py), so ``torch.optim`` and hooks see ordinary ``.grad`` tensors.
"""
import ctypes as C
import os

import numpy as np
import torch

import lt_engine as E
import lt_hip as H

BN_EPS = 1e-5


class TrainTape:
    """PlanBuilder-shaped object for ``record()`` (conv / maxpool / alloc / release).  RECORD ONCE, REPLAY EVERY STEP: while the
    modules' ``record()`` methods run, every launch is executed AND appended to ``fwd_ops``; the first backward walks the layers in
    reverse the same way into ``bwd_ops``.  Every buffer (activations, gradients, GEMM-layout weights, workspaces, the flat arena of
    parameter gradients) is allocated during recording and lives as long as the tape, so a later step is two loops over closures of
    raw pointers -- no allocation, no host-side weight handling, no synchronisation.  The weights are LIVE: each layer's first op is
    an ``lt_gather_f32`` from the Parameter (wherever the optimiser has left it) into the layer's GEMM layout, with an index map built
    at record time by pushing a tensor of indices through the very host code that lays out inference weights."""

    def __init__(self, device, params=(), momentum=0.1, reducer=None, bucket_bytes=64 << 20, mixed=False, act16=False, fp8_3d=False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("training runs on the GPU only (device=%s); there is no CPU fallback" % device)
        self.dtype, self.code, self.dry_run = torch.float32, H.LT_F32, False
        self.pb = E.PlanBuilder(device, torch.float32)          # builds the lt_conv_fwd descriptors
        # mixed precision: the convolutions (forward and input gradients) take bf16 COPIES of their operands to the bf16 MFMA and store
        # fp32 (LT_EPI_STORE_F32); activations, BatchNorm, weight gradients, optimiser stay fp32 (the master weights are the Parameters)
        # act16 (train_precision "act16", BASELINE config 5's "fp16 activations" -- bf16 here): on top of ``mixed``, every activation and every
        # activation gradient is STORED in bf16 only -- the convolutions read and write bf16 through their ordinary (inference) epilogues, so every
        # bf16 kernel of the forward applies to the forward and to the input gradients; BatchNorm reads / writes 2 bytes per element (its
        # statistics, gamma / beta, their gradients, the weight gradients, the master weights and the optimiser stay fp32)
        self.act16 = bool(act16)
        self.fp8_3d = bool(fp8_3d) and self.act16          # 'fp8v2v': the 3x3x3 convolutions (V2V) and their input gradients on the fp8 MFMA
        self.mixed = bool(mixed) or self.act16
        self.adt = torch.bfloat16 if self.act16 else torch.float32          # storage type of activations and activation gradients
        self.acode = H.LT_BF16 if self.act16 else H.LT_F32
        self.aflag = H.ACT_BF16 if self.act16 else 0
        self.pbh = E.PlanBuilder(device, torch.bfloat16) if self.mixed else None
        if self.pbh is not None:
            self.pbh.live_weights = True
            # round 6 (VERDICT r5 "next" 2 iv): the 16-bit step's convolutions read bf16 and store bf16 through their ordinary epilogues, so the fragment-order /
            # halo kernels of the inference forward apply -- given their weights in fragment order EVERY step: the builder packs the fragment copies as in
            # inference and the tape gathers the live Parameter straight into them through the composed index map (_frag_index_map).  LT_TRAIN_NO_FRAG=1: off (A/B)
            self.pbh.live_frag = self.act16 and os.environ.get("LT_TRAIN_NO_FRAG") is None
        # mixed precision, opt-in (LT_TRAIN_Y16=1): the output of a convolution that feeds a BatchNorm is STORED in bf16 -- its statistics, the
        # normalisation and the BatchNorm backward read 2 bytes instead of 4, and every bf16 kernel of the forward applies (fp32 stores are
        # restricted to the generic one and the column walk); BatchNorm's output, the gradients and everything else stay fp32.  Measured: the
        # step at 8 samples 87.5 -> 83.9 ms, the deviation from the fp32 step on the small fixture 3.8e-2 -> 5.5e-2 of the joints: off by default.
        self.y16 = self.mixed and (self.act16 or os.environ.get("LT_TRAIN_Y16") == "1")
        self._bf16 = {}                                         # id(Act) -> (Act, bf16 copy); cast op recorded with the first consumer
        self.fwd_ops, self.bwd_ops = [], []                     # fn(stream) closures, in launch order
        self._cur = self.fwd_ops
        self.recorders = []                                     # one per layer: records (= runs once) that layer's backward
        self.layer_outputs = []                                 # (label, output Act) of every convolution layer, in forward order
        self.grads = {}                                         # id(Act) -> (Act, gradient tensor of act.t's shape)
        self.momentum = momentum
        self.no_grad_ids = set()                                # id(Act) of inputs that need no gradient (the images)
        self.npre = 0
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        self._ws = torch.empty(1 << 16, dtype=torch.uint8, device=self.device)
        # weight gradients are off the backward's critical path (only the optimiser reads them): they run on a SIDE stream behind an event
        # that marks "this layer's dY is complete", next to the input-gradient convolutions and the (memory-bound) BatchNorm backward of the
        # layers in front -- their own workspace, their own unpack gather, the gradient all-reduce behind them.  LT_TRAIN_NO_OVERLAP=1: one stream.
        self.overlap = os.environ.get("LT_TRAIN_NO_OVERLAP") is None
        self.side = torch.cuda.Stream(device=self.device) if self.overlap else None
        self._ws2 = torch.empty(64 << 20, dtype=torch.uint8, device=self.device)         # >= the 48 MiB cap of lt_conv_wgrad's partial sums
        self._n_wgrad = 0
        # mixed precision: the weight gradients run on the bf16 MFMA too (lt_conv_wgrad_bf16 over image-octet packed operands, packed by
        # lt_pack_n8_bf16 into scratch buffers right in front of the kernel, on the stream the kernel runs on).  LT_TRAIN_WGRAD_FP32=1 keeps
        # them on the exact-fp32 MFMA (then every sixth one stays on the main stream: the side stream would be the longer one).
        self.wgrad16 = self.mixed and (self.act16 or os.environ.get("LT_TRAIN_WGRAD_FP32") is None)
        self.wgrad_main_every = int(os.environ.get("LT_TRAIN_WGRAD_MAIN_EVERY", "6" if (mixed and not self.wgrad16) else "0"))
        self._pk_main = [torch.empty(16, dtype=torch.uint8, device=self.device) for _ in range(2)]          # octet-packed dY / X of the layer in flight
        self._pk_side = [torch.empty(16, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.keep = []
        self.labels, self._label = {}, "op"              # id(closure) -> label, for profile()
        self.batched, self.fwd_jobs, self.bwd_jobs, self._job_tabs = {}, [], [], {}      # parameter gathers: one launch per replay
        # parameter gradients: ONE flat fp32 arena, slices handed out in the order the backward produces them (so a bucket of the
        # data-parallel all-reduce is a contiguous range that is complete early), 16-byte aligned
        total = sum((p.numel() + 3) // 4 * 4 for p in params if p.requires_grad)
        self.arena = torch.zeros(max(total, 4), dtype=torch.float32, device=self.device)
        self.arena_off, self.bucket_start = 0, 0
        self.param_grads = {}                                   # Parameter -> view of the arena
        self.reducer, self.bucket_elems = reducer, max(1, int(bucket_bytes) // 4)
        self.bwd_recorded = False
        # three timing events per backward and a synchronize() on the previous step's last one (backward_timing(): bench.py's N-GPU diagnosis): OFF by default --
        # they keep the host from running ahead of the GPU.  LT_TRAIN_TIMING=1 (bench.py sets it) or ``tape.timing = True``.
        self.timing = os.environ.get("LT_TRAIN_TIMING") == "1"
        if self.fp8_3d:
            # 'fp8v2v' (BASELINE config 5: "fp8 MFMA for V2V 3D convs"): e4m3 operands with per-tensor amax scaling, everything on the device -- the
            # maxima live in one pool that the forward's first op zeroes (lt_amax_* takes the maximum INTO its slot), the scales next to them
            self.pb8 = E.PlanBuilder(device, torch.float8_e4m3fn)
            self.pb8.live_weights = True
            self._fp8 = {}                                      # id(Act) -> (Act, e4m3 copy, scale slot)
            self._q_pool = torch.zeros(8192, dtype=torch.float32, device=self.device)
            self._q_used = 0
            pool, nb = self._q_pool, self._q_pool.numel() * 4
            self.do(lambda st: H.check(H.lib().lt_zero(pool.data_ptr(), nb, st), "lt_zero"), "zero")

    # ---- fp8 operands ------------------------------------------------------------------------------------------------------------
    def _q_slot(self):
        if self._q_used >= self._q_pool.numel():
            raise RuntimeError("fp8 scale pool exhausted")
        self._q_used += 1
        return self._q_pool[self._q_used - 1:self._q_used]

    def _fp8_of(self, act):
        """(e4m3 copy of ``act``, device scalar with its scale): amax over the tensor, then one quantisation pass; recorded with the first
        consumer (the producer has written ``act`` by then), shared by the later ones."""
        e = self._fp8.get(id(act))
        if e is None:
            lib = H.lib()
            t = act.t
            n, dt = t.numel(), H.dtype_code(t.dtype)
            q = torch.empty(t.shape, dtype=torch.float8_e4m3fn, device=self.device)
            am, sc = self._q_slot(), self._q_slot()
            self.keep += [t, q]
            self.do(lambda st: H.check(lib.lt_amax_dt(dt, t.data_ptr(), n, am.data_ptr(), st), "lt_amax_dt"), "amax")
            self.do(lambda st: H.check(lib.lt_quant_fp8_dt(dt, t.data_ptr(), q.data_ptr(), n, am.data_ptr(), sc.data_ptr(), st), "lt_quant_fp8_dt"), "quant fp8")
            e = self._fp8[id(act)] = (act, q, sc)
        return E.Act(e[1]), e[2]

    def _fp8_conv(self, x, wparam, idx, bias, kw):
        """A 3x3x3 / stride-1 convolution (forward, or an input gradient = the same shape over dY with the flipped filter) on the fp8 MFMA:
        x and the live weights quantised to e4m3 with per-tensor scales, the product of the two scales in the epilogue's ``scale``, the bias in its
        ``shift``; bf16 output (and bf16 residual) like every other convolution of the 16-bit-activation step."""
        lib = H.lib()
        x8, sx = self._fp8_of(x)
        y = self.pb8.conv(x8, torch.zeros(idx.shape), None, None, **kw)
        fn, info = self.pb8.ops[-1][0], self.pb8.last_info
        spec_idx = E.make_conv_spec(idx, None, None, x.shape, kw.get("stride", 1), kw.get("pad", 0), torch.float8_e4m3fn)
        assert len(spec_idx.phases) == len(info["wdev"]) == 1
        wam, sw = self._q_slot(), self._q_slot()
        wn = wparam.numel()
        self.do(lambda st: H.check(lib.lt_amax_dt(H.LT_F32, wparam.data_ptr(), wn, wam.data_ptr(), st), "lt_amax_dt"), "amax")
        ph, wdev = spec_idx.phases[0], info["wdev"][0]
        assert tuple(ph.weight.shape) == tuple(wdev.shape)
        imap = (ph.weight.round().to(torch.int32) - 1).contiguous().to(self.device)
        n8 = wdev.numel()
        self.keep += [imap, wdev]
        self.do(lambda st: H.check(lib.lt_gather_f32_fp8(wparam.data_ptr(), imap.data_ptr(), wdev.data_ptr(), n8, wam.data_ptr(), sw.data_ptr(), st), "lt_gather_f32_fp8"),
                "gather fp8")
        sc, sh = info["scale_dev"], info["shift_dev"]
        nsc = sc.numel()
        self.do(lambda st: H.check(lib.lt_scale_product(sc.data_ptr(), nsc, sx.data_ptr(), sw.data_ptr(), st), "lt_scale_product"), "scale")
        if bias is not None:          # (acc * sx sw) + bias: the bias rides in the epilogue's shift
            bmap = torch.full((sh.numel(),), -1, dtype=torch.int32)
            bmap[:bias.numel()] = torch.arange(bias.numel(), dtype=torch.int32)
            self._gather(bias, bmap.to(self.device), sh, "w")
        self.do(fn, ("dgrad fp8 " if self._cur is self.bwd_ops else "conv fp8 ") + self.pb8.ops[-1][1]["label"])
        return y

    # ---- PlanBuilder surface -------------------------------------------------------------------------------------------------
    def alloc(self, shape, dtype=None):
        return E.Act(torch.empty(shape, dtype=dtype or self.adt, device=self.device))

    def release(self, act):            # activations are needed again by the backward: nothing is recycled
        pass

    def const(self, t, dtype=None):
        return t.to(device=self.device, dtype=dtype or t.dtype).contiguous()

    def can_stem_pool(self, *a, **k):
        return False

    def can_chain_pointwise(self, *a, **k):
        return False

    def can_bottleneck(self, *a, **k):     # the training tape records every layer on its own (its backward needs the inner activations)
        return False

    def global_avgpool(self, x):
        """Mean over the map of every sample (GlobalAveragePoolingHead, pose_resnet.py:166-168): x Act [N,1,H,W,C] -> Act [1,1,1,N,C]; backward
        spreads dy / HW over the map (accumulating when the input already has a gradient)."""
        N, D, Hh, W, Cc = x.shape
        HW = D * Hh * W
        y = self.alloc((1, 1, 1, N, Cc))
        ac = self.acode
        self.do(lambda st: H.check(H.lib().lt_global_avgpool(ac, x.t.data_ptr(), y.t.data_ptr(), N, HW, Cc, st), "lt_global_avgpool"), "avgpool")

        def bwd():
            dy = self.grad_of(y)
            if dy is None:
                return
            dx = self.grad_of(x)
            acc = 1 if dx is not None else 0
            if dx is None:
                dx = torch.empty_like(x.t)
                self._add_grad(x, dx)
            self.do(lambda st: H.check(H.lib().lt_global_avgpool_bwd_dt(ac, dy.data_ptr(), dx.data_ptr(), N, HW, Cc, acc, st), "lt_global_avgpool_bwd"), "avgpool_bwd")
        self.recorders.append(bwd)
        return y

    def do(self, fn, label=None):
        """Run fn(stream) now and keep it for every later step (forward list while record() runs, backward list afterwards)."""
        fn(self.stream)
        self._cur.append(fn)
        self.labels[id(fn)] = label or self._label

    def profile(self, ops, reps=3):
        """[(label, ms)] of every recorded op in ``ops`` (fwd_ops / bwd_ops), each launch between its own pair of events on the stream
        (after a replay of the whole list, so that every buffer holds sane values); collective ops are skipped."""
        st = torch.cuda.current_stream(self.device)
        out = []
        for fn in ops:
            lab = self.labels.get(id(fn)) or "op"
            if lab == "allreduce":
                continue
            best = None
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); fn(st.cuda_stream); e1.record(st)
                e1.synchronize()
                t = e0.elapsed_time(e1)
                best = t if best is None else min(best, t)
            out.append((lab, best))
        return out

    def _ws_need(self, nbytes):
        if self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)       # closures read self._ws at call time

    def _add_grad(self, act, g):
        self.grads[id(act)] = (act, g)

    def grad_of(self, act):
        e = self.grads.get(id(act))
        return None if e is None else e[1]

    def _gather(self, src, imap, dst, group=None):
        """dst <- src[imap] every step; src is read where it lives AT CALL TIME (a Parameter the optimiser updates in place).  Gathers of
        parameters (``group`` = "w": the GEMM layouts of the forward and of the input gradients) run one by one while the tape is
        recorded and as ONE launch per replay (lt_gather_f32_multi at the head of the forward / of the backward: the weights do not
        change in between) -- 750 few-microsecond kernels per step were launch-bound."""
        self.keep += [imap, dst]
        n = dst.numel()
        if dst.dtype == torch.bfloat16:          # mixed precision: the bf16 weights straight from the fp32 Parameter
            fn = lambda st: H.check(H.lib().lt_gather_f32_bf16(src.data_ptr(), imap.data_ptr(), dst.data_ptr(), n, st), "lt_gather_f32_bf16")
        else:
            fn = lambda st: H.check(H.lib().lt_gather_f32(src.data_ptr(), imap.data_ptr(), dst.data_ptr(), n, st), "lt_gather_f32")
        self.do(fn, "gather")
        if group is not None:
            self.batched[id(fn)] = True
            (self.fwd_jobs if self._cur is self.fwd_ops else self.bwd_jobs).append((src, imap, dst, n))

    def _frag_index_map(self, plain, pack):
        """Index map of a fragment-order weight copy: ``plain`` is the layer's [cout_pad][k_pad] matrix of (1 + flat Parameter index), 0 = padding (what the
        gather of the plain GEMM layout uses); ``pack(src_ptr, dst_ptr)`` is the device routine that permutes a bf16 matrix of that shape into the fragment
        order.  The permutation is taken FROM the routine: the indices go through it as three byte planes (0..255 are exact in bf16), so whatever order a
        kernel's packer defines, the gather reproduces it -- no host restatement of any fragment layout."""
        idx = plain.round().to(torch.int64).contiguous()
        assert int(idx.max()) < (1 << 24)
        out = torch.zeros(idx.numel(), dtype=torch.int64, device=self.device)
        for sh in (0, 8, 16):
            src = ((idx >> sh) & 255).to(torch.bfloat16).to(self.device).contiguous()
            dst = torch.zeros_like(src)
            pack(src.data_ptr(), dst.data_ptr())
            out += dst.reshape(-1).float().round().to(torch.int64) << sh
        return (out - 1).to(torch.int32)

    def _cast(self, src, dst):
        n = src.numel()
        self.keep += [src, dst]
        self.do(lambda st: H.check(H.lib().lt_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), n, st), "lt_cast_f32_bf16"), "cast")

    def _bf16_of(self, act):
        if act.t.dtype == torch.bfloat16:          # act16: the activation IS the bf16 operand
            return act
        e = self._bf16.get(id(act))
        if e is None:
            t16 = torch.empty(act.t.shape, dtype=torch.bfloat16, device=self.device)
            self._bf16[id(act)] = (act, t16)
            self._cast(act.t, t16)
            return E.Act(t16)
        return E.Act(e[1])

    def _live_conv(self, x, wparam, wt=None, bias=None, out_f32=None, **kw):
        """lt_conv_fwd over the CURRENT values of ``wparam`` (optionally seen through the view transform ``wt``: the transposed / flipped
        filter of an input gradient) and of ``bias``."""
        assert wparam.numel() < (1 << 24)
        if out_f32 is None:          # fp32 storage unless the tape keeps its activations in bf16
            out_f32 = not self.act16
        idx = torch.arange(1, wparam.numel() + 1, dtype=torch.float32).reshape(wparam.shape)      # 1 + flat index; 0 = padding
        if wt is not None:
            idx = wt(idx).contiguous()
        if self.act16:
            if (self.fp8_3d and idx.dim() == 5 and tuple(idx.shape[2:]) == (3, 3, 3) and kw.get("stride", 1) == 1 and not kw.get("transposed", False)
                    and not out_f32 and x.shape[-1] >= 16 and x.shape[-1] & (x.shape[-1] - 1) == 0 and idx.shape[0] % 8 == 0):
                return self._fp8_conv(x, wparam, idx, bias, kw)
            # bf16 in, bf16 out (fp32 for the logits layer, ``out_f32``) through the ordinary epilogue of whichever bf16 kernel lt_conv_fwd picks:
            # activation flags and a bf16 residual (the input gradient that is already there) apply in the reference's order inside the kernel
            y = self.pbh.conv(x, torch.zeros(idx.shape), bias, None, out_f32=out_f32, **kw)
            fn, info = self.pbh.ops[-1][0], self.pbh.last_info
            spec_idx = E.make_conv_spec(idx, None, None, x.shape, kw.get("stride", 1), kw.get("pad", 0), torch.bfloat16, kw.get("transposed", False), 0,
                                        kw.get("output_padding", 0))
            assert len(spec_idx.phases) == len(info["wdev"])
            for ph, wdev in zip(spec_idx.phases, info["wdev"]):
                assert tuple(ph.weight.shape) == tuple(wdev.shape)
                self._gather(wparam, (ph.weight.round().to(torch.int32) - 1).contiguous().to(self.device), wdev, "w")      # fp32 Parameter -> bf16 GEMM layout
            for pi, wfr, pack in info.get("wfrag", ()):          # ... and into the fragment-order copy the fast kernels read (live_frag)
                self._gather(wparam, self._frag_index_map(spec_idx.phases[pi].weight, pack), wfr.reshape(-1), "w")
                self.n_frag_layers = getattr(self, "n_frag_layers", 0) + 1
            if bias is not None:
                bi = info["bias_dev"]
                bmap = torch.full((bi.numel(),), -1, dtype=torch.int32)
                bmap[:bias.numel()] = torch.arange(bias.numel(), dtype=torch.int32)
                self._gather(bias, bmap.to(self.device), bi, "w")
            self.do(fn, ("dgrad " if self._cur is self.bwd_ops else "conv ") + self.pbh.ops[-1][1]["label"])
            return y
        if self.mixed:
            residual = kw.pop("residual", None)
            if residual is not None and (kw.get("relu") or kw.get("relu_pre")):
                # the bf16 kernel's epilogue would apply the activation BEFORE the separate fp32 residual add below: relu(v) + res, not
                # relu(v + res).  No layer of these networks takes this path (the residual convolutions carry BatchNorm); refuse it loudly.
                raise NotImplementedError("mixed precision: a BatchNorm-less convolution with both an activation and a residual")
            x16 = self._bf16_of(x)
            spec_idx = E.make_conv_spec(idx, None, None, x16.shape, kw.get("stride", 1), kw.get("pad", 0), torch.bfloat16, kw.get("transposed", False), 0,
                                        kw.get("output_padding", 0))
            # the fp32 gradient that is already there rides in the epilogue (LT_EPI_RES_F32) -- except on the 3^3 32 -> 32 layers, whose column-walk
            # kernel has no fp32 residual path and is worth more than the saved pass (LT_TRAIN_NO_FUSED_RES=1: always the separate pass)
            col_walk = tuple(idx.shape) == (32, 32, 3, 3, 3) and kw.get("stride", 1) == 1 and not kw.get("transposed", False)
            fuse_res = residual is not None and not col_walk and os.environ.get("LT_TRAIN_NO_FUSED_RES") is None
            if fuse_res:
                kw = dict(kw, residual=residual, residual_f32=True)
            y = self.pbh.conv(x16, torch.zeros(idx.shape), bias, None, out_f32=out_f32, **kw)
            fn, info = self.pbh.ops[-1][0], self.pbh.last_info
            assert len(spec_idx.phases) == len(info["wdev"])
            for ph, wdev in zip(spec_idx.phases, info["wdev"]):
                assert tuple(ph.weight.shape) == tuple(wdev.shape)
                self._gather(wparam, (ph.weight.round().to(torch.int32) - 1).contiguous().to(self.device), wdev, "w")      # fp32 Parameter -> bf16 GEMM layout
            if bias is not None:
                bi = info["bias_dev"]
                bmap = torch.full((bi.numel(),), -1, dtype=torch.int32)
                bmap[:bias.numel()] = torch.arange(bias.numel(), dtype=torch.int32)
                self._gather(bias, bmap.to(self.device), bi, "w")
            self.do(fn, ("dgrad " if self._cur is self.bwd_ops else "conv ") + self.pbh.ops[-1][1]["label"])
            if residual is not None and not fuse_res:
                yt, rt = y.t, residual.t
                n_add = yt.numel()
                self.do(lambda st: H.check(H.lib().lt_add_f32(yt.data_ptr(), rt.data_ptr(), n_add, st), "lt_add_f32"), "add")
            return y
        y = self.pb.conv(x, idx, bias, None, **kw)
        fn, info = self.pb.ops[-1][0], self.pb.last_info
        for wdev in info["wdev"]:
            self._gather(wparam, (wdev.round().to(torch.int32) - 1).contiguous(), wdev, "w")
        if bias is not None:
            bi = info["bias_dev"]
            bmap = torch.full((bi.numel(),), -1, dtype=torch.int32)
            bmap[:bias.numel()] = torch.arange(bias.numel(), dtype=torch.int32)
            self._gather(bias, bmap.to(self.device), bi, "w")
        self.do(fn, ("dgrad " if self._cur is self.bwd_ops else "conv ") + self.pb.ops[-1][1]["label"])
        return y

    # ---- layers --------------------------------------------------------------------------------------------------------------
    def conv(self, x, weight, bias=None, bn=None, stride=1, pad=0, transposed=False, relu=False, relu_pre=False, residual=None, out_f32=False,
             out=None, sigmoid=False):
        lib = H.lib()
        flags = (H.EPI_RELU_POST if relu else 0) | (H.EPI_RELU_PRE if relu_pre else 0) | (H.EPI_SIGMOID if sigmoid else 0)
        if sigmoid and (bn is not None or relu or relu_pre or residual is not None):
            raise NotImplementedError("a sigmoid epilogue next to BatchNorm / ReLU / a residual (the confidence heads end in Linear + Sigmoid only)")
        if bn is None:          # convolution (+ bias) -> activation in the conv epilogue, as in inference
            if relu_pre and residual is not None:
                # lt_act_bwd would rebuild the mask of z = relu(v) + res as (z - res) > 0, which drops a live gradient when 0 < relu(v) < ulp(res) / 2
                # (ADVICE r2); the layers of these networks that add a residual behind a ReLU all carry BatchNorm (lt_bn_act_bwd recomputes v)
                raise NotImplementedError("a BatchNorm-less layer with ReLU before a residual add")
            kw = dict(stride=stride, pad=pad, transposed=transposed, relu=relu, relu_pre=relu_pre, residual=residual)
            if sigmoid:
                if self.act16 and not out_f32:
                    raise NotImplementedError("train_precision 'act16': a sigmoid layer that stores bf16 (the confidence heads store fp32)")
                kw["sigmoid"] = True
            z = self._live_conv(x, weight, None, bias, out_f32=True if (out_f32 or not self.act16) else False, **kw)
            y_raw = stats = None
        else:
            y16 = self.y16 and weight.shape[1 if transposed else 0] % 8 == 0
            if self.act16 and not y16:
                raise NotImplementedError("train_precision 'act16': a BatchNorm layer whose channel count is no multiple of 8")
            y_raw = self._live_conv(x, weight, None, bias, out_f32=not y16, stride=stride, pad=pad, transposed=transposed)
            gamma, beta, rmean, rvar = bn
            Cc = y_raw.shape[-1]
            rows = y_raw.t.numel() // Cc
            ydt = H.LT_BF16 if y16 else H.LT_F32
            if y16:
                assert y_raw.t.dtype == torch.bfloat16
                flags |= H.BN_Y_BF16
            if getattr(bn, "training", True):
                mean = torch.empty(Cc, dtype=torch.float32, device=self.device)
                var = torch.empty(Cc, dtype=torch.float32, device=self.device)
                self._ws_need(lib.lt_bn_stats_workspace(rows, Cc))
                mom = float(self.momentum)
                self.do(lambda st: H.check(lib.lt_bn_stats_fwd(ydt, y_raw.t.data_ptr(), rows, Cc, mean.data_ptr(), var.data_ptr(), rmean.data_ptr(),
                                                               rvar.data_ptr(), mom, self._ws.data_ptr(), st), "lt_bn_stats_fwd"), "bn_stats %dx%d" % (rows, Cc))
            else:
                # a BatchNorm module left in eval() inside a training step (frozen statistics, e.g. a frozen backbone): normalise with the LIVE
                # running statistics (read where they are at every replay), no statistics pass, no update; the backward drops the batch terms
                mean, var = rmean, rvar
                flags |= H.BN_FROZEN
            z = self.alloc(y_raw.shape)
            rp = residual.t if residual is not None else None
            z16 = None
            flags |= self.aflag          # act16: z and the residual are bf16 tensors
            if self.mixed and not self.act16 and Cc % 4 == 0:          # the next convolution's bf16 operand is written on the way (no separate cast pass)
                z16 = torch.empty(z.t.shape, dtype=torch.bfloat16, device=self.device)
                self._bf16[id(z)] = (z, z16)
            self.do(lambda st: H.check(lib.lt_bn_act_fwd(y_raw.t.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), H.ptr(rp),
                                                         z.t.data_ptr(), H.ptr(z16), rows, Cc, BN_EPS, flags, st), "lt_bn_act_fwd"), "bn_act %dx%d" % (rows, Cc))
            stats = (mean, var)
        self.recorders.append(lambda: self._conv_bwd(x, weight, bias, bn, stride, pad, transposed, flags, residual, y_raw, stats, z))
        self.layer_outputs.append(("%s %s" % ("deconv" if transposed else "conv", "x".join(map(str, weight.shape))), z))      # diagnostics (tools/mixed_trace.py)
        return z

    def maxpool(self, x, k, s, p, nd):
        pb = self.pbh if self.act16 else self.pb
        y = pb.maxpool(x, k, s, p, nd)
        self.do(pb.ops[-1][0])
        kk = (1, k, k) if nd == 2 else (k, k, k)
        ss = (1, s, s) if nd == 2 else (s, s, s)
        pp = (0, p, p) if nd == 2 else (p, p, p)

        def bwd():
            dy = self.grad_of(y)
            if dy is None:
                return
            dx = self.grad_of(x)
            if dx is None:
                dx = torch.empty_like(x.t)
                self._add_grad(x, dx)
                nb = dx.numel() * dx.element_size()
                self.do(lambda st: H.check(H.lib().lt_zero(dx.data_ptr(), nb, st), "lt_zero"), "zero")
            N, D, Hh, W, Cc = x.shape
            ac = self.acode
            self.do(lambda st: H.check(H.lib().lt_maxpool_bwd_dt(ac, x.t.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, D, Hh, W, Cc, H.i3(kk), H.i3(ss), H.i3(pp), st),
                                       "lt_maxpool_bwd"))
        self.recorders.append(bwd)
        return y

    # ---- parameter gradients -----------------------------------------------------------------------------------------------------
    def _grad_view(self, p):
        """Slice of the gradient arena for parameter ``p``; a same-size VIEW of a Parameter (a Linear weight handed over as a 1x1 convolution
        filter, ``w[:, :, None, None]``) lands in its base Parameter's entry -- same memory order, the base's shape."""
        base = p._base if (getattr(p, "_base", None) is not None and p._base.numel() == p.numel() and p.is_contiguous() and p._base.is_contiguous()) else p
        if base in self.param_grads:
            raise NotImplementedError("a parameter shared by two layers (not the case in these networks)")
        n, off = p.numel(), self.arena_off
        if off + n > self.arena.numel():
            raise RuntimeError("parameter-gradient arena too small: pass every trainable parameter to TrainTape(params=...)")
        self.arena_off += (n + 3) // 4 * 4
        v = self.arena[off:off + n]
        self.param_grads[base] = v.view(base.shape)
        return v.view(p.shape)

    def _grads_ready(self, final=False):
        """Called after the op that completes a parameter gradient has been recorded: a full bucket starts its all-reduce here, behind
        the kernels that produced it and in front of the rest of the backward (lt_dist.GradReducer)."""
        if self.reducer is None:
            return
        a, b = self.bucket_start, self.arena_off
        if b > a and (final or b - a >= self.bucket_elems):
            chunk = self.arena[a:b]

            rev = torch.cuda.Event() if self.side is not None else None

            def reduce(st):
                if self.side is not None:          # behind the weight-gradient kernels that complete the bucket, on either stream
                    rev.record(torch.cuda.current_stream(self.device))
                    self.side.wait_event(rev)
                    with torch.cuda.stream(self.side):
                        self.reducer.reduce_inplace(chunk)
                else:
                    self.reducer.reduce_inplace(chunk)
            self.do(reduce, "allreduce")
            self.bucket_start = b

    # ---- backward of one convolution layer ---------------------------------------------------------------------------------------
    def _conv_bwd(self, x, weight, bias, bn, stride, pad, transposed, flags, residual, y_raw, stats, z):
        lib = H.lib()
        dz = self.grad_of(z)
        if dz is None:
            return
        Cout = z.shape[-1]
        rows = z.t.numel() // Cout
        dy = torch.empty(z.t.shape, dtype=self.adt, device=self.device)
        dres, acc_res = None, 0
        rp = residual.t if residual is not None else None
        if residual is not None:
            dres = self.grad_of(residual)
            acc_res = 1 if dres is not None else 0
            if dres is None:
                dres = torch.empty_like(residual.t)
                self._add_grad(residual, dres)
        if bn is not None:
            gamma, beta, _, _ = bn
            mean, var = stats
            dgamma, dbeta = self._grad_view(gamma), self._grad_view(beta)
            self._ws_need(lib.lt_bn_act_bwd_workspace(rows, Cout))
            dy16 = None
            if self.mixed and not self.act16 and id(x) not in self.no_grad_ids and Cout >= 4 and Cout & (Cout - 1) == 0:      # the input gradient's bf16 operand on the way
                dy16 = torch.empty(dy.shape, dtype=torch.bfloat16, device=self.device)
            self.do(lambda st: H.check(lib.lt_bn_act_bwd(dz.data_ptr(), y_raw.t.data_ptr(), H.ptr(rp), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
                                                         beta.data_ptr(), dy.data_ptr(), H.ptr(dy16), dgamma.data_ptr(), dbeta.data_ptr(), H.ptr(dres), acc_res, rows,
                                                         Cout, BN_EPS, flags, self._ws.data_ptr(), st), "lt_bn_act_bwd"), "bn_bwd %dx%d" % (rows, Cout))
        else:
            total = z.t.numel()
            if self.act16 and z.t.dtype != torch.bfloat16:
                # a layer of the 16-bit step that STORES fp32 (the logits / heatmap layers, the confidence heads' sigmoid outputs: their consumers -- soft-argmax,
                # unprojection, the autograd tail -- are fp32): its upstream gradient arrives in fp32 too; activation backward in fp32, then ONE rounding of dy
                if residual is not None or dz.dtype != torch.float32:
                    raise NotImplementedError("train_precision 'act16': an fp32-storing layer with a residual / a bf16 upstream gradient")
                dy32 = torch.empty(z.t.shape, dtype=torch.float32, device=self.device)
                self.do(lambda st: H.check(lib.lt_act_bwd(dz.data_ptr(), z.t.data_ptr(), None, dy32.data_ptr(), None, 0, total, flags, st), "lt_act_bwd"))
                self.do(lambda st: H.check(lib.lt_convert_pad(H.LT_F32, dy32.data_ptr(), H.LT_BF16, dy.data_ptr(), rows, Cout, Cout, st), "lt_convert_pad"), "cast")
            else:
                self.do(lambda st: H.check(lib.lt_act_bwd(dz.data_ptr(), z.t.data_ptr(), H.ptr(rp), dy.data_ptr(), H.ptr(dres), acc_res, total, flags | self.aflag, st), "lt_act_bwd"))
        bias_sum = None
        if bias is not None and bias.requires_grad:
            db = self._grad_view(bias)
            cs_need = lib.lt_channel_sum_workspace(rows, Cout)
            ac = self.acode
            bias_sum = lambda st, ws: H.check(lib.lt_channel_sum_dt(ac, dy.data_ptr(), rows, Cout, db.data_ptr(), 0, ws, st), "lt_channel_sum")
            if not (weight.requires_grad and self.overlap):          # (with a side stream: in front of the layer's weight gradient over there -- only Adam reads it)
                self._ws_need(cs_need)
                self.do(lambda st, fn=bias_sum: fn(st, self._ws.data_ptr()))          # (fn bound now: the name is cleared below and the closure is replayed later)
                bias_sum = None
        nd = weight.dim() - 2
        st3 = ((1,) + (stride,) * 2) if nd == 2 else (stride,) * 3
        pd3 = ((0,) + (pad,) * 2) if nd == 2 else (pad,) * 3
        ks = tuple(weight.shape[2:])
        ks3 = ((1,) + ks) if nd == 2 else ks
        taps_all = torch.tensor([(a, b, c, 0) for a in range(ks3[0]) for b in range(ks3[1]) for c in range(ks3[2])], dtype=torch.int32, device=self.device)
        ntaps = taps_all.shape[0]
        N, D, Hh, W, cin_buf = x.shape
        # lt_conv_fwd wants a power-of-two channel count on its input: dY widened with zero channels (17 joints -> 32) for the input gradient
        # (and, in the 16-bit step, for the octet pack of the weight gradient)
        dy_in, cpad = dy, Cout
        if (Cout & (Cout - 1) or Cout < 4) and (self.act16 or id(x) not in self.no_grad_ids):
            cpad = max(4, 1 << (Cout - 1).bit_length())
            if self.act16:
                cpad = max(8, cpad)
            dyp = torch.empty(*dy.shape[:-1], cpad, dtype=self.adt, device=self.device)
            ac_ = self.acode
            self.do(lambda st: H.check(lib.lt_convert_pad(ac_, dy.data_ptr(), ac_, dyp.data_ptr(), rows, Cout, cpad, st), "lt_convert_pad"), "pad")
            dy_in = dyp                     # (dy itself must keep its name: the closures above read it when they are replayed)
        if weight.requires_grad:
            # ---- weight gradient: dw[co][tap * Cin + ci] over the GEMM rows of the forward convolution, then into the Parameter's layout
            if not transposed:
                cop, kp = E.cout_pad_of(Cout), ntaps * cin_buf
                a_ptr, b_ptr = dy, x.t
                geo = (N, D, Hh, W, cin_buf, z.shape[1], z.shape[2], z.shape[3], Cout, Cout)
                wrows = rows
                ar = torch.arange(cop * kp, dtype=torch.int32).reshape(cop, kp)
                imap = ar[:Cout].reshape(Cout, *ks3, cin_buf)[..., :weight.shape[1]].permute(0, 4, 1, 2, 3)
            else:
                # ConvTranspose: out[2 q - p + a] += x[q][ci] W[ci][co][a]  ->  dW[ci][a][co] = sum_q x[q][ci] dY[2 q - p + a][co]: the same
                # kernel with the layer's INPUT in the role of "dy" (rows = input pixels) and the output gradient in the role of "x"
                cin_t = weight.shape[0]
                cop, kp = E.cout_pad_of(cin_t), ntaps * Cout
                a_ptr, b_ptr = x.t, dy
                oN, oD, oH, oW, _ = z.shape
                geo = (oN, oD, oH, oW, Cout, D, Hh, W, cin_t, cin_buf)
                wrows = N * D * Hh * W
                ar = torch.arange(cop * kp, dtype=torch.int32).reshape(cop, kp)
                imap = ar[:cin_t].reshape(cin_t, *ks3, Cout).permute(0, 4, 1, 2, 3)
            imap = (imap[:, :, 0] if nd == 2 else imap).contiguous().reshape(-1).to(self.device)
            dw = torch.empty(cop, kp, dtype=torch.float32, device=self.device)
            use16, direct = self.wgrad16, False
            n_img = geo[0]
            pa_px, pb_px = geo[5] * geo[6] * geo[7], geo[1] * geo[2] * geo[3]          # pixels per image of the dY-role / X-role tensor
            if use16:
                need = lib.lt_conv_wgrad_bf16_workspace((n_img + 7) // 8 * pa_px, cop, kp)
                pk_need = (lib.lt_pack_n8_bf16_bytes(n_img, pa_px, geo[9]), lib.lt_pack_n8_bf16_bytes(n_img, pb_px, geo[4]))
                # where the bf16 copy the convolutions read exists already (same layout), the pack reads that: 2 instead of 4 bytes per element
                if self.act16:          # the activations / gradients are the bf16 operands (dY through its zero-padded copy when Cout % 8)
                    if cin_buf % 8 or (transposed and Cout % 8):
                        raise NotImplementedError("train_precision 'act16': a weight gradient over %d input channels" % cin_buf)
                    if transposed and dy_in is not dy:
                        # dY was widened to ``cpad`` channels (Cout a multiple of 8 but no power of two): the transposed layer's kernels are told geo[4] = Cout as
                        # channel count AND leading dimension of the X-role operand, which the widened copy does not have -- silently wrong gradients.
                        # No layer of V2V / pose_resnet has this shape; refuse it instead of guessing.
                        raise NotImplementedError("train_precision 'act16': the weight gradient of a transposed convolution with %d output channels "
                                                  "(not a power of two: dY is read through a zero-padded copy with another leading dimension)" % Cout)
                    x16, d16 = x.t, dy_in
                    if not transposed and dy_in is not dy:
                        geo = geo[:8] + (cpad, cpad)          # Cout, ldy of the widened dY (cout_pad_of(Cout) == cpad rows of dw, the extra ones zero)
                        assert cop >= cpad
                    pk_need = (lib.lt_pack_n8_bf16_bytes(n_img, pa_px, geo[9]), lib.lt_pack_n8_bf16_bytes(n_img, pb_px, geo[4]))
                else:
                    x16e = self._bf16.get(id(x))
                    x16 = x16e[1] if x16e is not None and x16e[1].shape == x.t.shape and cin_buf % 8 == 0 else None
                    d16 = dy16 if (bn is not None and dy16 is not None and Cout % 8 == 0) else None
                a16, b16 = (d16, x16) if not transposed else (x16, d16)
                self.keep += [t for t in (a16, b16) if t is not None]
                # both operands there as channels-last bf16 tensors (always in the 16-bit step) and a shape of the kernel that transposes in registers:
                # no octet packs (LT_WGRAD16_PACKED=1: the packed kernels everywhere)
                direct = bool(a16 is not None and b16 is not None and lib.lt_conv_wgrad_bf16_nhwc_ok(
                    geo[0], geo[1], geo[2], geo[3], geo[4], geo[4], geo[5], geo[6], geo[7], H.i3(st3), H.i3(pd3), geo[8], geo[9], cop, kp, ntaps))
            else:
                need = lib.lt_conv_wgrad_workspace(wrows, cop, kp)
            self.keep += [taps_all, a_ptr, b_ptr, imap, dw]
            gview = self._grad_view(weight)
            ng = gview.numel()
            label = "wgrad %s %s rows %d cout_pad %d K %d" % ("x".join(map(str, weight.shape)), "T" if transposed else "", wrows, cop, kp)
            # load balance of the two streams: with the convolutions on the bf16 MFMA the main stream's backward (input gradients + BatchNorm)
            # is shorter than the side stream's fp32 weight gradients, so every k-th weight gradient stays on the main stream
            self._n_wgrad += 1
            on_main = self.wgrad_main_every > 0 and self._n_wgrad % self.wgrad_main_every == 0
            ev = torch.cuda.Event() if (self.overlap and not on_main) else None
            if bias_sum is not None:
                need = max(need, cs_need)
            if ev is None:          # on the main stream: the main stream's workspace (the side stream's may be in use by a concurrent weight gradient)
                self._ws_need(need)
            elif self._ws2.numel() < need:
                self._ws2.record_stream(self.side)          # a side-stream kernel of the recording step may still be using it: not to be handed out before that
                self._ws2 = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            if use16 and not direct:
                pk = self._pk_main if ev is None else self._pk_side
                for i in range(2):
                    if pk[i].numel() < pk_need[i]:
                        if ev is not None:
                            pk[i].record_stream(self.side)
                        pk[i] = torch.empty(int(pk_need[i]), dtype=torch.uint8, device=self.device)

            def wgrad(st):
                if ev is not None:          # dY (and this layer's BatchNorm / bias gradients) are complete on the main stream: the side stream may go
                    ev.record(torch.cuda.current_stream(self.device))
                    self.side.wait_event(ev)
                    st = self.side.cuda_stream
                ws = (self._ws if ev is None else self._ws2).data_ptr()
                if bias_sum is not None:
                    bias_sum(st, ws)
                if use16 and direct:
                    H.check(lib.lt_conv_wgrad_bf16_nhwc(a16.data_ptr(), b16.data_ptr(), taps_all.data_ptr(), dw.data_ptr(), geo[0], geo[1], geo[2], geo[3], geo[4], geo[4],
                                                        geo[5], geo[6], geo[7], H.i3(st3), H.i3(pd3), geo[8], geo[9], cop, kp, ntaps, 0, ws, st), "lt_conv_wgrad_bf16_nhwc")
                elif use16:
                    pk = self._pk_main if ev is None else self._pk_side
                    for src, src16, dst, px, ch in ((a_ptr, a16, pk[0], pa_px, geo[9]), (b_ptr, b16, pk[1], pb_px, geo[4])):
                        if src16 is not None:
                            H.check(lib.lt_pack_n8_from_bf16(src16.data_ptr(), dst.data_ptr(), n_img, px, ch, ch, st), "lt_pack_n8_from_bf16")
                        else:
                            H.check(lib.lt_pack_n8_bf16(src.data_ptr(), dst.data_ptr(), n_img, px, ch, ch, st), "lt_pack_n8_bf16")
                    H.check(lib.lt_conv_wgrad_bf16(pk[0].data_ptr(), pk[1].data_ptr(), taps_all.data_ptr(), dw.data_ptr(), geo[0], geo[1], geo[2], geo[3], geo[4],
                                                   geo[5], geo[6], geo[7], H.i3(st3), H.i3(pd3), geo[8], geo[9], cop, kp, ntaps, 0, ws, st), "lt_conv_wgrad_bf16")
                else:
                    H.check(lib.lt_conv_wgrad(a_ptr.data_ptr(), b_ptr.data_ptr(), taps_all.data_ptr(), dw.data_ptr(), geo[0], geo[1], geo[2], geo[3], geo[4], geo[5],
                                              geo[6], geo[7], H.i3(st3), H.i3(pd3), geo[8], geo[9], cop, kp, ntaps, 0, ws, st), "lt_conv_wgrad")
                H.check(lib.lt_gather_f32(dw.data_ptr(), imap.data_ptr(), gview.data_ptr(), ng, st), "lt_gather_f32")      # into the Parameter's layout
            self.do(wgrad, label)
        self._grads_ready()
        # ---- input gradient (skipped for the network input)
        if id(x) in self.no_grad_ids:
            return
        prev = self.grad_of(x)
        dya = E.Act(dy_in)
        if bn is not None and not self.act16 and dy16 is not None:
            self._bf16[id(dya)] = (dya, dy16)
        res = E.Act(prev) if prev is not None else None
        if not transposed and stride == 1:
            # correlation with the flipped, transposed filter [Cin, Cout, k..]
            dx = self._live_conv(dya, weight, lambda t: t.transpose(0, 1).flip(*range(2, 2 + nd)), None, stride=1, pad=ks[0] - 1 - pad, residual=res)
        elif not transposed:
            assert stride == 2 and all(d % 2 == 0 for d in ((Hh, W) if nd == 2 else (D, Hh, W))), "stride-2 layers need even input sizes"
            # the adjoint of a strided convolution is the transposed convolution with the SAME weight tensor read as [in = Cout, out = Cin]
            dx = self._live_conv(dya, weight, None, None, stride=2, pad=pad, transposed=True, output_padding=1, residual=res)
        else:
            # the adjoint of ConvTranspose(weight [Cin, Cout, k..], stride 2) is Conv(stride 2) with that tensor read as [out = Cin, in = Cout]
            dx = self._live_conv(dya, weight, None, None, stride=2, pad=pad, residual=res)
        g = dx.t
        if g.shape[-1] != cin_buf:
            raise RuntimeError("input gradient has %d channels, the activation %d" % (g.shape[-1], cin_buf))
        self._add_grad(x, g)

    # ---- driver ---------------------------------------------------------------------------------------------------------------------
    def seed(self, act, grad):
        """Gradient buffer of the loss with respect to an output Act of the recorded forward (written before every backward)."""
        assert grad.is_contiguous()
        self._add_grad(act, grad)

    def add_backward(self, fn):
        """A custom op's backward (unprojection): fn() reads grad_of(output), registers the inputs' gradients and records its launches
        with ``do``."""
        self.recorders.append(fn)

    JOB = np.dtype([("src", "u8"), ("idx", "u8"), ("dst", "u8"), ("n", "i8"), ("fb", "i4"), ("pad", "i4")])

    def _gather_all(self, which, jobs):
        """One lt_gather_f32_multi launch for ``jobs``; the table is rebuilt only when a Parameter's storage has moved."""
        if not jobs:
            return
        key = tuple(j[0].data_ptr() for j in jobs)
        tab = self._job_tabs.get(which)
        if tab is None or tab[0] != key:
            arr = np.zeros(len(jobs), dtype=self.JOB)
            fb = 0
            for i, (src, imap, dst, n) in enumerate(jobs):
                arr[i] = (src.data_ptr(), imap.data_ptr(), dst.data_ptr(), n, fb, 1 if dst.dtype == torch.bfloat16 else 0)
                fb += (n + 1023) // 1024
            dev = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)
            tab = self._job_tabs[which] = (key, dev, len(jobs), fb)
        H.check(H.lib().lt_gather_f32_multi(tab[1].data_ptr(), tab[2], tab[3], self.stream), "lt_gather_f32_multi")

    def replay(self, ops, start=0, stop=None):
        for fn in ops[start:stop]:
            if id(fn) not in self.batched:
                fn(self.stream)

    def run_forward(self):
        """Replay of the recorded forward on the current stream (the first forward IS the recording)."""
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        self._gather_all("fwd", self.fwd_jobs)
        self.replay(self.fwd_ops)

    def run_backward(self):
        """First call: records (= runs) the backward; later calls replay it.  Returns {Parameter: gradient view of the arena} (averaged
        over the ranks when a reducer is attached)."""
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        main = torch.cuda.current_stream(self.device)
        timing = self.timing
        if timing:
            self._collect_bwd_times()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]          # start, main stream's kernels issued, side stream joined
            self._bwd_events = ev if self.bwd_recorded else None          # (the recording backward -- index maps, allocations, host work -- is not a sample)
            ev[0].record(main)
        if not self.bwd_recorded:
            self._cur = self.bwd_ops
            for rec in reversed(self.recorders):
                rec()
            self._grads_ready(final=True)
            self.recorders, self.bwd_recorded = [], True
        else:
            self._gather_all("bwd", self.bwd_jobs)
            self.replay(self.bwd_ops)
        if timing:
            ev[1].record(main)
        if self.side is not None:
            if self.reducer is not None:
                with torch.cuda.stream(self.side):
                    self.reducer.wait_all()
            torch.cuda.current_stream(self.device).wait_stream(self.side)          # the gradients are complete for whoever reads them next
        elif self.reducer is not None:
            self.reducer.wait_all()
        if timing:
            ev[2].record(main)
        return self.param_grads

    def _collect_bwd_times(self):
        ev = getattr(self, "_bwd_events", None)
        if ev is not None:
            ev[2].synchronize()
            t = self.__dict__.setdefault("_bwd_times", [0.0, 0.0, 0])
            t[0] += ev[0].elapsed_time(ev[1]); t[1] += ev[0].elapsed_time(ev[2]); t[2] += 1
            self._bwd_events = None

    def backward_timing(self):
        """Average GPU time of the recorded backward: on the main stream alone (input gradients, BatchNorm, ...) and until the side stream -- the
        weight gradients and, with a reducer, the gradient all-reduce -- has joined: the difference is what the exchange (and the weight
        gradients) leave EXPOSED behind the backward."""
        self._collect_bwd_times()
        t = self.__dict__.get("_bwd_times")
        if not t or not t[2]:
            return None
        return {"backward_main_stream_ms": t[0] / t[2], "backward_until_gradients_complete_ms": t[1] / t[2], "exposed_behind_main_stream_ms": (t[1] - t[0]) / t[2]}


def adam_groups(model, config_opt):
    """The reference's three parameter groups for the volumetric model (train.py:430-437): backbone at ``lr``, ``process_features`` and
    ``volume_net`` at their own rates when the config names them (``process_features_lr`` / ``volume_net_lr``), else at ``lr``."""
    lr = config_opt.lr
    return [{"params": list(model.backbone.parameters())},
            {"params": list(model.process_features.parameters()), "lr": getattr(config_opt, "process_features_lr", lr)},
            {"params": list(model.volume_net.parameters()), "lr": getattr(config_opt, "volume_net_lr", lr)}]


class Adam:
    """torch.optim.Adam's update through liblt_hip, same hyper-parameter surface for what the reference uses (train.py:430-437: lr per
    group, default betas / eps, no weight decay).  All tensors of a group with the same betas / eps / weight decay are updated by ONE
    launch (lt_adam_step_multi) driven by a small job table that is rebuilt and uploaded every step (the gradients are new tensors)."""
    JOB = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i8"), ("lr", "f4"), ("fb", "i4")])

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        groups = params if params and isinstance(params[0], dict) else [{"params": params}]
        self.param_groups = [dict({"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}, **dict(g, params=list(g["params"]))) for g in groups]
        self.state = {}
        self.steps = 0
        self._tables = {}

    def zero_grad(self):
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None

    def step(self):
        lib = H.lib()
        self.steps += 1
        batches = {}          # (betas, eps, weight decay, device, the parameters' own step count) -> jobs
        touched = []
        fresh = []            # parameters seen for the first time: their moments come out of ONE zero-filled allocation
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None or not p.requires_grad:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32:
                    raise RuntimeError("lt_train.Adam updates fp32 parameters on the GPU")
                if not p.is_contiguous():
                    raise RuntimeError("lt_train.Adam updates parameters in place through their data pointer: a non-contiguous parameter "
                                       "(shape %s, strides %s) is not supported" % (tuple(p.shape), p.stride()))
                if p not in self.state:
                    fresh.append(p)
        for dev in {p.device for p in fresh}:
            ps = [p for p in fresh if p.device == dev]
            flat = torch.zeros(2 * sum((p.numel() + 3) // 4 * 4 for p in ps), dtype=torch.float32, device=dev)
            o = 0
            for p in ps:
                n, npad = p.numel(), (p.numel() + 3) // 4 * 4
                self.state[p] = {"m": flat[o:o + n].view(p.shape), "v": flat[o + npad:o + npad + n].view(p.shape), "step": 0}
                o += 2 * npad
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None or not p.requires_grad:
                    continue
                st = self.state[p]
                st["step"] += 1           # torch.optim.Adam's bias correction uses the PARAMETER's own step count (a parameter whose first
                                          # gradient arrives later -- an unfrozen layer -- starts its correction there)
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), p.device, st["step"])
                batches.setdefault(key, []).append((p, grad, st, float(g["lr"])))
                touched.append(p)
        self._tables_used = []
        for (b1, b2, eps, wd, dev, pstep), items in batches.items():
            tab = np.zeros(len(items), dtype=self.JOB)
            fb = 0
            for i, (p, grad, st, lr_) in enumerate(items):
                tab[i] = (p.data_ptr(), grad.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), p.numel(), lr_, fb)
                fb += (p.numel() + 1023) // 1024
            tkey = (dev, len(items), len(self._tables_used))          # one staging slot per launch of this step() (several step-count groups may share a size)
            self._tables_used.append(tkey)
            slot = self._tables.get(tkey)
            if slot is None:
                slot = self._tables[tkey] = (torch.empty(tab.nbytes, dtype=torch.uint8).pin_memory(), torch.empty(tab.nbytes, dtype=torch.uint8, device=dev),
                                                          torch.cuda.Event())
            host, devt, ev = slot
            ev.synchronize()                       # the previous step's upload of this table has been consumed
            host.numpy()[:] = tab.view(np.uint8)
            with torch.cuda.device(dev):
                devt.copy_(host, non_blocking=True)
                ev.record()
                H.check(lib.lt_adam_step_multi(devt.data_ptr(), len(items), fb, b1, b2, eps, wd, pstep, torch.cuda.current_stream(dev).cuda_stream),
                        "lt_adam_step_multi")
        for p in touched:          # version counters: cached inference plans see that the weights have changed
            torch.autograd.graph.increment_version(p)
