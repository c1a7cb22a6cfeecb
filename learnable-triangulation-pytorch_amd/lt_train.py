"""Training-mode execution of the reference-shaped networks (BASELINE config 5): the modules'
``record()`` methods -- the same code that records the inference plan -- run against a ``TrainTape`` that EXECUTES every layer at
once in training mode (convolution without a folded BatchNorm -> batch statistics -> normalise + ReLU + residual) and remembers
what the backward needs; ``TrainTape.backward()`` then walks the layers in reverse:

    activation / BatchNorm backward   lt_bn_act_bwd / lt_act_bwd   (csrc/train.hip)
    bias gradient                     lt_channel_sum
    weight gradient                   lt_conv_wgrad               (exact-fp32 MFMA, no atomics)
    input gradient                    lt_conv_fwd over dY with the weights transposed + flipped (stride 1), as a parity-phase transposed
                                      convolution (stride-2 layers: k=3/p=1 or k=1/p=0 with output_padding 1) or as the strided convolution a
                                      transposed layer is the adjoint of; an already existing gradient of the input rides in as the
                                      epilogue's residual, so sums over consumers cost no extra pass
    max pool                          lt_maxpool_bwd

Everything is fp32 (the reference trains in fp32); all arithmetic is liblt_hip's, torch only owns the memory.  What the reference
does with ``total_loss.backward()`` (train.py:233-236) arrives here through the autograd Function that wraps a network's training
forward (mvn/models/triangulation.py), so ``torch.optim`` / DDP hooks see ordinary ``.grad`` tensors.
"""
import ctypes as C

import numpy as np
import torch

import lt_engine as E
import lt_hip as H

BN_EPS = 1e-5


class TrainTape:
    """PlanBuilder-shaped object for ``record()``: conv / maxpool / alloc / release; executes eagerly on ``stream``."""

    def __init__(self, device, momentum=0.1):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("training runs on the GPU only (device=%s); there is no CPU fallback" % device)
        self.dtype, self.code, self.dry_run = torch.float32, H.LT_F32, False
        self.pb = E.PlanBuilder(device, torch.float32)          # builds the lt_conv_fwd descriptors; every op is run as it is recorded
        self.tape = []                                          # backward closures, run in reverse
        self.grads = {}                                         # id(Act) -> (Act, gradient tensor of act.t's shape)
        self.param_grads = {}                                   # Parameter -> gradient in the Parameter's own layout
        self.momentum = momentum
        self.no_grad_ids = set()                                # id(Act) of inputs that need no gradient (the images)
        self.npre = 0
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        self._ws = None

    # ---- PlanBuilder surface -------------------------------------------------------------------------------------------------
    def alloc(self, shape, dtype=None):
        return E.Act(torch.empty(shape, dtype=dtype or torch.float32, device=self.device))

    def release(self, act):            # activations are needed again by the backward: nothing is recycled
        pass

    def const(self, t, dtype=None):
        return t.to(device=self.device, dtype=dtype or t.dtype).contiguous()

    def can_stem_pool(self, *a, **k):
        return False

    def can_chain_pointwise(self, *a, **k):
        return False

    def global_avgpool(self, x):
        raise NotImplementedError("training with the confidence heads (volume_aggregation_method conf*) is not built")

    def _run_last(self):
        fn, _ = self.pb.ops[-1]
        fn(self.stream)

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=self.device)
        return self._ws

    def _add_grad(self, act, g):
        """Registers g as (part of) the gradient of ``act``; returns the tensor a later producer should ADD to (or None)."""
        self.grads[id(act)] = (act, g)

    def grad_of(self, act):
        e = self.grads.get(id(act))
        return None if e is None else e[1]

    # ---- layers --------------------------------------------------------------------------------------------------------------
    def conv(self, x, weight, bias=None, bn=None, stride=1, pad=0, transposed=False, relu=False, relu_pre=False, residual=None, out_f32=False,
             out=None, sigmoid=False):
        if sigmoid:
            raise NotImplementedError("sigmoid heads are not part of the training path")
        lib = H.lib()
        flags = (H.EPI_RELU_POST if relu else 0) | (H.EPI_RELU_PRE if relu_pre else 0)
        if bn is None:          # convolution (+ bias) -> activation in the conv epilogue, as in inference
            z = self.pb.conv(x, weight, bias, None, stride=stride, pad=pad, transposed=transposed, relu=relu, relu_pre=relu_pre, residual=residual)
            self._run_last()
            y_raw = stats = None
        else:
            y_raw = self.pb.conv(x, weight, bias, None, stride=stride, pad=pad, transposed=transposed)
            self._run_last()
            gamma, beta, rmean, rvar = bn
            Cc = y_raw.shape[-1]
            rows = y_raw.t.numel() // Cc
            mean = torch.empty(Cc, dtype=torch.float32, device=self.device)
            var = torch.empty(Cc, dtype=torch.float32, device=self.device)
            ws = self._workspace(lib.lt_bn_stats_workspace(rows, Cc))
            H.check(lib.lt_bn_stats_fwd(H.LT_F32, y_raw.t.data_ptr(), rows, Cc, mean.data_ptr(), var.data_ptr(), rmean.data_ptr(), rvar.data_ptr(),
                                        float(self.momentum), ws.data_ptr(), self.stream), "lt_bn_stats_fwd")
            z = self.alloc(y_raw.shape)
            H.check(lib.lt_bn_act_fwd(y_raw.t.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), H.ptr(residual.t) if residual is not None else None,
                                      z.t.data_ptr(), rows, Cc, BN_EPS, flags, self.stream), "lt_bn_act_fwd")
            stats = (mean, var)
        self.tape.append(lambda: self._conv_bwd(x, weight, bias, bn, stride, pad, transposed, flags, residual, y_raw, stats, z))
        return z

    def maxpool(self, x, k, s, p, nd):
        y = self.pb.maxpool(x, k, s, p, nd)
        self._run_last()
        kk = (1, k, k) if nd == 2 else (k, k, k)
        ss = (1, s, s) if nd == 2 else (s, s, s)
        pp = (0, p, p) if nd == 2 else (p, p, p)

        def bwd():
            dy = self.grad_of(y)
            if dy is None:
                return
            dx = self.grad_of(x)
            if dx is None:
                dx = torch.zeros_like(x.t)
                self._add_grad(x, dx)
            N, D, Hh, W, Cc = x.shape
            H.check(H.lib().lt_maxpool_bwd(x.t.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, D, Hh, W, Cc, H.i3(kk), H.i3(ss), H.i3(pp), self.stream),
                    "lt_maxpool_bwd")
        self.tape.append(bwd)
        return y

    # ---- backward of one convolution layer ---------------------------------------------------------------------------------------
    def _conv_bwd(self, x, weight, bias, bn, stride, pad, transposed, flags, residual, y_raw, stats, z):
        lib = H.lib()
        dz = self.grad_of(z)
        if dz is None:
            return
        Cout = z.shape[-1]
        rows = z.t.numel() // Cout
        dy = torch.empty_like(z.t)
        dres, acc_res = None, 0
        if residual is not None:
            dres = self.grad_of(residual)
            acc_res = 1 if dres is not None else 0
            if dres is None:
                dres = torch.empty_like(residual.t)
                self._add_grad(residual, dres)
        if bn is not None:
            gamma, beta, _, _ = bn
            mean, var = stats
            dgamma, dbeta = torch.empty_like(mean), torch.empty_like(mean)
            ws = self._workspace(lib.lt_bn_act_bwd_workspace(rows, Cout))
            H.check(lib.lt_bn_act_bwd(dz.data_ptr(), y_raw.t.data_ptr(), H.ptr(residual.t) if residual is not None else None, mean.data_ptr(), var.data_ptr(),
                                      gamma.data_ptr(), beta.data_ptr(), dy.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), H.ptr(dres), acc_res, rows, Cout,
                                      BN_EPS, flags, ws.data_ptr(), self.stream), "lt_bn_act_bwd")
            self._param_grad(gamma, dgamma)
            self._param_grad(beta, dbeta)
        else:
            H.check(lib.lt_act_bwd(dz.data_ptr(), z.t.data_ptr(), H.ptr(residual.t) if residual is not None else None, dy.data_ptr(), H.ptr(dres), acc_res,
                                   z.t.numel(), flags, self.stream), "lt_act_bwd")
        if bias is not None and bias.requires_grad:
            db = torch.empty(Cout, dtype=torch.float32, device=self.device)
            ws = self._workspace(lib.lt_channel_sum_workspace(rows, Cout))
            H.check(lib.lt_channel_sum(dy.data_ptr(), rows, Cout, db.data_ptr(), 0, ws.data_ptr(), self.stream), "lt_channel_sum")
            self._param_grad(bias, db)
        nd = weight.dim() - 2
        st3 = ((1,) + (stride,) * 2) if nd == 2 else (stride,) * 3
        pd3 = ((0,) + (pad,) * 2) if nd == 2 else (pad,) * 3
        ks = tuple(weight.shape[2:])
        ks3 = ((1,) + ks) if nd == 2 else ks
        taps_all = torch.tensor([(a, b, c, 0) for a in range(ks3[0]) for b in range(ks3[1]) for c in range(ks3[2])], dtype=torch.int32, device=self.device)
        ntaps = taps_all.shape[0]
        N, D, Hh, W, cin_buf = x.shape
        if weight.requires_grad:
            # ---- weight gradient: dw[co][tap * Cin + ci] over the GEMM rows of the forward convolution
            if not transposed:
                cop = E.cout_pad_of(Cout)
                kp = ntaps * cin_buf
                dw = torch.empty(cop, kp, dtype=torch.float32, device=self.device)
                H.check(lib.lt_conv_wgrad(dy.data_ptr(), x.t.data_ptr(), taps_all.data_ptr(), dw.data_ptr(), N, D, Hh, W, cin_buf, z.shape[1], z.shape[2], z.shape[3],
                                          H.i3(st3), H.i3(pd3), Cout, Cout, cop, kp, ntaps, 0, self.stream), "lt_conv_wgrad")
                g = dw[:Cout].reshape(Cout, *ks3, cin_buf)[..., :weight.shape[1]].permute(0, 4, 1, 2, 3)
                self._param_grad(weight, (g[:, :, 0] if nd == 2 else g).contiguous())
            else:
                # ConvTranspose: out[2 q - p + a] += x[q][ci] W[ci][co][a]  ->  dW[ci][a][co] = sum_q x[q][ci] dY[2 q - p + a][co]: the same
                # kernel with the layer's INPUT in the role of "dy" (rows = input pixels) and the output gradient in the role of "x"
                cin_t = weight.shape[0]
                cop = E.cout_pad_of(cin_t)
                kp = ntaps * Cout
                dw = torch.empty(cop, kp, dtype=torch.float32, device=self.device)
                oN, oD, oH, oW, _ = z.shape
                H.check(lib.lt_conv_wgrad(x.t.data_ptr(), dy.data_ptr(), taps_all.data_ptr(), dw.data_ptr(), oN, oD, oH, oW, Cout, D, Hh, W,
                                          H.i3(st3), H.i3(pd3), cin_t, cin_buf, cop, kp, ntaps, 0, self.stream), "lt_conv_wgrad(transposed)")
                g = dw[:cin_t].reshape(cin_t, *ks3, Cout).permute(0, 4, 1, 2, 3)
                self._param_grad(weight, (g[:, :, 0] if nd == 2 else g).contiguous())
        # ---- input gradient (skipped for the network input)
        if id(x) in self.no_grad_ids:
            return
        prev = self.grad_of(x)
        w = weight.detach()
        if Cout & (Cout - 1) or Cout < 4:       # lt_conv_fwd wants a power-of-two channel count on its input: pad dY with zero channels (17 joints -> 32)
            cpad = max(4, 1 << (Cout - 1).bit_length())
            dyp = torch.zeros(*dy.shape[:-1], cpad, dtype=torch.float32, device=self.device)
            dyp[..., :Cout] = dy
            dy = dyp
        dya = E.Act(dy)
        if not transposed and stride == 1:
            wt = w.transpose(0, 1).flip(*range(2, 2 + nd)).contiguous()           # [Cin, Cout, k..]: correlation with the flipped, transposed filter
            dx = self.pb.conv(dya, wt, None, None, stride=1, pad=ks[0] - 1 - pad, residual=E.Act(prev) if prev is not None else None)
        elif not transposed:
            assert stride == 2 and all(d % 2 == 0 for d in ((Hh, W) if nd == 2 else (D, Hh, W))), "stride-2 layers need even input sizes"
            # the adjoint of a strided convolution is the transposed convolution with the SAME weight tensor read as [in = Cout, out = Cin]
            dx = self.pb.conv(dya, w, None, None, stride=2, pad=pad, transposed=True, output_padding=1, residual=E.Act(prev) if prev is not None else None)
        else:
            # the adjoint of ConvTranspose(weight [Cin, Cout, k..], stride 2) is Conv(stride 2) with that tensor read as [out = Cin, in = Cout]
            dx = self.pb.conv(dya, w, None, None, stride=2, pad=pad, residual=E.Act(prev) if prev is not None else None)
        self._run_last()
        g = dx.t
        if g.shape[-1] != cin_buf:
            raise RuntimeError("input gradient has %d channels, the activation %d" % (g.shape[-1], cin_buf))
        self._add_grad(x, g)

    def _param_grad(self, p, g):
        old = self.param_grads.get(p)
        self.param_grads[p] = g if old is None else old + g          # a parameter used twice (not the case in these nets)

    # ---- driver ---------------------------------------------------------------------------------------------------------------------
    def seed(self, act, grad):
        """Gradient of the loss with respect to an output Act of the recorded forward."""
        self._add_grad(act, grad.contiguous())

    def add_backward(self, fn):
        """A custom op's backward (unprojection): fn() reads grad_of(output) and registers the inputs' gradients."""
        self.tape.append(fn)

    def backward(self):
        for fn in reversed(self.tape):
            fn()
        self.tape = []
        return self.param_grads


def adam_groups(model, config_opt):
    """The reference's three parameter groups for the volumetric model (train.py:430-437): backbone at ``lr``, ``process_features`` and
    ``volume_net`` at their own rates when the config names them."""
    lr = config_opt.lr
    return [{"params": list(model.backbone.parameters())},
            {"params": list(model.process_features.parameters()), "lr": getattr(config_opt, "process_features_lr", lr) if hasattr(config_opt, "process_features_lr") else lr},
            {"params": list(model.volume_net.parameters()), "lr": getattr(config_opt, "volume_net_lr", lr) if hasattr(config_opt, "volume_net_lr") else lr}]


class Adam:
    """torch.optim.Adam's update through lt_adam_step (one launch per tensor), same hyper-parameter surface for what the reference uses
    (train.py:430-437: lr per group, default betas / eps, no weight decay)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        groups = params if params and isinstance(params[0], dict) else [{"params": list(params)}]
        self.param_groups = [dict({"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}, **g) for g in groups]
        self.state = {}

    def zero_grad(self):
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None

    def step(self):
        lib = H.lib()
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None or not p.requires_grad:
                    continue
                st = self.state.setdefault(p, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
                st["step"] += 1
                grad = p.grad.contiguous()
                with torch.no_grad():
                    H.check(lib.lt_adam_step(p.data_ptr(), grad.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), p.numel(), float(g["lr"]),
                                             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), st["step"],
                                             torch.cuda.current_stream(p.device).cuda_stream), "lt_adam_step")
                    p.add_(0)      # bumps the version counter: cached inference plans are rebuilt from the new weights
