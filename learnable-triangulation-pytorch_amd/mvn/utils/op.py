"""Functional ops of the volumetric path with the reference's names and signatures
(mvn/utils/op.py of the reference), executed by liblt_hip.so.

Public tensors keep the reference's layouts (heatmaps B,NV,C,h,w; volumes B,C,V,V,V); internally the
kernels are channels-last, so a channels-last-strided input (what our own networks produce) is consumed
without a copy, and outputs are returned as permuted views of channels-last storage."""
import numpy as np
import torch

import lt_hip as H

# workspace of lt_unproject_bwd's deterministic gather: at most this much (or one sample's share, if that is more); with less than the whole
# batch's worth the entry point walks the batch in chunks.  One constant for the autograd path here and the recorded training step.
UNPROJECT_BWD_WORKSPACE_CAP = 2 << 30

_METHODS = ("sum", "max", "softmax", "conf", "conf_norm")


def _as_channels_last(x, n_lead):
    """x: (*lead, C, *spatial) -> contiguous (*lead, *spatial, C) tensor (zero-copy when x already is a
    permuted view of such storage)."""
    nd = x.dim()
    perm = list(range(n_lead)) + list(range(n_lead + 1, nd)) + [n_lead]
    return x.permute(*perm).contiguous()


def build_coord_volumes(base_points, cuboid_side, volume_size, thetas=None, axis=(0, 0, 1), cmu_transfer=False, device="cuda:0"):
    """Per-sample voxel-centre grids (B,V,V,V,3) fp32 -- the loop body of
    VolumetricTriangulationNet.forward (reference mvn/models/triangulation.py:298-339) as one launch.

    base_points: (B,3) array-like (mm), fp64 on the host like the reference; thetas: per-sample rotation
    about ``axis`` through the base point (None = 0).
    """
    from mvn.utils import volumetric
    base = np.asarray(base_points, dtype=np.float64).reshape(-1, 3)
    B = base.shape[0]
    pos = (base - cuboid_side / 2.0).astype(np.float32)
    rot = np.stack([volumetric.get_rotation_matrix(axis, 0.0 if thetas is None else float(thetas[b])) for b in range(B)]).astype(np.float32)
    host = torch.from_numpy(np.concatenate([pos.reshape(-1), base.astype(np.float32).reshape(-1), rot.reshape(-1)]))
    dev = host.to(device)
    out = torch.empty(B, volume_size, volume_size, volume_size, 3, dtype=torch.float32, device=device)
    step = float(np.float32(cuboid_side / (volume_size - 1)))
    H.check(H.lib().lt_coord_volumes(dev.data_ptr(), dev.data_ptr() + 12 * B, dev.data_ptr() + 24 * B, step, B, volume_size,
                                     int(bool(cmu_transfer)), out.data_ptr(), H.cur_stream()), "lt_coord_volumes")
    return out


def _unproject_launch(feats_cl, P, cv, conf, method):
    """feats_cl (B,NV,h,w,C) contiguous channels-last -> (B,v0,v1,v2,C) channels-last, same dtype."""
    B, NV, h, w, Cc = feats_cl.shape
    v0, v1, v2 = cv.shape[1:4]
    out = torch.empty(B, v0, v1, v2, Cc, dtype=feats_cl.dtype, device=feats_cl.device)
    H.check(H.lib().lt_unproject_fwd(H.dtype_code(feats_cl.dtype), feats_cl.data_ptr(), P.data_ptr(), cv.data_ptr(), H.ptr(conf), out.data_ptr(),
                                     B, NV, Cc, h, w, v0, v1, v2, H.AGG["conf"] if conf is not None else H.AGG[method], H.cur_stream()), "lt_unproject_fwd")
    return out


class _UnprojectFn(torch.autograd.Function):
    """lt_unproject_fwd / lt_unproject_bwd as one autograd node: what autograd builds in the reference from grid_sample, the
    in-place depth mask and the view aggregation (op.py:113-162).  Gradients reach the heatmaps and (conf modes) the confidences;
    projections and coordinates get none, exactly as in the reference (they do not require grad there)."""

    @staticmethod
    def forward(ctx, heatmaps, P, cv, conf, method):
        feats = _as_channels_last(heatmaps, 2)
        out = _unproject_launch(feats, P, cv, conf, method)
        ctx.save_for_backward(feats, P, cv, conf if conf is not None else torch.empty(0, device=feats.device))
        ctx.method, ctx.has_conf = method, conf is not None
        return out.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, grad_out):
        feats, P, cv, conf = ctx.saved_tensors
        conf = conf if ctx.has_conf else None
        B, NV, h, w, Cc = feats.shape
        g = grad_out.permute(0, 2, 3, 4, 1).float().contiguous()               # (B, v0, v1, v2, C) channels-last fp32
        gfeats = torch.empty(B, NV, h, w, Cc, dtype=torch.float32, device=feats.device)      # written completely by the gather
        gconf = torch.empty_like(conf) if conf is not None else None
        v0, v1, v2 = cv.shape[1:4]
        lib = H.lib()
        # deterministic gather: workspace for as many samples as fit 2 GiB at a time (the entry walks the batch in chunks)
        per_sample = lib.lt_unproject_bwd_workspace(1, NV, Cc, v0, v1, v2)
        ws = torch.empty(max(16, min(per_sample * B, max(per_sample, UNPROJECT_BWD_WORKSPACE_CAP))), dtype=torch.uint8, device=feats.device)
        H.check(lib.lt_unproject_bwd(H.dtype_code(feats.dtype), feats.data_ptr(), P.data_ptr(), cv.data_ptr(), H.ptr(conf), g.data_ptr(),
                                     gfeats.data_ptr(), H.ptr(gconf), B, NV, Cc, h, w, v0, v1, v2,
                                     H.AGG["conf"] if conf is not None else H.AGG[ctx.method], ws.data_ptr(), ws.numel(), H.cur_stream()), "lt_unproject_bwd")
        return gfeats.permute(0, 1, 4, 2, 3).to(feats.dtype), None, None, gconf, None


def unproject_heatmaps(heatmaps, proj_matricies, coord_volumes, volume_aggregation_method="sum", vol_confidences=None):
    """heatmaps (B,NV,C,h,w), proj_matricies (B,NV,3,4), coord_volumes (B,V0,V1,V2,3),
    vol_confidences (B,NV,C) for 'conf*' -> volumes (B,C,V0,V1,V2).  Reference: op.py:99-166.  Differentiable with respect to the
    heatmaps and the confidences (lt_unproject_bwd) whenever one of them requires grad."""
    if volume_aggregation_method not in _METHODS:
        raise ValueError("Unknown volume_aggregation_method: {}".format(volume_aggregation_method))
    H.require_gpu(heatmaps, "heatmaps")
    H.dtype_code(heatmaps.dtype)
    P = proj_matricies.to(heatmaps.device, torch.float32).contiguous()
    cv = coord_volumes.to(heatmaps.device, torch.float32).contiguous()
    conf = None
    if volume_aggregation_method.startswith("conf"):
        conf = vol_confidences.to(heatmaps.device, torch.float32).contiguous()
    if torch.is_grad_enabled() and (heatmaps.requires_grad or (conf is not None and vol_confidences.requires_grad)):
        return _UnprojectFn.apply(heatmaps, P, cv, conf, volume_aggregation_method)
    return _unproject_launch(_as_channels_last(heatmaps, 2), P, cv, conf, volume_aggregation_method).permute(0, 4, 1, 2, 3)


def _softargmax3d_launch(volumes, cv, softmax):
    B, J = volumes.shape[:2]
    nvox = int(np.prod(volumes.shape[2:]))
    cl = volumes.permute(0, 2, 3, 4, 1).is_contiguous() and not volumes.is_contiguous()
    lg = (volumes.permute(0, 2, 3, 4, 1) if cl else volumes).float().contiguous()
    kp = torch.empty(B, J, 3, dtype=torch.float32, device=volumes.device)
    probs = torch.empty((B, J) + tuple(volumes.shape[2:]), dtype=torch.float32, device=volumes.device)
    lib = H.lib()
    ws = torch.empty(max(1, lib.lt_softargmax3d_workspace(B, J, nvox)), dtype=torch.uint8, device=volumes.device)
    H.check(lib.lt_softargmax3d_fwd(lg.data_ptr(), cv.data_ptr(), 1.0, int(bool(softmax)), int(cl), J, kp.data_ptr(), probs.data_ptr(),
                                    B, J, nvox, ws.data_ptr(), H.cur_stream()), "lt_softargmax3d_fwd")
    return kp, probs


class _SoftArgmax3dFn(torch.autograd.Function):
    """lt_softargmax3d_fwd / _bwd as one node with TWO outputs (coordinates, probabilities): a dense gradient on the
    probabilities goes into the same kernel (lt_softargmax3d_bwd_dense); VolumetricCELoss (mvn/models/loss.py) instead hands over its
    one-voxel-per-joint gradient in sparse form (``sparse_prob_grad``), so no (B,J,V^3) gradient tensor is ever materialised."""

    @staticmethod
    def forward(ctx, volumes, cv, softmax):
        kp, probs = _softargmax3d_launch(volumes, cv, softmax)
        ctx.save_for_backward(probs, cv, kp)
        ctx.softmax, ctx.in_dtype = bool(softmax), volumes.dtype
        return kp, probs

    @staticmethod
    def backward(ctx, g_kp, g_probs):
        probs, cv, kp = ctx.saved_tensors
        B, J = probs.shape[:2]
        nvox = probs[0, 0].numel()
        g_kp = torch.zeros_like(kp) if g_kp is None else g_kp.float().contiguous()
        # sparse gradients on the probabilities left here by consumers that ran before us (VolumetricCELoss: one voxel per (b, j))
        sparse = getattr(ctx, "_lt_sparse_prob_grads", [])
        idx = val = None
        if len(sparse) == 1:
            idx, val = sparse[0]
        dense = None
        if len(sparse) > 1:       # several sparse consumers: scatter them into one dense gradient (rare)
            dense = torch.zeros(B, J, nvox, dtype=torch.float32, device=probs.device)
            for i2, v2 in sparse:
                dense.scatter_add_(2, i2.long()[..., None], v2[..., None])
            dense = dense.reshape(probs.shape)
        if g_probs is not None and not is_sparse_placeholder(ctx, g_probs):   # a real dense gradient on the returned volumes (a uniform one -- volumes.sum() -- included)
            dense = g_probs.float() if dense is None else dense + g_probs.float()
        ws = None
        if dense is not None:     # a_i += gp_i inside the kernel (softmax: minus <p, gp>, reduced per joint into the workspace first)
            dense = dense.contiguous()
            ws = torch.empty(B * J, dtype=torch.float32, device=probs.device)
        gl = torch.empty_like(probs)
        H.check(H.lib().lt_softargmax3d_bwd_dense(probs.data_ptr(), cv.data_ptr(), kp.data_ptr(), g_kp.data_ptr(), H.ptr(idx), H.ptr(val), H.ptr(dense), H.ptr(ws), 1.0,
                                                  int(ctx.softmax), 0, gl.data_ptr(), B, J, nvox, H.cur_stream()), "lt_softargmax3d_bwd")
        return gl.to(ctx.in_dtype), None, None


def is_sparse_placeholder(node, g):
    """True when ``g`` is the all-zero stride-0 tensor ``sparse_prob_grad`` returned for a gradient it left on ``node`` in sparse form.  Recognised by the
    STORAGE of the placeholder: any other zero-stride tensor -- the expanded ones of ``volumes.sum()``, a uniform ``volumes.mean()`` -- is a real dense
    gradient (with ``volume_softmax=False`` a uniform gradient on the ReLU volumes is not zero after the backward)."""
    if g.stride() != (0,) * g.dim():
        return False
    return any(z.data_ptr() == g.data_ptr() for z in getattr(node, "_lt_placeholders", []))


def sparse_prob_grad(volumes, idx, val):
    """Hands a one-voxel-per-(sample, joint) gradient on ``volumes`` -- the probabilities returned by
    integrate_tensor_3d_with_coordinates -- to the node that produced them.  Returns the tensor the consumer's backward
    should return for ``volumes``: an all-zero stride-0 view when the producer is our soft-argmax node (which then applies the sparse
    part itself inside lt_softargmax3d_bwd), else the dense scatter."""
    node = volumes.grad_fn
    if isinstance(node, _SoftArgmax3dFn._backward_cls) or getattr(node, "_lt_accepts_sparse_prob_grads", False):
        if not hasattr(node, "_lt_sparse_prob_grads"):
            node._lt_sparse_prob_grads = []
        node._lt_sparse_prob_grads.append((idx, val))
        zero = torch.zeros((), dtype=volumes.dtype, device=volumes.device)
        if not hasattr(node, "_lt_placeholders"):
            node._lt_placeholders = []
        node._lt_placeholders.append(zero)          # kept alive on the node: the placeholder is recognised by its storage, not by its strides
        return zero.expand(volumes.shape)
    B, J = volumes.shape[:2]
    dense = torch.zeros(B, J, volumes[0, 0].numel(), dtype=torch.float32, device=volumes.device)
    dense.scatter_(2, idx.long()[..., None], val[..., None])
    return dense.reshape(volumes.shape).to(volumes.dtype)


def integrate_tensor_3d_with_coordinates(volumes, coord_volumes, softmax=True):
    """volumes (B,J,V0,V1,V2) logits, coord_volumes (B,V0,V1,V2,3) -> (coordinates (B,J,3), volumes after
    softmax / ReLU).  Reference: op.py:84-96.  Differentiable with respect to the logits (lt_softargmax3d_bwd)."""
    H.require_gpu(volumes, "volumes")
    cv = coord_volumes.to(volumes.device, torch.float32).contiguous()
    if torch.is_grad_enabled() and volumes.requires_grad:
        return _SoftArgmax3dFn.apply(volumes, cv, softmax)
    return _softargmax3d_launch(volumes, cv, softmax)


def batchnorm_batch_stats(x, running_mean=None, running_var=None, momentum=0.1):
    """Training-mode BatchNorm statistics of x (N,C,*spatial; fp32 or bf16): per-channel batch mean and BIASED variance (what
    F.batch_norm normalises with when training=True), fp64 accumulation (lt_bn_stats_fwd); running_mean / running_var (fp32, on the
    GPU) are updated in place the way torch does (momentum, unbiased variance).  Channels-last inputs are consumed without a copy."""
    H.require_gpu(x, "x")
    xc = _as_channels_last(x, 1)
    Cc = xc.shape[-1]
    rows = xc.numel() // Cc
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    var = torch.empty(Cc, dtype=torch.float32, device=x.device)
    lib = H.lib()
    ws = torch.empty(max(16, lib.lt_bn_stats_workspace(rows, Cc)), dtype=torch.uint8, device=x.device)
    if running_mean is not None:
        assert running_mean.is_cuda and running_mean.dtype == torch.float32 and running_mean.is_contiguous()
        assert running_var.is_cuda and running_var.dtype == torch.float32 and running_var.is_contiguous()
    H.check(lib.lt_bn_stats_fwd(H.dtype_code(xc.dtype), xc.data_ptr(), rows, Cc, mean.data_ptr(), var.data_ptr(), H.ptr(running_mean),
                                H.ptr(running_var), float(momentum), ws.data_ptr(), H.cur_stream()), "lt_bn_stats_fwd")
    return mean, var


def _softargmax2d_launch(heatmaps, softmax):
    N, J, h, w = heatmaps.shape
    hm = heatmaps.float().contiguous()
    coords = torch.empty(N, J, 2, dtype=torch.float32, device=hm.device)
    probs = torch.empty_like(hm)
    H.check(H.lib().lt_softargmax2d_fwd(hm.data_ptr(), 1.0, int(bool(softmax)), coords.data_ptr(), probs.data_ptr(), N * J, h, w, H.cur_stream()),
            "lt_softargmax2d_fwd")
    return coords, probs


class _SoftArgmax2dFn(torch.autograd.Function):
    """lt_softargmax2d_fwd / _bwd as one node (the algebraic model's training, train.py:189-236): the gradient of the coordinates reaches the
    heatmaps; a gradient on the returned (softmaxed) heatmaps is refused -- no loss of the reference uses them."""

    @staticmethod
    def forward(ctx, heatmaps, softmax):
        coords, probs = _softargmax2d_launch(heatmaps, softmax)
        ctx.save_for_backward(coords, probs)
        ctx.softmax, ctx.in_dtype = bool(softmax), heatmaps.dtype
        ctx.mark_non_differentiable(probs)
        return coords, probs

    @staticmethod
    def backward(ctx, g_coords, g_probs):
        coords, probs = ctx.saved_tensors
        N, J, h, w = probs.shape
        g = g_coords.float().contiguous()
        ghm = torch.empty_like(probs)
        H.check(H.lib().lt_softargmax2d_bwd(probs.data_ptr(), coords.data_ptr(), g.data_ptr(), 1.0, int(ctx.softmax), ghm.data_ptr(), N * J, h, w,
                                            H.cur_stream()), "lt_softargmax2d_bwd")
        return ghm.to(ctx.in_dtype), None


def integrate_tensor_2d(heatmaps, softmax=True):
    """heatmaps (N,J,h,w) -> (coordinates (N,J,2) as (x,y), heatmaps after softmax / ReLU).  Reference: op.py:11-47.  Differentiable with
    respect to the heatmaps (through the coordinates) when they require grad."""
    H.require_gpu(heatmaps, "heatmaps")
    if torch.is_grad_enabled() and heatmaps.requires_grad:
        return _SoftArgmax2dFn.apply(heatmaps, softmax)
    return _softargmax2d_launch(heatmaps, softmax)