"""Triangulation networks with the reference's surface (mvn/models/triangulation.py of the reference):

    VolumetricTriangulationNet(config, device).forward(images, proj_matricies, batch)
    AlgebraicTriangulationNet(config, device).forward(images, proj_matricies, batch)

Same constructor side effects on ``config``, same ``state_dict()`` keys, same return tuples.  The whole
forward is ONE recorded plan of liblt_hip launches per input shape, captured into a hipGraph:

    images --lt_nchw_to_nhwc--> [graph: PoseResNet convs -> process_features 1x1 -> lt_coord_volumes ->
    lt_unproject_fwd -> V2V convs -> lt_softargmax3d_fwd] --> outputs

Host work per call is numpy fp64 camera algebra for B*NV 3x4 matrices (vectorised; the reference
deep-copies B*NV Camera objects) and one small H2D copy of the geometry block.
"""
import ctypes as C

from collections import OrderedDict

import os

import numpy as np
import torch
import torch.nn as nn

import lt_engine as E
import lt_hip as H
from mvn.models import pose_resnet
from mvn.models.v2v import V2VModel
from mvn.utils import multiview, op, volumetric


GEO_RING = 4          # pinned geometry staging slots per plan (forwards the host may run ahead of the GPU)
MAX_LAUNCH_ELEMS = 2 ** 31   # ONE SAMPLE's largest activation: liblt_hip's convolution kernels index a launch with 32-bit element offsets; batches beyond that
                             # are walked in sample chunks INSIDE lt_conv_fwd (64-bit base pointers, round 6), the other kernels of the path index in 64 bits


def _bn_in_train_mode(m):
    return any(isinstance(c, nn.modules.batchnorm._BatchNorm) and c.training for c in m.modules())


def _sync_buffers_before_eval(model):
    """DistributedDataParallel(broadcast_buffers=True) hands rank 0's buffers to every rank at EVERY forward, evaluation included (the reference
    validates through the wrapped model, train.py:450-460).  The training forward does it through its reducer; here the first evaluation forward
    after a training step does (the BatchNorm statistics of the ranks differ by their last momentum update until then; buffers do not change in
    eval(), so once is enough).  A COLLECTIVE: like under DDP, every rank has to run that evaluation forward -- a rank-0-only forward (master-only
    visualisation, checkpoint evaluation) on a model with an attached reducer would wait for the other ranks forever.  For that use set
    ``model.sync_buffers_on_eval = False`` (or LT_NO_EVAL_BUFFER_SYNC=1): the forward then runs on this rank's own statistics and no collective is issued
    (ADVICE r4)."""
    if not getattr(model, "sync_buffers_on_eval", True) or os.environ.get("LT_NO_EVAL_BUFFER_SYNC") == "1":
        return
    red = getattr(model, "grad_reducer", None)
    if red is not None and red.attached and model.__dict__.get("_buffers_stale"):
        red.sync_buffers()
        model.__dict__["_buffers_stale"] = False


class _VolTrainPlan:
    """The training step of VolumetricTriangulationNet for one input shape, recorded once and replayed (lt_train.TrainTape): the first
    forward / backward run the layers while recording them; later steps re-launch the recorded closures over the same buffers.  What
    changes from step to step enters through fixed buffers: the images (layout change into ``x_in``), the geometry block (one pinned
    H2D copy, as in inference), the parameters (gathered live into the GEMM layouts), the loss gradient (``lt_softargmax3d_bwd`` into
    ``gl``)."""

    def __init__(self, model, B, NV, Hh, W, device):
        self.model, self.B, self.NV, self.Hh, self.W, self.device = model, B, NV, Hh, W, device
        self.tape = None
        self.step_id = 0

    def forward(self, images, batch):
        import lt_train
        model, B, NV, Hh, W, device = self.model, self.B, self.NV, self.Hh, self.W, self.device
        V, J = model.volume_size, model.num_joints
        lib = H.lib()
        st = torch.cuda.current_stream(device).cuda_stream
        x = images.reshape(B * NV, 3, Hh, W)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        first = self.tape is None
        if first:
            prec = getattr(model, "train_precision", "fp32")
            mixed, act16 = prec != "fp32", prec in ("act16", "fp8v2v")
            self.tape = tape = lt_train.TrainTape(device, params=list(model.parameters()), reducer=getattr(model, "grad_reducer", None), mixed=mixed, act16=act16,
                                                  fp8_3d=prec == "fp8v2v")
            self.x_in = tape.alloc((B * NV, 1, Hh, W, E.min_cin_of(torch.bfloat16 if mixed else torch.float32)))
            tape.no_grad_ids.add(id(self.x_in))
        tape = self.tape
        ac = tape.acode          # element type of the activations: fp32, or bf16 in the 16-bit-activation step
        H.check(lib.lt_nchw_to_nhwc(ac, x.data_ptr(), self.x_in.t.data_ptr(), B * NV, 3, Hh * W, self.x_in.t.shape[-1], st), "lt_nchw_to_nhwc")
        if first:
            # layers in front of the unprojection (its launch needs the map size, the geometry block the launch reads needs the maps' size too)
            _, feats256, _, volc = model.backbone.record(tape, self.x_in, want_heatmaps=False)
            self.volc = volc          # conf / conf_norm: the vol_confidences head's sigmoid output, Act [1,1,1,B*NV,32] (pose_resnet.py:140-174)
            pf = model.process_features[0]
            self.feats = feats = tape.conv(feats256, pf.weight, pf.bias, None)
            h, w = feats.shape[2], feats.shape[3]
            n_geo = B * NV * 12 + B * 15
            self.G = {"hw": (h, w), "offs": (B * NV * 12, B * NV * 12 + 3 * B, B * NV * 12 + 6 * B),
                      "geo_ring": [torch.zeros(n_geo, dtype=torch.float32).pin_memory() for _ in range(GEO_RING)],
                      "geo_events": [None] * GEO_RING, "geo_slot": 0}
            self.geo = torch.zeros(n_geo, dtype=torch.float32, device=device)
            self.n_front = len(tape.fwd_ops)
        else:
            tape.stream = st
            tape._gather_all("fwd", tape.fwd_jobs)          # every layer's weights into its GEMM layout, one launch
            tape.replay(tape.fwd_ops, 0, self.n_front)
        G = self.G
        h, w = G["hw"]
        o_pos, o_cen, o_rot = G["offs"]
        # host geometry exactly as in inference (numpy fp64; theta ~ U(0, 2 pi) because self.training), one pinned H2D copy
        position, base, sides = model._host_geometry(batch, B, (Hh, W), G)
        self.geo.copy_(G["geo_host"], non_blocking=True)
        ev = G["geo_events"][G["geo_slot"]] = G["geo_events"][G["geo_slot"]] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        if first:
            feats, gp = self.feats, self.geo.data_ptr()
            self.coords = coords = torch.empty(B, V, V, V, 3, dtype=torch.float32, device=device)
            vol = tape.alloc((B, V, V, V, 32))
            step = float(np.float32(model.cuboid_side / (V - 1)))
            agg = H.AGG[model.volume_aggregation_method]
            cmu = int(bool(model.transfer_cmu_to_human36m))
            volc = self.volc
            conf_p = None if volc is None else volc.t.data_ptr()          # (B, NV, 32) raw confidences; 'conf_norm' is normalised inside the kernels
            tape.do(lambda s_: H.check(lib.lt_unproject_grid_fwd(ac, feats.t.data_ptr(), gp, gp + 4 * o_pos, gp + 4 * o_cen, gp + 4 * o_rot, step, cmu,
                                                                 coords.data_ptr(), conf_p, vol.t.data_ptr(), B, NV, 32, h, w, V, agg, s_), "lt_unproject_grid_fwd"),
                    "unproject")

            def unproject_bwd():
                dvol = tape.grad_of(vol)
                if dvol is None:
                    return
                gfe32 = torch.empty(feats.t.shape, dtype=torch.float32, device=device)                    # written completely by the gather (no zero fill)
                gconf = None if volc is None else torch.empty_like(volc.t)
                if tape.act16:
                    # lt_unproject_bwd takes and gives fp32 gradients: the volume's bf16 gradient is widened in front of it, the maps' gradient rounded behind it
                    dvol16 = dvol
                    dvol32 = torch.empty(dvol16.shape, dtype=torch.float32, device=device)
                    nvol = dvol16.numel()
                    tape.do(lambda s_: H.check(lib.lt_convert_pad(H.LT_BF16, dvol16.data_ptr(), H.LT_F32, dvol32.data_ptr(), nvol // 32, 32, 32, s_), "lt_convert_pad"), "cast")
                else:
                    dvol32 = dvol
                per_sample = lib.lt_unproject_bwd_workspace(1, NV, 32, V, V, V)
                # the gather's workspace comes out of the tape's shared main-stream scratch (it is only live during this one op; a private buffer per
                # cached plan was up to 4 GiB each), capped like op.unproject_heatmaps' autograd path: below the whole batch's worth the
                # entry point walks the batch in chunks
                nws = int(max(16, min(per_sample * B, max(per_sample, op.UNPROJECT_BWD_WORKSPACE_CAP))))
                tape._ws_need(nws)
                tape.do(lambda s_: H.check(lib.lt_unproject_bwd(ac, feats.t.data_ptr(), gp, coords.data_ptr(), conf_p, dvol32.data_ptr(), gfe32.data_ptr(),
                                                                H.ptr(gconf), B, NV, 32, h, w, V, V, V, agg, tape._ws.data_ptr(), nws, s_), "lt_unproject_bwd"),
                        "unproject_bwd")
                gfe = gfe32
                if tape.act16:
                    gfe = torch.empty(feats.t.shape, dtype=torch.bfloat16, device=device)
                    nfe = gfe.numel()
                    tape.do(lambda s_: H.check(lib.lt_convert_pad(H.LT_F32, gfe32.data_ptr(), H.LT_BF16, gfe.data_ptr(), nfe // 32, 32, 32, s_), "lt_convert_pad"), "cast")
                tape.seed(feats, gfe)
                if volc is not None:
                    tape.seed(volc, gconf)            # the head's backward (recorded earlier, so replayed later) starts from here
            tape.add_backward(unproject_bwd)
            self.logits = logits = model.volume_net.record(tape, vol)          # channels-last (B,V,V,V,J) fp32
            self.kp = kp = torch.empty(B, J, 3, dtype=torch.float32, device=device)
            self.probs = probs = torch.empty(B, J, V, V, V, dtype=torch.float32, device=device)
            ws = torch.empty(max(1, lib.lt_softargmax3d_workspace(B, J, V ** 3)), dtype=torch.uint8, device=device)
            self.mult, self.sm = float(model.volume_multiplier), int(bool(model.volume_softmax))
            mult, sm = self.mult, self.sm
            tape.do(lambda s_: H.check(lib.lt_softargmax3d_fwd(logits.t.data_ptr(), coords.data_ptr(), mult, sm, 1, J, kp.data_ptr(), probs.data_ptr(), B, J, V ** 3,
                                                               ws.data_ptr(), s_), "lt_softargmax3d_fwd"))
            self.gl = torch.empty(B, V ** 3, J, dtype=torch.float32, device=device)         # d loss / d logits, channels-last like the logits (fp32 in every
            tape.seed(logits, self.gl.view(logits.t.shape))                                  # precision: the logits layer stores fp32, TrainTape._conv_bwd rounds its dY once)
        else:
            tape.replay(tape.fwd_ops, self.n_front)
        self.step_id += 1
        conf_out = None
        if self.volc is not None:          # the returned vol_confidences (reference :266-269, :355): (B, NV, 32), normalised over the views for conf_norm
            conf_out = self.volc.t.reshape(B, NV, 32).clone()
            if model.volume_aggregation_method == "conf_norm":
                conf_out = conf_out / conf_out.sum(dim=1, keepdim=True)
        self.conf_out = conf_out
        feats_out = torch.empty(B, NV, 32, h, w, dtype=torch.float32, device=device)          # the returned features, contiguous like the reference's
        H.check(lib.lt_nhwc_to_nchw_f32(ac, self.feats.t.data_ptr(), feats_out.data_ptr(), B * NV, 32, h * w, 32, st), "lt_nhwc_to_nchw_f32")
        base_points = self.geo[o_cen:o_rot].reshape(B, 3).clone()
        return self.kp.clone(), self.probs.clone(), feats_out, self.coords.clone(), base_points, position, sides

    def backward(self, g_kp, idx, val, g_dense=None):
        B, J = self.probs.shape[:2]
        nvox = self.probs[0, 0].numel()
        st = torch.cuda.current_stream(self.device).cuda_stream
        ws = None
        if g_dense is not None:          # a dense gradient on the returned volumes (any loss on them, as the reference's autograd accepts)
            ws = self.__dict__.setdefault("_pg_ws", torch.empty(B * J, dtype=torch.float32, device=self.device))
        H.check(H.lib().lt_softargmax3d_bwd_dense(self.probs.data_ptr(), self.coords.data_ptr(), self.kp.data_ptr(), g_kp.data_ptr(), H.ptr(idx), H.ptr(val), H.ptr(g_dense),
                                                  H.ptr(ws), self.mult, self.sm, 1, self.gl.data_ptr(), B, J, nvox, st), "lt_softargmax3d_bwd")
        pg = self.tape.run_backward()
        flat = self.tape.arena.clone()          # autograd gets its own copy: the arena is overwritten by the next step
        off = self.tape.arena.data_ptr()
        return {p: flat[(v.data_ptr() - off) // 4:(v.data_ptr() - off) // 4 + v.numel()].view(v.shape) for p, v in pg.items()}


class _VolTrainFn(torch.autograd.Function):
    """The whole training-mode forward of VolumetricTriangulationNet as ONE autograd node: forward runs (the first time: records) the
    layers through the plan's lt_train.TrainTape (liblt_hip, fp32, BatchNorm on batch statistics), backward replays the tape's backward
    and hands the parameter gradients back to autograd -- so ``loss.backward()``, ``torch.optim`` / ``lt_train.Adam`` work as with the
    reference (train.py:233-243).  Inputs: (plan, images, batch, *parameters)."""

    @staticmethod
    def forward(ctx, plan, images, batch, *params):
        ctx.set_materialize_grads(False)              # an unused output arrives as None in backward, not as a dense zero tensor
        ctx._lt_accepts_sparse_prob_grads = True      # VolumetricCELoss's one-voxel gradients are applied inside lt_softargmax3d_bwd
        with torch.cuda.device(images.device):
            kp, probs, feats_out, coords, base_points, position, sides = plan.forward(images, batch)
        ctx.plan, ctx.step_id, ctx.params = plan, plan.step_id, params
        ctx.mark_non_differentiable(feats_out, coords, base_points)
        plan.model._train_extra = (position, sides)
        return kp, probs, feats_out, coords, base_points

    @staticmethod
    def backward(ctx, g_kp, g_probs, *unused):
        plan = ctx.plan
        if ctx.step_id != plan.step_id:
            raise RuntimeError("only the LATEST training forward of a shape can be backpropagated (its activations live in the plan's buffers, "
                               "which a newer forward has overwritten)")
        with torch.cuda.device(plan.device):
            g_kp = torch.zeros_like(plan.kp) if g_kp is None else g_kp.float().contiguous()
            sparse = getattr(ctx, "_lt_sparse_prob_grads", [])
            idx = val = g_dense = None
            if len(sparse) == 1:
                idx, val = sparse[0]
            elif len(sparse) > 1:
                raise NotImplementedError("more than one sparse (VolumetricCELoss) gradient on the returned volumes of one forward")
            if g_probs is not None and not op.is_sparse_placeholder(ctx, g_probs):
                # any other loss on the volumes: a dense (B, J, V, V, V) gradient, added to the soft-argmax backward's a_i (round 4; VolumetricCELoss's
                # placeholder next to its sparse gradient is recognised by its storage -- a uniform gradient, volumes.sum(), is a real one)
                g_dense = g_probs.float().contiguous()
            pg = plan.backward(g_kp, idx, val, g_dense)
        grads = tuple(pg.get(p) if p.requires_grad else None for p in ctx.params)
        return (None, None, None) + grads


class _AlgTrainPlan:
    """The training step of AlgebraicTriangulationNet's backbone for one input shape (lt_train.TrainTape, recorded once / replayed): images ->
    heatmap logits (N, J, h, w) and, with ``use_confidences``, the alg_confidences head's sigmoid output (N, J).  The tail of the model -- 2D
    soft-argmax, confidence normalisation, DLT -- runs on these two small tensors as ordinary autograd nodes (mvn.utils.op / multiview)."""

    def __init__(self, model, N, Hh, W, device):
        self.model, self.N, self.Hh, self.W, self.device = model, N, Hh, W, device
        self.tape = None
        self.step_id = 0

    def forward(self, x):
        import lt_train
        model, N, Hh, W, device = self.model, self.N, self.Hh, self.W, self.device
        lib = H.lib()
        st = torch.cuda.current_stream(device).cuda_stream
        first = self.tape is None
        if first:
            prec = getattr(model, "train_precision", "fp32")
            if prec not in ("fp32", "bf16", "act16"):
                raise ValueError("AlgebraicTriangulationNet.train_precision must be 'fp32', 'bf16' or 'act16' (there is no 3D convolution for 'fp8v2v')")
            mixed = prec != "fp32"
            self.tape = tape = lt_train.TrainTape(device, params=list(model.parameters()), reducer=getattr(model, "grad_reducer", None), mixed=mixed, act16=prec == "act16")
            self.x_in = tape.alloc((N, 1, Hh, W, E.min_cin_of(torch.bfloat16 if mixed else torch.float32)))
            tape.no_grad_ids.add(id(self.x_in))
        tape = self.tape
        H.check(lib.lt_nchw_to_nhwc(tape.acode, x.data_ptr(), self.x_in.t.data_ptr(), N, 3, Hh * W, self.x_in.t.shape[-1], st), "lt_nchw_to_nhwc")
        if first:
            hm, _, algc, _ = model.backbone.record(tape, self.x_in, want_heatmaps=True)
            self.hm, self.algc = hm, algc
            self.g_hm = torch.empty_like(hm.t)
            tape.seed(hm, self.g_hm)
            self.g_conf = None
            if algc is not None:
                self.g_conf = torch.empty_like(algc.t)
                tape.seed(algc, self.g_conf)
        else:
            tape.stream = st
            tape._gather_all("fwd", tape.fwd_jobs)
            tape.replay(tape.fwd_ops)
        self.step_id += 1
        _, _, h, w, J = self.hm.shape
        hm_out = torch.empty(N, J, h, w, dtype=torch.float32, device=device)
        H.check(lib.lt_nhwc_to_nchw_f32(H.LT_F32, self.hm.t.data_ptr(), hm_out.data_ptr(), N, J, h * w, J, st), "lt_nhwc_to_nchw_f32")
        conf_out = self.algc.t.reshape(N, J).clone() if self.algc is not None else torch.empty(0, device=device)
        return hm_out, conf_out

    def backward(self, g_hm, g_conf):
        N = self.N
        _, _, h, w, J = self.hm.shape
        lib = H.lib()
        st = torch.cuda.current_stream(self.device).cuda_stream
        if g_hm is None:
            H.check(lib.lt_zero(self.g_hm.data_ptr(), self.g_hm.numel() * 4, st), "lt_zero")
        else:
            g = g_hm.float().contiguous()
            H.check(lib.lt_nchw_to_nhwc(H.LT_F32, g.data_ptr(), self.g_hm.data_ptr(), N, J, h * w, J, st), "lt_nchw_to_nhwc")
        if self.g_conf is not None:
            if g_conf is None:
                H.check(lib.lt_zero(self.g_conf.data_ptr(), self.g_conf.numel() * 4, st), "lt_zero")
            else:
                self.g_conf.view(N, J).copy_(g_conf.float())
        pg = self.tape.run_backward()
        flat = self.tape.arena.clone()          # autograd gets its own copy: the arena is overwritten by the next step
        off = self.tape.arena.data_ptr()
        return {p: flat[(v.data_ptr() - off) // 4:(v.data_ptr() - off) // 4 + v.numel()].view(v.shape) for p, v in pg.items()}


class _AlgTrainFn(torch.autograd.Function):
    """The training-mode backbone of AlgebraicTriangulationNet as ONE autograd node (the counterpart of _VolTrainFn).  Inputs: (plan, images
    (N,3,H,W), *parameters); outputs: heatmap logits (N,J,h,w), raw confidences (N,J) (empty without the head)."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        ctx.set_materialize_grads(False)
        with torch.cuda.device(x.device):
            hm, conf = plan.forward(x)
        ctx.plan, ctx.step_id, ctx.params = plan, plan.step_id, params
        if conf.numel() == 0:
            ctx.mark_non_differentiable(conf)
        return hm, conf

    @staticmethod
    def backward(ctx, g_hm, g_conf):
        plan = ctx.plan
        if ctx.step_id != plan.step_id:
            raise RuntimeError("only the LATEST training forward of a shape can be backpropagated (its activations live in the plan's buffers, "
                               "which a newer forward has overwritten)")
        with torch.cuda.device(plan.device):
            pg = plan.backward(g_hm, g_conf)
        return (None, None) + tuple(pg.get(p) if p.requires_grad else None for p in ctx.params)


class _PlannedNet(E.PlanCache):
    """A whole triangulation net behind one plan cache (lt_engine.PlanCache: LRU + weights fingerprint)."""

    def __init__(self):
        super().__init__()
        self.compute_dtype = torch.float32   # the reference's precision; torch.bfloat16 = throughput mode
        self.use_graph = True
        self.copy_outputs = True             # False: return views of plan-owned buffers (valid until the next forward)
        self.tile_override = 0
        self._stream = None
        self._init_plan_cache()

    def set_compute_dtype(self, dtype):
        H.dtype_code(dtype)
        self.compute_dtype = dtype
        return self

    def _side_stream(self, device):
        if self._stream is None or self._stream.device != device:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream


class VolumetricTriangulationNet(_PlannedNet):
    def __init__(self, config, device="cuda:0"):
        super().__init__()
        m = config.model
        self.num_joints = m.backbone.num_joints
        self.volume_aggregation_method = m.volume_aggregation_method
        self.volume_softmax = m.volume_softmax
        self.volume_multiplier = m.volume_multiplier
        self.volume_size = m.volume_size
        self.cuboid_side = m.cuboid_side
        self.kind = m.kind
        self.use_gt_pelvis = m.use_gt_pelvis
        self.heatmap_softmax = m.heatmap_softmax
        self.heatmap_multiplier = m.heatmap_multiplier
        self.transfer_cmu_to_human36m = m.transfer_cmu_to_human36m if hasattr(m, "transfer_cmu_to_human36m") else False
        if self.volume_aggregation_method not in H.AGG:
            raise ValueError("Unknown volume_aggregation_method: {}".format(self.volume_aggregation_method))
        # same config mutation as the reference (:228-231)
        m.backbone.alg_confidences = False
        m.backbone.vol_confidences = self.volume_aggregation_method.startswith("conf")
        self.backbone = pose_resnet.get_pose_net(m.backbone, device=device)
        for p in self.backbone.final_layer.parameters():
            p.requires_grad = False
        self.process_features = nn.Sequential(nn.Conv2d(256, 32, 1))
        self.volume_net = V2VModel(32, self.num_joints)
        if hasattr(m, "compute_dtype"):
            self.compute_dtype = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                                  "bfloat16": torch.bfloat16}[str(m.compute_dtype)]

    # ---------------------------------------------------------------------------------------
    def _build_plan(self, B, NV, Hh, W, device, dry_run=False):
        dt = self.compute_dtype
        V, J = self.volume_size, self.num_joints
        b = E.PlanBuilder(device, dt, self.tile_override, dry_run=dry_run)
        lib = None if dry_run else H.lib()
        x_in = b.alloc((B * NV, 1, Hh, W, E.min_cin_of(dt)))
        x_in.pooled = False
        # the 1x1 heatmap head is dead in the volumetric path: only its SHAPE is used (reference :264)
        # bf16 plans: the fused stem reads the caller's fp32 images (pointer handed over per forward through this cell)
        image_cell = {"ptr": None, "ref": None}
        _, feats256, _, volc = self.backbone.record(b, x_in, want_heatmaps=False, image_cell=image_cell)
        if not b.npre:
            image_cell = None
        pf = self.process_features[0]
        feats = b.conv(feats256, pf.weight, pf.bias, None)
        b.release(feats256)
        feats.pooled = False
        h, w = feats.shape[2], feats.shape[3]
        # geometry block (fp32, one H2D copy per forward): proj B*NV*12 | pos B*3 | center B*3 | rot B*9
        n_geo = B * NV * 12 + B * 15
        geo = torch.zeros(n_geo, dtype=torch.float32, device=device)
        # pinned staging RING: forward N+1 fills the next slot while forward N's copy may still be queued; a slot is rewritten only
        # after the event recorded behind its last copy has completed (forward never blocks on the GPU otherwise)
        geo_ring = [torch.zeros(n_geo, dtype=torch.float32) for _ in range(1 if dry_run else GEO_RING)]
        if not dry_run:
            geo_ring = [g.pin_memory() for g in geo_ring]
        geo_host = geo_ring[0]
        o_pos, o_cen, o_rot = B * NV * 12, B * NV * 12 + 3 * B, B * NV * 12 + 6 * B
        coords = torch.empty(B, V, V, V, 3, dtype=torch.float32, device=device)
        step = float(np.float32(self.cuboid_side / (V - 1)))
        gp = geo.data_ptr()
        cmu = int(bool(self.transfer_cmu_to_human36m))
        conf = None
        if volc is not None:
            conf = volc.t.reshape(B, NV, 32)   # 'conf_norm' is normalised inside the kernel (LT_AGG_CONF_NORM)
        vol = b.alloc((B, V, V, V, 32))
        esz = torch.empty((), dtype=dt).element_size()
        agg = H.AGG[self.volume_aggregation_method]
        # ONE launch: the voxel centres are computed in registers from (position, centre, rotation, step) and written to the returned
        # coordinate tensor on the way (reference :298-339 + op.py:99-166); configurations without a fused kernel run
        # lt_coord_volumes + the generic gather inside the same call
        b.custom(lambda st: H.check(lib.lt_unproject_grid_fwd(b.code, feats.t.data_ptr(), gp, gp + 4 * o_pos, gp + 4 * o_cen, gp + 4 * o_rot, step, cmu,
                                                              coords.data_ptr(), H.ptr(conf), vol.t.data_ptr(), B, NV, 32, h, w, V, agg, st),
                                    "lt_unproject_grid_fwd"),
                 "unproject", nbytes=(B * NV * h * w * 32 + B * V ** 3 * 32) * esz + B * V ** 3 * 12,   # feats once + volume once + the returned coords
                 info={"feats": feats, "geo": geo, "coords": coords, "conf": conf, "vol": vol, "agg": self.volume_aggregation_method, "NV": NV,
                       "offs": (o_pos, o_cen, o_rot), "step": step, "cmu": cmu})
        logits = self.volume_net.record(b, vol)
        kp = torch.empty(B, J, 3, dtype=torch.float32, device=device)
        probs = torch.empty(B, J, V, V, V, dtype=torch.float32, device=device)
        ws = torch.empty(1 if dry_run else max(1, lib.lt_softargmax3d_workspace(B, J, V ** 3)), dtype=torch.uint8, device=device)
        mult, sm = float(self.volume_multiplier), int(bool(self.volume_softmax))
        cl = int(logits.t.is_contiguous())   # channels-last rows of J floats, or planar (N, J, V, V, V) storage (bf16 pwchain tail)
        assert cl or logits.t.permute(0, 4, 1, 2, 3).is_contiguous()
        # tail op: the RETURNED features (B, NV, 32, h, w) fp32, contiguous like the reference's (triangulation.py:346,355) -- one liblt_hip layout
        # launch from the plan's channels-last map straight into the tensor forward() hands out (it was a permuted view + an ATen cast copy);
        # skipped when the caller asked for views of the plan's buffers (copy_outputs = False)
        b.custom(lambda st, outs=None: None if not outs or len(outs) < 3 or outs[2] is None else H.check(lib.lt_nhwc_to_nchw_f32(
                     b.code, feats.t.data_ptr(), outs[2].data_ptr(), B * NV, 32, h * w, 32, st), "lt_nhwc_to_nchw_f32"),
                 "features_out", nbytes=B * NV * h * w * 32 * (esz + 4), info={"feats": feats}, tail=True)
        # tail op (outside the captured graph): forward() points it at the tensors it returns, so the 17 x 64^3 probabilities
        # are written once, where the caller gets them, instead of being cloned out of a plan buffer (0.45 ms of copies per step)
        b.custom(lambda st, outs=None: H.check(lib.lt_softargmax3d_fwd(
                     logits.t.data_ptr(), coords.data_ptr(), mult, sm, cl, J, (outs[0] if outs else kp).data_ptr(),
                     (outs[1] if outs else probs).data_ptr(), B, J, V ** 3, ws.data_ptr(), st), "lt_softargmax3d_fwd"),
                 "softargmax3d", nbytes=2 * B * J * V ** 3 * 4,  # read logits + write probabilities
                 info={"logits": logits, "coords": coords, "mult": mult, "softmax": sm, "kp": kp, "probs": probs}, tail=True)
        plan = b.finish()
        plan.keep += [geo, geo_ring, coords, kp, probs, ws]
        if image_cell is not None and not dry_run:
            x_in.t.untyped_storage().resize_(0)   # the fused stem reads the caller's images: x_in is only a shape (150 MB at B = 32)
        return {"plan": plan, "x_in": x_in, "image_cell": image_cell, "feats": feats, "geo": geo, "geo_host": geo_host, "geo_ring": geo_ring,
                "geo_events": [None] * len(geo_ring), "geo_slot": 0, "coords": coords, "kp": kp,
                "probs": probs, "conf": conf, "logits": logits, "vol": vol, "hw": (h, w), "offs": (o_pos, o_cen, o_rot),
                "captured": False}

    def _host_geometry(self, batch, B, image_shape, P):
        """numpy fp64 camera / cuboid algebra of the reference (:272-296, :318-328), vectorised; fills the plan's pinned
        geometry block (fp32: projections, cuboid origins, centres, rotations).  Returns (position, base, sides)."""
        h, w = P["hw"]
        K, R, t = multiview.stack_cameras(batch["cameras"])
        proj = multiview.resized_projections(K, R, t, image_shape, (h, w))
        kp3d = batch["keypoints_3d"] if self.use_gt_pelvis else batch["pred_keypoints_3d"]
        base = np.empty((B, 3), dtype=np.float64)
        for i in range(B):
            k3 = np.asarray(kp3d[i])
            base[i] = (k3[11, :3] + k3[12, :3]) / 2 if self.kind == "coco" else k3[6, :3]
        sides = np.array([self.cuboid_side] * 3)
        position = base - sides / 2
        axis = [0, 1, 0] if self.kind == "coco" else [0, 0, 1]
        rot = np.empty((B, 9))
        for i in range(B):
            theta = np.random.uniform(0.0, 2 * np.pi) if self.training else 0.0
            rot[i] = volumetric.get_rotation_matrix(axis, theta).reshape(-1)
        o_pos, o_cen, o_rot = P["offs"]
        slot = P["geo_slot"] = (P["geo_slot"] + 1) % len(P["geo_ring"])
        ev = P["geo_events"][slot]
        if ev is not None:
            ev.synchronize()         # the copy that last read this slot (GEO_RING forwards ago) has completed
        gh = P["geo_host"] = P["geo_ring"][slot]
        gh[:o_pos] = torch.from_numpy(proj.astype(np.float32).reshape(-1))
        gh[o_pos:o_cen] = torch.from_numpy(position.astype(np.float32).reshape(-1))
        gh[o_cen:o_rot] = torch.from_numpy(base.astype(np.float32).reshape(-1))
        gh[o_rot:] = torch.from_numpy(rot.astype(np.float32).reshape(-1))
        return position, base, sides

    # ---------------------------------------------------------------------------------------
    def max_samples_per_launch(self, NV, Hh, W):
        """Largest batch of ONE plan.  Since round 6 a plan covers any batch: tensors beyond 2^31 elements (BASELINE config 4 at 32 samples: 32 x 128^3 voxels x
        32 channels = 2^31) are walked in sample chunks inside the convolution entry points (lt_conv_chunk_samples, 64-bit base pointers; the gather, pooling,
        pointwise-chain and soft-argmax kernels index in 64 bits), so the only remaining bound is a single SAMPLE whose widest activation exceeds 32-bit
        offsets -- then 1, and liblt_hip refuses the launch loudly."""
        V = self.volume_size
        per_sample = max(32 * V ** 3, NV * 64 * ((Hh + 1) // 2) * ((W + 1) // 2), NV * 256 * ((Hh + 3) // 4) * ((W + 3) // 4))
        return (1 << 30) if per_sample < MAX_LAUNCH_ELEMS else 1

    def forward(self, images, proj_matricies, batch):
        """images (B,NV,3,H,W) fp32 on the GPU; ``proj_matricies`` is ignored exactly as in the reference
        (overwritten at :277); ``batch`` as built by datasets/utils.py:14-37 (``cameras``, and
        ``pred_keypoints_3d`` or ``keypoints_3d``).  Returns the reference's 7-tuple (:355).

        Asynchronous like the reference's CUDA forward: nothing here waits for the GPU (results are ordered on the current
        stream).  One plan per batch shape whatever its size: BASELINE config 4 at 32 samples (2^31 elements per 32-channel volume) is ONE plan and one
        captured graph -- the convolution entry points walk such tensors in sample chunks (``max_samples_per_launch``)."""
        H.require_gpu(images, "images")
        if _bn_in_train_mode(self):          # any BatchNorm module in train(): the training step (modules left in eval() keep frozen statistics)
            return self._forward_train(images, batch)
        _sync_buffers_before_eval(self)
        B, NV = images.shape[:2]
        cap = self.max_samples_per_launch(NV, images.shape[3], images.shape[4])
        if B <= cap:
            with torch.cuda.device(images.device):
                return self._forward_chunk(images, batch, 0, B)
        n = -(-B // cap)
        size = -(-B // n)            # equal-sized chunks (one plan), a shorter last one only if B does not divide
        if not self.__dict__.get("_warned_sub_batches"):
            import warnings
            warnings.warn("VolumetricTriangulationNet.forward: %d samples exceed the %d per launch that liblt_hip's 32-bit element offsets allow at this "
                          "shape (%d views, %d^3 voxels); running %d consecutive sub-batches of %d (results identical, one plan)" %
                          (B, cap, NV, self.volume_size, n, size), stacklevel=2)
            self.__dict__["_warned_sub_batches"] = True
        parts = []
        with torch.cuda.device(images.device):
            for lo in range(0, B, size):
                parts.append(self._forward_chunk(images, batch, lo, min(B, lo + size)))
        cat = lambda i: torch.cat([p[i] for p in parts], dim=0)
        conf = None if parts[0][3] is None else cat(3)
        return cat(0), cat(1), cat(2), conf, [c for p in parts for c in p[4]], cat(5), cat(6)

    def _forward_train(self, images, batch):
        """Training mode (any BatchNorm in train()): fp32, batch statistics, running statistics updated, random cuboid rotation; the
        result carries the autograd node whose backward is liblt_hip's (lt_train.py).  Same 7-tuple as the inference forward.  Every
        ``volume_aggregation_method`` trains: for conf / conf_norm the gradient reaches the backbone through the vol_confidences head as
        well (GlobalAveragePoolingHead, pose_resnet.py:140-174)."""
        # BatchNorm modules in train() use batch statistics (and update the running ones); modules left in eval() -- a frozen backbone while V2V
        # is fine-tuned, say -- normalise with their frozen running statistics (round 3; lt_train.TrainTape.conv)
        bns = [c for c in self.modules() if isinstance(c, nn.modules.batchnorm._BatchNorm) and c.training]
        if any(c.momentum is None or abs(c.momentum - 0.1) > 1e-12 or not c.track_running_stats or not c.affine for c in bns):
            raise NotImplementedError("BatchNorm with a momentum other than 0.1, without running statistics or without affine parameters")
        params = tuple(self.parameters())
        off = [n for n, t in list(self.named_parameters()) + list(self.named_buffers()) if t.device != images.device]
        if off:
            raise RuntimeError("training updates the parameters where they live: move the model to %s first (model.to(device), as "
                               "train.py:424 does); %d tensors are elsewhere, e.g. %s" % (images.device, len(off), off[0]))
        B, NV = images.shape[:2]
        key = (B, NV, images.shape[3], images.shape[4], images.device, self.volume_size, float(self.cuboid_side), float(self.volume_multiplier),
               bool(self.volume_softmax), self.volume_aggregation_method, bool(self.transfer_cmu_to_human36m), self.num_joints,
               tuple(p.requires_grad for p in params), id(getattr(self, "grad_reducer", None)), getattr(self, "train_precision", "fp32"),
               tuple(c.training for c in self.modules() if isinstance(c, nn.modules.batchnorm._BatchNorm)))
        if getattr(self, "train_precision", "fp32") not in ("fp32", "bf16", "act16", "fp8v2v"):
            raise ValueError("train_precision must be 'fp32' (the reference's precision), 'bf16' (bf16 MFMA convolutions, fp32 storage), 'act16' (bf16 MFMA "
                             "convolutions AND bf16 activations / activation gradients) or 'fp8v2v' (act16 with V2V's 3x3x3 convolutions on the fp8 MFMA)")
        red = getattr(self, "grad_reducer", None)
        if red is not None:
            # DistributedDataParallel's semantics (reference train.py:453): rank 0's parameters and buffers at "construction" (here: the
            # first training forward with this reducer, unless the caller has attached it already), rank 0's buffers at every forward
            if not red.attached:
                red.attach(self)
            else:
                red.sync_buffers()
            self.__dict__["_buffers_stale"] = True          # this step updates the running statistics per rank: the next eval forward re-syncs
        plans = self.__dict__.setdefault("_train_plans", OrderedDict())
        plan = plans.get(key)
        if plan is None:
            while len(plans) >= 2:
                plans.popitem(last=False)          # frees its activations and gradient buffers
            plan = plans[key] = _VolTrainPlan(self, B, NV, images.shape[3], images.shape[4], images.device)
        plans.move_to_end(key)
        kp, probs, feats, coords, base_points = _VolTrainFn.apply(plan, images, batch, *params)
        # every BatchNorm's num_batches_tracked += 1 (what nn.BatchNorm does in train()): ONE liblt_hip launch over a table of the counters'
        # addresses (209 one-element torch kernels per step before); the table is rebuilt only when a buffer has moved
        nbt = [c.num_batches_tracked for c in bns]
        tab_key = tuple(t.data_ptr() for t in nbt)
        tab = self.__dict__.get("_nbt_table")
        if tab is None or tab[0] != tab_key:
            tab = self.__dict__["_nbt_table"] = (tab_key, torch.tensor(tab_key, dtype=torch.int64).to(images.device))
        H.check(H.lib().lt_add_i64_multi(tab[1].data_ptr(), len(nbt), 1, torch.cuda.current_stream(images.device).cuda_stream), "lt_add_i64_multi")
        position, sides = self.__dict__.pop("_train_extra")
        cuboids = [volumetric.Cuboid3D(position[i], sides) for i in range(images.shape[0])]
        return kp, feats, probs, plan.conf_out, cuboids, coords, base_points

    def _forward_chunk(self, images, batch, lo, hi):
        device = images.device
        NV = images.shape[1]
        Hh, W = images.shape[3:]
        if lo != 0 or hi != images.shape[0]:
            images = images[lo:hi]
            batch = dict(batch, cameras=[cams[lo:hi] for cams in batch["cameras"]])
            for k in ("keypoints_3d", "pred_keypoints_3d"):
                if k in batch:
                    batch[k] = batch[k][lo:hi]
        B = hi - lo
        # everything a plan bakes in besides the weights (those: the fingerprint inside _plan_for)
        key = (B, NV, Hh, W, self.compute_dtype, device, self.use_graph, self.volume_size, float(self.cuboid_side),
               float(self.volume_multiplier), bool(self.volume_softmax), self.volume_aggregation_method,
               bool(self.transfer_cmu_to_human36m), self.tile_override, self.num_joints)
        P = self._plan_for(key, lambda: self._build_plan(B, NV, Hh, W, device))
        h, w = P["hw"]
        position, base, sides = self._host_geometry(batch, B, (Hh, W), P)
        # ---- device side ----
        cur = torch.cuda.current_stream(device)
        side = self._side_stream(device)
        side.wait_stream(cur)
        x = images.reshape(B * NV, 3, Hh, W)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        with torch.cuda.stream(side):
            st = side.cuda_stream
            P["geo"].copy_(P["geo_host"], non_blocking=True)
            ev = P["geo_events"][P["geo_slot"]] = P["geo_events"][P["geo_slot"]] or torch.cuda.Event()
            ev.record(side)
            if P["image_cell"] is not None:     # the plan's first op reads the images where they are
                P["image_cell"]["ptr"], P["image_cell"]["ref"] = x.data_ptr(), x
            else:
                H.check(H.lib().lt_nchw_to_nhwc(H.dtype_code(self.compute_dtype), x.data_ptr(), P["x_in"].t.data_ptr(), B * NV, 3, Hh * W,
                                                P["x_in"].t.shape[-1], st), "lt_nchw_to_nhwc")
            plan = P["plan"]
            if self.use_graph and not P["captured"]:
                plan.run_eager(st)          # warm-up launch outside capture (sets func attributes, loads code objects)
                side.synchronize()
                plan.capture(st)
                P["captured"] = True
            kp, probs, coords = P["kp"], P["probs"], P["coords"]
            if self.copy_outputs:       # fresh result tensors like the reference's: the tail ops write them directly
                kp, probs = torch.empty_like(kp), torch.empty_like(probs)
                feats = torch.empty(B, NV, 32, h, w, dtype=torch.float32, device=device)
                plan.run(st, (kp, probs, feats))
            else:
                plan.run(st)
                feats = P["feats"].t.reshape(B, NV, h, w, 32).permute(0, 1, 4, 2, 3)
            conf = P["conf"]
            o_pos, o_cen, o_rot = P["offs"]
            base_points = P["geo"][o_cen:o_rot].reshape(B, 3).clone()    # fp32(base), from the block that was just copied in
            if self.copy_outputs:
                coords = coords.clone()
                conf = None if conf is None else conf.clone()
            if conf is not None and self.volume_aggregation_method == "conf_norm":
                conf = conf / conf.sum(dim=1, keepdim=True)   # the RETURNED confidences are the normalised ones (reference :268-269, :355)
            for t in (kp, probs, coords, feats, base_points, conf):
                if t is not None:
                    t.record_stream(cur)     # allocated on the side stream, handed to the caller's stream
        x.record_stream(side)
        cur.wait_stream(side)
        cuboids = [volumetric.Cuboid3D(position[i], sides) for i in range(B)]
        return kp, feats, probs, conf, cuboids, coords, base_points


class AlgebraicTriangulationNet(_PlannedNet):
    def __init__(self, config, device="cuda:0"):
        super().__init__()
        m = config.model
        self.use_confidences = m.use_confidences
        m.backbone.alg_confidences = bool(self.use_confidences)   # reference :137-141
        m.backbone.vol_confidences = False
        self.backbone = pose_resnet.get_pose_net(m.backbone, device=device)
        self.heatmap_softmax = m.heatmap_softmax
        self.heatmap_multiplier = m.heatmap_multiplier

    def _build_plan(self, B, NV, Hh, W, device):
        dt = self.compute_dtype
        b = E.PlanBuilder(device, dt, self.tile_override)
        lib = H.lib()
        x_in = b.alloc((B * NV, 1, Hh, W, E.min_cin_of(dt)))
        x_in.pooled = False
        hm, feats, algc, _ = self.backbone.record(b, x_in, want_heatmaps=True)
        b.release(feats)
        N = B * NV
        J, h, w = hm.shape[4], hm.shape[2], hm.shape[3]
        hm_nchw = torch.empty(N, J, h, w, dtype=torch.float32, device=device)
        b.custom(lambda st: H.check(lib.lt_nhwc_to_nchw_f32(H.LT_F32, hm.t.data_ptr(), hm_nchw.data_ptr(), N, J, h * w, J, st), "lt_nhwc_to_nchw_f32"))
        kp2d = torch.empty(N, J, 2, dtype=torch.float32, device=device)
        probs = torch.empty_like(hm_nchw)
        mult, sm = float(self.heatmap_multiplier), int(bool(self.heatmap_softmax))
        b.custom(lambda st: H.check(lib.lt_softargmax2d_fwd(hm_nchw.data_ptr(), mult, sm, kp2d.data_ptr(), probs.data_ptr(), N * J, h, w, st),
                                    "lt_softargmax2d_fwd"))
        plan = b.finish()
        plan.keep += [hm_nchw, kp2d, probs]
        return {"plan": plan, "x_in": x_in, "kp2d": kp2d, "probs": probs, "algc": algc, "hw": (h, w), "J": J}

    def forward(self, images, proj_matricies, batch):
        """Returns (keypoints_3d (B,J,3), keypoints_2d (B,NV,J,2) in image pixels, heatmaps (B,NV,J,h,w) after
        softmax, alg_confidences (B,NV,J)) -- reference :149-200."""
        H.require_gpu(images, "images")
        if _bn_in_train_mode(self):
            return self._forward_train(images, proj_matricies)
        _sync_buffers_before_eval(self)
        device = images.device
        B, NV = images.shape[:2]
        Hh, W = images.shape[3:]
        key = (B, NV, Hh, W, self.compute_dtype, device, float(self.heatmap_multiplier), bool(self.heatmap_softmax), self.tile_override)
        with torch.cuda.device(device):
            P = self._plan_for(key, lambda: self._build_plan(B, NV, Hh, W, device))
            return self._run(P, images, proj_matricies, B, NV, Hh, W, device)

    def _forward_train(self, images, proj_matricies):
        """Training mode (round 3; the reference's loop for model_type "alg", train.py:189-236): the backbone -- heatmap head and, with
        ``use_confidences``, the alg_confidences head included -- runs its recorded training step (lt_train.TrainTape: batch-statistics
        BatchNorm, liblt_hip backward); 2D soft-argmax, confidence normalisation and the DLT are autograd nodes over liblt_hip kernels
        (lt_softargmax2d_bwd, lt_triangulate_dlt_bwd: what autograd derives through torch.svd in the reference).  Same 4-tuple as inference."""
        # BatchNorm modules in train() use batch statistics (and update the running ones); modules left in eval() -- a frozen backbone while V2V
        # is fine-tuned, say -- normalise with their frozen running statistics (round 3; lt_train.TrainTape.conv)
        bns = [c for c in self.modules() if isinstance(c, nn.modules.batchnorm._BatchNorm) and c.training]
        if any(c.momentum is None or abs(c.momentum - 0.1) > 1e-12 or not c.track_running_stats or not c.affine for c in bns):
            raise NotImplementedError("BatchNorm with a momentum other than 0.1, without running statistics or without affine parameters")
        params = tuple(self.parameters())
        off = [n for n, t in list(self.named_parameters()) + list(self.named_buffers()) if t.device != images.device]
        if off:
            raise RuntimeError("training updates the parameters where they live: move the model to %s first (model.to(device), as "
                               "train.py:424 does); %d tensors are elsewhere, e.g. %s" % (images.device, len(off), off[0]))
        device = images.device
        B, NV = images.shape[:2]
        Hh, W = images.shape[3:]
        red = getattr(self, "grad_reducer", None)
        if red is not None:          # DistributedDataParallel's semantics, as in the volumetric model
            if not red.attached:
                red.attach(self)
            else:
                red.sync_buffers()
            self.__dict__["_buffers_stale"] = True          # this step updates the running statistics per rank: the next eval forward re-syncs
        key = (B * NV, Hh, W, device, tuple(p.requires_grad for p in params), id(red), getattr(self, "train_precision", "fp32"),
               tuple(c.training for c in self.modules() if isinstance(c, nn.modules.batchnorm._BatchNorm)))
        plans = self.__dict__.setdefault("_train_plans", OrderedDict())
        plan = plans.get(key)
        if plan is None:
            while len(plans) >= 2:
                plans.popitem(last=False)
            plan = plans[key] = _AlgTrainPlan(self, B * NV, Hh, W, device)
        plans.move_to_end(key)
        x = images.reshape(B * NV, 3, Hh, W)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        hm, conf_raw = _AlgTrainFn.apply(plan, x, *params)
        nbt = [c.num_batches_tracked for c in bns]
        tab_key = tuple(t.data_ptr() for t in nbt)
        tab = self.__dict__.get("_nbt_table")
        if tab is None or tab[0] != tab_key:
            tab = self.__dict__["_nbt_table"] = (tab_key, torch.tensor(tab_key, dtype=torch.int64).to(device))
        H.check(H.lib().lt_add_i64_multi(tab[1].data_ptr(), len(nbt), 1, torch.cuda.current_stream(device).cuda_stream), "lt_add_i64_multi")
        N, J, h, w = hm.shape
        kp2d, heatmaps = op.integrate_tensor_2d(hm * self.heatmap_multiplier, self.heatmap_softmax)      # reference :158
        conf = conf_raw.reshape(B, NV, J) if conf_raw.numel() else torch.ones(B, NV, J, dtype=torch.float32, device=device)
        conf = conf / conf.sum(dim=1, keepdim=True) + 1e-5                                                 # reference :173-174
        scale = torch.tensor([W / w, Hh / h], dtype=torch.float32, device=device)
        kp2d = kp2d.reshape(B, NV, J, 2) * scale                                                          # reference :181-184
        kp3d = multiview.triangulate_batch_of_points(proj_matricies.to(device), kp2d, confidences_batch=conf)
        return kp3d, kp2d, heatmaps.reshape(B, NV, J, h, w), conf

    def _run(self, P, images, proj_matricies, B, NV, Hh, W, device):
        h, w = P["hw"]; J = P["J"]
        st = torch.cuda.current_stream(device).cuda_stream
        x = images.reshape(B * NV, 3, Hh, W).float().contiguous()
        H.check(H.lib().lt_nchw_to_nhwc(H.dtype_code(self.compute_dtype), x.data_ptr(), P["x_in"].t.data_ptr(), B * NV, 3, Hh * W,
                                        P["x_in"].t.shape[-1], st), "lt_nchw_to_nhwc")
        P["plan"].run_eager(st)
        heatmaps = P["probs"].reshape(B, NV, J, h, w).clone()
        kp2d = P["kp2d"].reshape(B, NV, J, 2)
        if P["algc"] is not None:
            conf = P["algc"].t.reshape(B, NV, J).float()
        else:
            conf = torch.ones(B, NV, J, dtype=torch.float32, device=device)
        # tiny (B*NV*J) host-style glue, reference :173-184
        conf = conf / conf.sum(dim=1, keepdim=True) + 1e-5
        scale = torch.tensor([W / w, Hh / h], dtype=torch.float32, device=device)
        kp2d = kp2d * scale
        kp3d = multiview.triangulate_batch_of_points(proj_matricies.to(device), kp2d, confidences_batch=conf)
        return kp3d, kp2d, heatmaps, conf
