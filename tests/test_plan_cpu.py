"""CPU: record the WHOLE volumetric forward as a dry-run plan (no GPU, nothing executes in liblt_hip) and interpret it
with tests/emul.py: checks the recorded network wiring, residual/skip routing, transposed-conv phases, BN folding and the
size-keyed buffer reuse against the oracle.  What remains for the GPU suite is kernel == contract."""
import numpy as np
import pytest
import torch

from emul import run_plan_on_cpu
from oracle import spec, synth
from oracle import vol_oracle as O


def _cameras(inp, B):
    from mvn.utils.multiview import Camera
    return [[Camera(inp["R"][v], inp["t"][v], inp["K"][v]) for _ in range(B)] for v in range(inp["K"].shape[0])]


@pytest.mark.parametrize("nl,method,kind,Himg", [(18, "softmax", "mpii", 64), (50, "conf_norm", "coco", 128)])
def test_recorded_plan_matches_oracle(nl, method, kind, Himg):
    from mvn.models.triangulation import VolumetricTriangulationNet
    B, NV, V = 2, 2, 32
    cfg = synth.vol_config(nl, V, method, kind=kind)
    sd = synth.make_state_dict(spec.vol_net_spec(nl, 17, method.startswith("conf")), seed=31, sharpen=True, basic_block=(nl < 50))
    inp = synth.make_inputs(B, NV, Himg, seed=31)
    m = VolumetricTriangulationNet(cfg, device="cpu")
    m.load_state_dict(sd, strict=True)
    m.eval()
    P = m._build_plan(B, NV, Himg, Himg, "cpu", dry_run=True)
    with pytest.raises(RuntimeError):
        P["plan"].run_eager(None)
    batch = {"cameras": _cameras(inp, B), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    m.training = True            # random theta (children stay in eval mode)
    np.random.seed(5)
    thetas = np.random.uniform(0.0, 2 * np.pi, size=B)
    np.random.seed(5)
    position, base, sides = m._host_geometry(batch, B, (Himg, Himg), P)
    P["geo"].copy_(P["geo_host"])
    x = inp["images"].reshape(B * NV, 3, Himg, Himg).permute(0, 2, 3, 1)
    P["x_in"].t.zero_()
    P["x_in"].t[:, 0, :, :, :3] = x
    run_plan_on_cpu(P["plan"])
    o = O.volumetric_forward(sd, cfg, inp["images"], inp["K"], inp["R"], inp["t"], inp["pred_keypoints_3d"], thetas=thetas, stages=True)
    h, w = P["hw"]
    feats = P["feats"].t.reshape(B, NV, h, w, 32).permute(0, 1, 4, 2, 3)
    rel = lambda a, r: float((a.double() - r.double()).abs().max() / r.double().abs().max())
    assert rel(P["coords"], o["coord_volumes"]) < 1e-6
    assert rel(feats, o["features"]) < 2e-5
    assert rel(P["logits"].t.permute(0, 4, 1, 2, 3), o["logits"]) < 1e-4
    assert rel(P["probs"], o["volumes"]) < 1e-3
    d = (P["kp"] - o["keypoints_3d"]).abs() / o["keypoints_3d"].abs().clamp(min=1.0)
    # a WIRING check (torch-CPU interpretation of the recorded launches, sharpened soft-argmax over 32^3 voxels), not the parity gate -- that is the GPU suite's,
    # against the reference's stored outputs.  2e-4: with the fold's constants computed like ATen's scalar path (IEEE sqrt, round 6) this case measures 1.03e-4,
    # with torch's vectorised sqrt it measured below 1e-4: last-bit changes of a few BatchNorm scales move these joints by that much (DESIGN (c): the gate sits
    # at the reference's own reproducibility floor)
    assert float(d.max()) < 2e-4, float(d.max())
    assert np.allclose(base, O.base_points_from_batch(inp["pred_keypoints_3d"], kind))
    assert P["plan"].flops > 0 and len(P["plan"].ops) > 50


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_plan_structure_pre_graph_tail(dt):
    """What is captured into the hipGraph and what stays outside: the tail ops write the tensors forward() returns (the fp32 features by a layout
    launch, joints and probabilities by the soft-argmax); bf16 plans fuse the stem (one 'stem' op instead of conv + max pool) and chain the V2V tail; a dry-run plan
    records no pre op (its interpreter reads the layout buffer, not the caller's images)."""
    from mvn.models.triangulation import VolumetricTriangulationNet
    cfg = synth.vol_config(18, 32, "softmax")     # 32^3 is the smallest cube V2V's five poolings accept
    m = VolumetricTriangulationNet(cfg, device="cpu")
    m.eval()
    m.compute_dtype = dt
    P = m._build_plan(1, 2, 64, 64, "cpu", dry_run=True)
    plan = P["plan"]
    kinds = [meta["kind"] for _, meta in plan.ops]
    assert plan.npre == 0 and P["image_cell"] is None
    assert plan.nhead == len(plan.ops) - 2 and kinds[-2:] == ["features_out", "softargmax3d"] and kinds.count("softargmax3d") == 1
    if dt == torch.bfloat16:
        assert kinds[0] == "stem" and kinds.count("pwchain") == 1 and kinds.count("maxpool") == 5   # the five 3D pools of V2V
    else:
        assert kinds[0] == "conv" and kinds[1] == "maxpool" and "stem" not in kinds and "pwchain" not in kinds
    with pytest.raises(RuntimeError):
        plan.run(None)


def test_pointwise_chain_is_recorded_for_bf16_and_matches_the_layers():
    """bf16 plans run V2V's pointwise tail (back_layers[1:] + output_layer) as ONE lt_pwchain_fwd; the recorded chain must
    be the same function as the three lt_conv_fwd launches an fp32 plan records (interpreted on the CPU)."""
    import lt_engine as E
    from mvn.models.v2v import V2VModel
    torch.manual_seed(3)
    m = V2VModel(32, 17).eval()
    for bn in [mm for mm in m.modules() if isinstance(mm, torch.nn.BatchNorm3d)]:
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.1)
    x = torch.randn(1, 4, 4, 4, 32)
    outs = {}
    for dt in (torch.bfloat16, torch.float32):
        b = E.PlanBuilder("cpu", dt, dry_run=True)
        inp = b.alloc(tuple(x.shape)); inp.pooled = False
        tail = list(m.back_layers)[1:]
        chain = [(t.block[0].weight, t.block[0].bias, (t.block[1].weight, t.block[1].bias, t.block[1].running_mean, t.block[1].running_var), True)
                 for t in tail] + [(m.output_layer.weight, m.output_layer.bias, None, False)]
        assert b.can_chain_pointwise(inp, chain) == (dt == torch.bfloat16)
        if dt == torch.bfloat16:
            y = b.pwchain(inp, chain)
        else:
            cur = inp
            for (w, bias, bn, relu) in chain[:-1]:
                cur = b.conv(cur, w, bias, bn, relu=relu)
            y = b.conv(cur, chain[-1][0], chain[-1][1], None, out_f32=True)
        plan = b.finish()
        kinds = [meta["kind"] for _, meta in plan.ops]
        assert kinds == (["pwchain"] if dt == torch.bfloat16 else ["conv"] * 3)
        inp.t.copy_(x)
        run_plan_on_cpu(plan)
        outs[dt] = y.t.float().clone()
        assert y.t.dtype == torch.float32 and tuple(y.t.shape) == (1, 4, 4, 4, 17)
    ref = outs[torch.float32]
    assert float((outs[torch.bfloat16] - ref).abs().max() / ref.abs().max()) < 3e-2   # bf16 inputs / intermediates vs fp32


def test_v2v_bf16_plan_uses_the_chain():
    import lt_engine as E
    from mvn.models.v2v import V2VModel
    m = V2VModel(32, 17).eval()
    b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
    inp = b.alloc((1, 32, 32, 32, 32)); inp.pooled = False
    out = m.record(b, inp)
    kinds = [meta["kind"] for _, meta in b.finish().ops]
    assert kinds[-1] == "pwchain" and kinds.count("pwchain") == 1
    assert tuple(out.t.shape) == (1, 32, 32, 32, 17) and out.t.dtype == torch.float32


def test_plan_cache_fingerprint_and_lru():
    """lt_engine.PlanCache: a cached plan bakes copies of the weights, so it must be rebuilt when ANY weight of the tree
    changes -- load_state_dict into a CHILD, an optimizer-style in-place update, a re-bound Parameter -- and the cache is a
    bounded LRU.  The key carries every scalar the plan bakes in (see VolumetricTriangulationNet._forward_chunk)."""
    from mvn.models.triangulation import VolumetricTriangulationNet
    m = VolumetricTriangulationNet(synth.vol_config(18, 32), device="cpu").eval()
    built = []

    def build(tag):
        def f():
            built.append(tag)
            return {"tag": tag}
        return f

    assert m._plan_for("a", build("a1"))["tag"] == "a1" and m._plan_for("a", build("a2"))["tag"] == "a1"     # cache hit
    fp0 = m.weights_fingerprint()
    # (1) child load_state_dict: only the child's own hook fires, the parent's cache must still notice
    m.backbone.load_state_dict(m.backbone.state_dict())
    assert m.weights_fingerprint() != fp0
    assert m._plan_for("a", build("a3"))["tag"] == "a3"
    # (2) in-place update under no_grad (what optimizers and EMA swaps do)
    with torch.no_grad():
        m.volume_net.output_layer.bias.add_(1.0)
    assert m._plan_for("a", build("a4"))["tag"] == "a4"
    # (3) a buffer (BatchNorm running statistics feed the folded scale/shift)
    with torch.no_grad():
        m.backbone.bn1.running_var.mul_(2.0)
    assert m._plan_for("a", build("a5"))["tag"] == "a5"
    # (4) a re-bound Parameter object and a re-bound child module
    m.process_features[0].bias = torch.nn.Parameter(torch.zeros(32))
    assert m._plan_for("a", build("a6"))["tag"] == "a6"
    m.process_features = torch.nn.Sequential(torch.nn.Conv2d(256, 32, 1))
    assert m._plan_for("a", build("a7"))["tag"] == "a7"
    assert m._plan_for("a", build("a8"))["tag"] == "a7"             # unchanged weights: still a hit
    # LRU: at most max_plans entries, least recently used goes first
    m.max_plans = 2
    m._plan_for("b", build("b1")); m._plan_for("a", build("x")); m._plan_for("c", build("c1"))
    assert list(m._plans) == ["a", "c"] and built[-2:] == ["b1", "c1"]
    m.invalidate_plans()
    assert not m._plans
    # the documented blind spot: writes through .data do not bump the version counter
    fp = m.weights_fingerprint()
    m.volume_net.output_layer.bias.data.add_(1.0)
    assert m.weights_fingerprint() == fp


def test_batches_beyond_32_bit_offsets_are_one_plan():
    """BASELINE config 4 at 32 samples is exactly 2^31 elements per 32-channel volume.  Until round 5 forward() split such batches into sub-batches of the
    model; since round 6 the convolution entry points walk them in sample chunks with 64-bit base pointers (lt_conv_chunk_samples) and a plan covers any
    batch.  Here: the model no longer splits, and the host restatement of the chunk rule (used by the builder's gates) equals the library's."""
    import lt_engine as E
    import lt_hip as H
    from mvn.models.triangulation import VolumetricTriangulationNet
    m = VolumetricTriangulationNet(synth.vol_config(18, 128), device="cpu")
    assert m.max_samples_per_launch(8, 384, 384) >= 1 << 20
    lib = H.lib()
    for N, per in ((32, 32 * 128 ** 3), (31, 32 * 128 ** 3), (64, 32 * 128 ** 3), (33, 32 * 128 ** 3), (5, 1 << 30), (7, (1 << 31) - 1), (3, 1 << 31), (1, 100),
                   (256, 64 * 192 * 192), (1000, 256 * 96 * 96), (20, 3 * 10 ** 8)):
        assert E.conv_chunk_samples(N, per) == lib.lt_conv_chunk_samples(N, per), (N, per)
    assert E.conv_chunk_samples(32, 32 * 128 ** 3) == 16 and E.conv_chunk_samples(64, 32 * 128 ** 3) == 24 and E.conv_chunk_samples(31, 32 * 128 ** 3) == 31
    assert E.conv_chunk_samples(33, 32 * 128 ** 3) == 24          # 24 + 9 (24 = the largest multiple of 8 below 31.99 samples of 2^26 elements)
    assert E.conv_chunk_samples(3, 1 << 31) == 0                  # one sample alone is too large
    b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
    w, sw = torch.zeros(32, 32, 3, 3, 3), torch.zeros(32, 16, 1, 1, 1)
    assert b.can_conv_skip((32, 128, 128, 128, 32), w, (32, 128, 128, 128, 16), sw)          # two chunks of 16 samples, each a shape the column-walk kernel takes


def test_splitk_tiny_volume_convolutions_are_recorded_and_equal_the_unsplit_convolution():
    """V2V's 3^3 128 -> 128 layers on volumes of <= 8^3 voxels are recorded as S tap-group phases of one lt_conv_fwd (fp32 partial sums,
    identity epilogue) + lt_splitk_reduce (the real epilogue): the recorded pair, interpreted on the CPU, equals torch's conv3d + folded
    BatchNorm + residual + ReLU; the tap groups partition the 27 taps; LT_CONV_NO_SPLITK=1 / fp32 plans record the single launch."""
    import os
    import torch.nn.functional as F
    import lt_engine as E
    g = torch.Generator().manual_seed(5)
    for N, sp in ((1, (2, 2, 2)), (2, (4, 4, 4)), (3, (8, 8, 8))):
        x = torch.randn(N, *sp, 128, generator=g)
        w = torch.randn(128, 128, 3, 3, 3, generator=g) * 0.02
        bias = torch.randn(128, generator=g) * 0.1
        bn = (0.5 + torch.rand(128, generator=g), torch.randn(128, generator=g) * 0.1, torch.randn(128, generator=g) * 0.1, 0.5 + torch.rand(128, generator=g))
        res = torch.randn(N, *sp, 128, generator=g)
        b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
        xa, ra = E.Act(x.clone().to(torch.bfloat16)), E.Act(res.clone().to(torch.bfloat16))
        y = b.conv(xa, w, bias, bn, stride=1, pad=1, relu=True, residual=ra)
        kinds = [(m["kind"], m["label"]) for _, m in b.ops]
        assert len(b.ops) == 2 and "split-K x" in kinds[0][1] and kinds[1][1].endswith("split-K reduce"), kinds
        pspec = b.ops[0][1]["info"]["spec"]
        S = len(pspec.phases)
        assert 2 <= S <= 8 and pspec.OD == S * sp[0] and sum(int(p.taps.shape[0]) for p in pspec.phases) == 27
        assert [tuple(p.out_off) for p in pspec.phases] == [(i * sp[0], 0, 0) for i in range(S)]
        plan = b.finish()
        xa.t, ra.t, y.t = x.clone(), res.clone(), y.t.float()     # the interpreter computes in fp32: host logic, not bf16 rounding, is under test
        run_plan_on_cpu(plan)
        ref = F.conv3d(x.permute(0, 4, 1, 2, 3), w, bias, 1, 1)
        ref = F.batch_norm(ref, bn[2], bn[3], bn[0], bn[1], False, 0.1, 1e-5)
        ref = torch.relu(ref + res.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1)
        assert float((y.t - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), float((y.t - ref).abs().max())
    os.environ["LT_CONV_NO_SPLITK"] = "1"
    try:
        b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
        b.conv(E.Act(torch.zeros(1, 4, 4, 4, 128)), w, bias, bn, stride=1, pad=1, relu=True)
        assert len(b.ops) == 1
    finally:
        del os.environ["LT_CONV_NO_SPLITK"]
    b = E.PlanBuilder("cpu", torch.float32, dry_run=True)
    b.conv(E.Act(torch.zeros(1, 4, 4, 4, 128)), w, bias, bn, stride=1, pad=1, relu=True)
    assert len(b.ops) == 1          # the exact-fp32 parity mode keeps its single accumulation chain


def test_identity_bottlenecks_are_recorded_as_one_launch_in_bf16_plans(monkeypatch):
    """bf16 plans run the identity Bottleneck blocks of ResNet layer1 / layer2 (256 / 64 and 512 / 128 wide, maps of 8 x 16 pixel tiles) as ONE
    lt_bottleneck_fwd each; the recorded block must be the same function as the three lt_conv_fwd launches LT_NO_BNECK=1 records, and
    blocks with a downsample branch / other widths / ragged maps keep the three launches."""
    import lt_engine as E
    from mvn.models.pose_resnet import PoseResNet
    torch.manual_seed(11)
    m = PoseResNet("bottleneck", [3, 4, 6, 3], 17).eval()
    for bn in [mm for mm in m.modules() if isinstance(mm, torch.nn.BatchNorm2d)]:
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.1)
    x = torch.randn(1, 1, 16, 32, 256)
    outs = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("LT_NO_BNECK", "1")
        b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
        inp = b.alloc(tuple(x.shape)); inp.pooled = False
        y = m.layer1[1].record(b, inp)
        plan = b.finish()
        labels = [meta["label"] for _, meta in plan.ops]
        assert len(labels) == (1 if fused else 3) and labels[0].startswith("bneck 256->64->256" if fused else "conv1x1 256->64")
        inp.t.copy_(x)
        run_plan_on_cpu(plan)
        outs[fused] = y.t.float().clone()
        assert y.t.data_ptr() != inp.t.data_ptr()
    assert torch.equal(outs[True], outs[False])           # the interpreter rounds the inner tensors where the launches would store them
    monkeypatch.delenv("LT_NO_BNECK")
    # what does NOT fuse: the first block of a strided level (downsample branch with stride 2), fp32 plans, maps that are not whole tiles, layer3's width
    for blk, shape, dt in ((m.layer2[0], (1, 1, 16, 32, 256), torch.bfloat16), (m.layer1[1], (1, 1, 16, 32, 256), torch.float32),
                           (m.layer1[1], (1, 1, 12, 32, 256), torch.bfloat16), (m.layer3[1], (1, 1, 8, 16, 1024), torch.bfloat16)):
        b = E.PlanBuilder("cpu", dt, dry_run=True)
        blk.record(b, b.alloc(shape))
        assert all(not meta["label"].startswith("bneck") for _, meta in b.finish().ops)
    b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
    m.layer2[3].record(b, b.alloc((2, 1, 8, 16, 512)))
    assert [meta["label"] for _, meta in b.finish().ops] == ["bneck 512->128->512 @2x1x8x16"]


def test_first_bottleneck_of_layer1_with_its_downsample_branch_is_one_launch_in_bf16_plans(monkeypatch):
    """Round 5: the FIRST block of ResNet layer1 (64 -> 64 -> 256, stride 1, `downsample` = conv1x1 + bn: pose_resnet.py:75-95, :196-206) is ONE
    lt_bottleneck_ds_fwd in bf16 plans.  It must be the function the four lt_conv_fwd launches compute, up to the one place where it is MORE exact:
    the downsample branch is added in fp32 instead of being stored in bf16 first (so: equal within one bf16 rounding of the branch, and equal to the
    fp32 module within the plan's rounding).  fp32 plans, ragged maps, LT_NO_BNECK_DS=1 and the strided first blocks of layer2-4 keep the launches."""
    import lt_engine as E
    from mvn.models.pose_resnet import PoseResNet
    torch.manual_seed(13)
    m = PoseResNet("bottleneck", [3, 4, 6, 3], 17).eval()
    for bn in [mm for mm in m.modules() if isinstance(mm, torch.nn.BatchNorm2d)]:
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.1)
    x = torch.randn(2, 1, 16, 32, 64)
    outs = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("LT_NO_BNECK_DS", "1")
            monkeypatch.setenv("LT_NO_CONV_CAT2", "1")       # (without it the block falls back to three launches: expand + downsample as one pointwise convolution)
        b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
        inp = b.alloc(tuple(x.shape)); inp.pooled = False
        y = m.layer1[0].record(b, inp)
        plan = b.finish()
        labels = [meta["label"] for _, meta in plan.ops]
        assert labels == (["bneck-ds 64->64->256 @2x1x16x32"] if fused else
                          ["conv1x1 64->256 @2x1x16x32", "conv1x1 64->64 @2x1x16x32", "conv3x3 64->64 @2x1x16x32", "conv1x1 64->256 @2x1x16x32"]), labels
        inp.t.copy_(x)
        run_plan_on_cpu(plan)
        outs[fused] = y.t.float().clone()
    monkeypatch.delenv("LT_NO_BNECK_DS")
    monkeypatch.delenv("LT_NO_CONV_CAT2")
    blk = m.layer1[0]
    xb = x[:, 0].permute(0, 3, 1, 2).to(torch.bfloat16).float()
    with torch.no_grad():
        ref = torch.relu(blk.bn3(blk.conv3(torch.relu(blk.bn2(blk.conv2(torch.relu(blk.bn1(blk.conv1(xb)))))))) + blk.downsample(xb)).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    d_sep = float((outs[True] - outs[False]).abs().max())
    assert d_sep <= 2.0 ** -7 * scale, (d_sep, scale)     # one bf16 rounding of the branch (relative 2^-9) plus one of the output
    e_f, e_s = float((outs[True][:, 0] - ref).abs().max()), float((outs[False][:, 0] - ref).abs().max())
    assert e_f <= 3e-2 * scale and e_f <= 1.5 * e_s + 1e-3, (e_f, e_s, scale)
    for blk2, shape, dt, env in ((m.layer1[0], (1, 1, 16, 32, 64), torch.float32, None), (m.layer1[0], (1, 1, 12, 32, 64), torch.bfloat16, None),
                                 (m.layer1[0], (1, 1, 16, 32, 64), torch.bfloat16, "LT_NO_BNECK"), (m.layer3[0], (1, 1, 16, 32, 512), torch.bfloat16, None)):
        if env:
            monkeypatch.setenv(env, "1")
        b = E.PlanBuilder("cpu", dt, dry_run=True)
        blk2.record(b, b.alloc(shape))
        assert all(not meta["label"].startswith("bneck") for _, meta in b.finish().ops)
        if env:
            monkeypatch.delenv(env)


def test_first_blocks_of_layer2_to_4_fold_the_downsample_branch_into_the_expand(monkeypatch):
    """Round 5: the first Bottleneck of ResNet layer2 / 3 / 4 (stride-2 `downsample` = conv1x1 + bn of the block input, pose_resnet.py:75-95, :196-206) records
    its expand and its downsample branch as ONE pointwise convolution over [t2 | x at the strided pixels] (lt_conv_cat2_fwd) in bf16 plans: three launches
    instead of four.  Same function as the four launches up to the rounding of the branch / of the scale-folded weights, equal to the fp32 module within the
    plan's rounding; fp32 plans and LT_NO_CONV_CAT2=1 keep the four launches."""
    import lt_engine as E
    from mvn.models.pose_resnet import PoseResNet
    torch.manual_seed(14)
    m = PoseResNet("bottleneck", [3, 4, 6, 3], 17).eval()
    for bn in [mm for mm in m.modules() if isinstance(mm, torch.nn.BatchNorm2d)]:
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.1)
    b0 = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)      # by default only from ~200 tiles of 288 x 256 on (small batches spread better as separate launches)
    w3, wd = m.layer2[0].conv3.weight, m.layer2[0].downsample[0].weight
    assert not b0.can_conv_cat2((2, 1, 6, 10, 128), w3, (2, 1, 12, 20, 256), wd, 2) and b0.can_conv_cat2((20, 1, 48, 48, 128), w3, (20, 1, 96, 96, 256), wd, 2)
    monkeypatch.setenv("LT_CAT2_ANY_SIZE", "1")
    for blk, cin, lab in ((m.layer2[0], 256, "conv1x1 128+256->512 @2x1x6x10 (expand + stride-2 downsample)"),
                          (m.layer3[0], 512, "conv1x1 256+512->1024 @2x1x6x10 (expand + stride-2 downsample)"),
                          (m.layer4[0], 1024, "conv1x1 512+1024->2048 @2x1x6x10 (expand + stride-2 downsample)")):
        x = torch.relu(torch.randn(2, 1, 12, 20, cin))
        outs = {}
        for fused in (True, False):
            if not fused:
                monkeypatch.setenv("LT_NO_CONV_CAT2", "1")
            b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
            inp = b.alloc(tuple(x.shape))
            y = blk.record(b, inp)
            plan = b.finish()
            labels = [meta["label"] for _, meta in plan.ops]
            assert len(labels) == (3 if fused else 4) and (labels[-1] == lab) == fused, labels
            inp.t.copy_(x)
            run_plan_on_cpu(plan)
            outs[fused] = y.t.float().clone()
        monkeypatch.delenv("LT_NO_CONV_CAT2")
        xb = x[:, 0].permute(0, 3, 1, 2).to(torch.bfloat16).float()
        with torch.no_grad():
            ref = torch.relu(blk.bn3(blk.conv3(torch.relu(blk.bn2(blk.conv2(torch.relu(blk.bn1(blk.conv1(xb)))))))) + blk.downsample(xb)).permute(0, 2, 3, 1)
        scale = float(ref.abs().max())
        assert float((outs[True] - outs[False]).abs().max()) <= 2.0 ** -5 * scale
        e_f, e_s = float((outs[True][:, 0] - ref).abs().max()), float((outs[False][:, 0] - ref).abs().max())
        assert e_f <= 3e-2 * scale and e_f <= 2.0 * e_s + 1e-3, (e_f, e_s, scale)
        b = E.PlanBuilder("cpu", torch.float32, dry_run=True)
        blk.record(b, b.alloc(tuple(x.shape)))
        assert len(b.finish().ops) == 4


def test_res3d_block_with_a_skip_convolution_records_the_skip_inside_the_second_convolution(monkeypatch):
    """Round 5: the 16 -> 32 Res3DBlock of V2V's 64^3 level (v2v.py:20-42, :76) in bf16 plans, on shapes the column-walk halo kernel takes: two launches
    (conv 16 -> 32, conv 32 -> 32 + computed skip: lt_conv_skip_fwd) instead of three.  The recorded block must be the function the three launches
    compute up to the rounding of the skip branch (stored in bf16 there; BatchNorm scale folded into its bf16 weights and added in fp32 here), and equal
    the fp32 module within the plan's rounding.  Small batches / volumes (too few tile columns), fp32 plans and LT_NO_CONV_SKIP=1 keep the three launches."""
    import lt_engine as E
    from mvn.models.v2v import Res3DBlock
    torch.manual_seed(21)
    blk = Res3DBlock(16, 32).eval()
    for bn in [mm for mm in blk.modules() if isinstance(mm, torch.nn.BatchNorm3d)]:
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.1)
    x = torch.relu(torch.randn(4, 16, 64, 64, 16))        # N, D, H, W, C: 256 columns of four tiles
    outs = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("LT_NO_CONV_SKIP", "1")
        b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
        inp = b.alloc(tuple(x.shape))
        y = blk.record(b, inp)
        plan = b.finish()
        labels = [meta["label"] for _, meta in plan.ops]
        assert labels == (["conv3x3x3 16->32 @4x16x64x64", "conv3x3x3 32->32 @4x16x64x64 + skip conv1x1x1 16->32"] if fused else
                          ["conv1x1x1 16->32 @4x16x64x64", "conv3x3x3 16->32 @4x16x64x64", "conv3x3x3 32->32 @4x16x64x64"]), labels
        inp.t.copy_(x)
        run_plan_on_cpu(plan)
        outs[fused] = y.t.float().clone()
    monkeypatch.delenv("LT_NO_CONV_SKIP")
    with torch.no_grad():
        xb = x.to(torch.bfloat16).float().permute(0, 4, 1, 2, 3)
        ref = torch.relu(blk.res_branch(xb) + blk.skip_con(xb)).permute(0, 2, 3, 4, 1)   # the container has no forward of its own (v2v.py:37-42)
    scale = float(ref.abs().max())
    assert float((outs[True] - outs[False]).abs().max()) <= 2.0 ** -6 * scale
    e_f, e_s = float((outs[True] - ref).abs().max()), float((outs[False] - ref).abs().max())
    assert e_f <= 3e-2 * scale and e_f <= 1.5 * e_s + 1e-3, (e_f, e_s, scale)
    for shape, dt in (((1, 64, 64, 64, 16), torch.bfloat16), ((4, 16, 64, 64, 16), torch.float32), ((4, 4, 64, 64, 16), torch.bfloat16)):
        b = E.PlanBuilder("cpu", dt, dry_run=True)
        blk.record(b, b.alloc(shape))
        assert len(b.finish().ops) == 3


def test_expand_reduce_seam_fusion_of_layer3_is_recorded_and_equals_the_separate_launches(monkeypatch):
    """Round 5: bf16 plans run the seam between two identity Bottleneck blocks of ResNet layer3 (1024 / 256 wide) as ONE lt_expand_reduce_fwd -- the
    expand of block i (+ residual + ReLU) and the reduce of block i + 1.  The recorded chain must be the same function as the separate lt_conv_fwd
    launches LT_NO_XR=1 records (the interpreter rounds y where the expand would store it), with 2 launches per block instead of 3 inside the run, the
    first block's reduce and the last block's expand as ordinary convolutions; fp32 plans, other widths and the training tape keep the three launches."""
    import lt_engine as E
    from mvn.models.pose_resnet import PoseResNet
    torch.manual_seed(12)
    m = PoseResNet("bottleneck", [3, 4, 6, 3], 17).eval()
    for bn in [mm for mm in m.modules() if isinstance(mm, torch.nn.BatchNorm2d)]:
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.2, 0.6); bn.bias.data.normal_(0, 0.1)
    x = torch.randn(2, 1, 6, 6, 1024)                      # 72 GEMM rows: one ragged tile
    b0 = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)          # by default the builder only fuses from 36 tiles of 96 rows (2 samples of 4 views) on: below, the two launches spread better
    assert not b0.can_expand_reduce(b0.alloc((2, 1, 6, 6, 256)), b0.alloc((2, 1, 6, 6, 1024)), m.layer3[1].conv3.weight, m.layer3[2].conv1.weight)
    assert b0.can_expand_reduce(b0.alloc((12, 1, 24, 24, 256)), b0.alloc((12, 1, 24, 24, 1024)), m.layer3[1].conv3.weight, m.layer3[2].conv1.weight)
    monkeypatch.setenv("LT_XR_ANY_SIZE", "1")

    def run_layer3_tail(b, inp):
        y, t1 = inp, None
        blocks = list(m.layer3)[1:5]                       # four identity blocks
        for bi, blk in enumerate(blocks):
            nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
            if nxt is not None:
                z, t1 = blk.record(b, y, t1=t1, next_block=nxt)
            else:
                z, t1 = (blk.record(b, y, t1=t1) if t1 is not None else blk.record(b, y)), None
            y = z
        return y
    outs, counts = {}, {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("LT_NO_XR", "1")
        b = E.PlanBuilder("cpu", torch.bfloat16, dry_run=True)
        inp = b.alloc(tuple(x.shape))
        y = run_layer3_tail(b, inp)
        plan = b.finish()
        labels = [meta["label"] for _, meta in plan.ops]
        counts[fused] = len(labels)
        inp.t.copy_(x)
        run_plan_on_cpu(plan)
        outs[fused] = y.t.float().clone()
        if fused:
            assert sum(l.startswith("xr expand 256->1024 + reduce 1024->256") for l in labels) == 3, labels
            assert labels[0].startswith("conv1x1 1024->256") and labels[-1].startswith("conv1x1 256->1024"), labels
    monkeypatch.delenv("LT_NO_XR")
    assert counts == {True: 9, False: 12}, counts          # reduce + 4 x 3x3 + 3 seams + last expand  vs  4 x 3
    assert torch.equal(outs[True], outs[False])
    # the whole backbone records the seams through PoseResNet.record; fp32 plans do not
    for dt, want in ((torch.bfloat16, 4), (torch.float32, 0)):          # layer3 of ResNet-50: six blocks, five identity ones: four seams
        b = E.PlanBuilder("cpu", dt, dry_run=True)
        m.record(b, b.alloc((1, 1, 64, 64, E.min_cin_of(dt))), False)
        labels = [meta["label"] for _, meta in b.finish().ops]
        assert sum(l.startswith("xr ") for l in labels) == want, (dt, [l for l in labels if l.startswith("xr ")])


def test_benchmark_model_plan_records_the_round5_fusions():
    """The launch structure of the benchmark's model (BASELINE config 2: ResNet-152, 4 views of 384 x 384, 64^3 voxels) as a bf16 plan at 8 samples, recorded
    dry: layer1's first block as one launch, 9 whole identity bottlenecks in layer1 / layer2, 34 expand + reduce seams in layer3, the first blocks of layer2 / 3
    with expand + strided downsample as one pointwise convolution (layer4's 128 rows x 8 column tiles stay below that kernel's size rule at 8 samples), V2V's
    16 -> 32 block with its skip convolution inside the second convolution, the pointwise tail as one chain -- and nothing of it in an fp32 plan."""
    from mvn.models.triangulation import VolumetricTriangulationNet
    cfg = synth.vol_config(152, 64, "softmax")
    m = VolumetricTriangulationNet(cfg, device="cpu")
    m.eval()
    counts = {}
    for dt in (torch.bfloat16, torch.float32):
        m.compute_dtype = dt
        P = m._build_plan(8 if dt == torch.bfloat16 else 1, 4, 384, 384, "cpu", dry_run=True)
        labels = [meta["label"] for _, meta in P["plan"].ops]
        counts[dt] = {"launches": len(labels), "xr": sum(l.startswith("xr ") for l in labels), "bneck": sum(l.startswith("bneck ") for l in labels),
                      "bneck-ds": sum(l.startswith("bneck-ds") for l in labels), "cat2": sum("downsample)" in l for l in labels),
                      "skip": sum("+ skip conv1x1x1" in l for l in labels), "pwchain": sum(l.startswith("pwchain") for l in labels),
                      "conv3x3 256->256": sum(l.startswith("conv3x3 256->256 @") for l in labels)}
        del P
    b, f = counts[torch.bfloat16], counts[torch.float32]
    assert (b["xr"], b["bneck"], b["bneck-ds"], b["cat2"], b["skip"], b["pwchain"]) == (34, 9, 1, 2, 1, 1), b
    assert b["conv3x3 256->256"] == 36              # layer3's 36 blocks: 35 stride-1 ones (the 2D halo kernel on the GPU) + the strided one of the first block
    assert (f["xr"], f["bneck"], f["bneck-ds"], f["cat2"], f["skip"], f["pwchain"]) == (0, 0, 0, 0, 0, 0), f
    assert b["launches"] <= f["launches"] - 40, (b["launches"], f["launches"])        # 177 against 220 when this was written
