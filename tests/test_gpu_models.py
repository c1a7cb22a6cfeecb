"""GPU (-m gpu): networks and the whole volumetric / algebraic forward through the reference-shaped module API,
against the committed REFERENCE outputs (tests/golden) and the oracle.

Gates (SURVEY.md section 8d, fp32 kernel mode): intermediates max|d| <= 1e-4 * max|ref| (coord volumes 1e-7, i.e.
bit-level); 3D joints max |ours - ref| / max(|ref|, 1 mm) <= 1e-4 on default AND sharpened weights.  bf16 mode is
not held to the gate: its measured deviation is recorded (the reference's own bf16 autocast deviates ~1e-2)."""
import os

import numpy as np
import pytest
import torch

import lt_hip as H
from gpu_util import check, record, rel_err
from oracle import spec, synth
from oracle import vol_oracle as O
from test_oracle_golden import VOL_CASES, build_vol_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sub(t, s):
    sl = (slice(None), slice(None)) + tuple(slice(None, None, s) for _ in range(t.dim() - 2))
    return t[sl]


def _cameras(inp, B):
    from mvn.utils.multiview import Camera
    return [[Camera(inp["R"][v], inp["t"][v], inp["K"][v]) for _ in range(B)] for v in range(inp["K"].shape[0])]


def test_v2v_vs_reference_golden(golden_dir):
    from mvn.models.v2v import V2VModel
    g = np.load(os.path.join(golden_dir, "nets.npz"))
    gen = torch.Generator().manual_seed(9)
    m = V2VModel(32, 17)
    m.load_state_dict(synth.make_state_dict(spec.v2v_spec(32, 17, ""), seed=3), strict=True)
    m.eval()
    x = torch.randn(1, 32, 32, 32, 32, generator=gen)
    y = m(x.to(DEV))
    check("v2v 32^3 fp32 vs reference", _sub(y.cpu(), 3), g["v2v_out_s3"], 1e-4)
    m.compute_dtype = torch.bfloat16; m._plans.clear()
    yb = m(x.to(DEV))
    record("v2v 32^3 bf16 deviation (max|d|/max|ref|)", rel_err(_sub(yb.cpu(), 3), g["v2v_out_s3"]))


@pytest.mark.parametrize("nl,hw,conf", [(152, 128, False), (50, 128, True), (18, 64, False)])
def test_pose_resnet_vs_reference_golden(golden_dir, nl, hw, conf):
    from mvn.models import pose_resnet
    g = np.load(os.path.join(golden_dir, "nets.npz"))
    gen = torch.Generator().manual_seed(9)
    torch.randn(1, 32, 32, 32, 32, generator=gen)  # keep the generator in step with oracle/make_golden.py
    for n2, h2, _ in ((152, 128, 0), (50, 128, 1), (18, 64, 0)):
        x = torch.randn(2, 3, h2, h2, generator=gen)
        if n2 == nl:
            break
    cfg = synth.AttrDict(num_layers=nl, style="simple", num_joints=17, alg_confidences=conf, vol_confidences=conf, init_weights=False, checkpoint="")
    m = pose_resnet.get_pose_net(cfg, device=DEV)
    m.load_state_dict(synth.make_state_dict(spec.pose_resnet_spec(nl, 17, conf, conf, ""), seed=nl, basic_block=(nl < 50)), strict=True)
    m.eval()
    hm, ft, ac, vc = m(x.to(DEV))
    check("resnet%d features fp32 vs reference" % nl, _sub(ft.cpu(), 2), g["rn%d_feat_s2" % nl], 1e-4)
    check("resnet%d heatmaps fp32 vs reference" % nl, hm.cpu(), g["rn%d_hm" % nl], 1e-4)
    if conf:
        check("resnet%d alg_confidences" % nl, ac.cpu(), g["rn%d_algc" % nl], 1e-4)
        check("resnet%d vol_confidences" % nl, vc.cpu(), g["rn%d_volc" % nl], 1e-4)
    m.compute_dtype = torch.bfloat16; m._plans.clear()
    _, ftb, _, _ = m(x.to(DEV))
    record("resnet%d bf16 feature deviation (max|d|/max|ref|)" % nl, rel_err(_sub(ftb.cpu(), 2), g["rn%d_feat_s2" % nl]))


def _run_vol(tag, dtype, use_graph=True):
    from mvn.models.triangulation import VolumetricTriangulationNet
    cfg, sd, inp, c = build_vol_case(tag)
    m = VolumetricTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.eval()
    m.compute_dtype = dtype
    m.use_graph = use_graph
    thetas = None
    if c["rotate"]:  # the reference draws theta with np.random.uniform in training mode (triangulation.py:318-319)
        m.training = True
        np.random.seed(c["seed"] + 100)
    batch = {"cameras": _cameras(inp, c["B"]), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    out = m(inp["images"].to(DEV), torch.zeros(c["B"], c["NV"], 3, 4, device=DEV), batch)
    return m, out, inp, c, batch


@pytest.mark.parametrize("tag", list(VOL_CASES))
def test_volumetric_forward_vs_reference_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "vol_%s.npz" % tag))
    m, (kp, feats, vols, conf, cuboids, cvs, bps), inp, c, batch = _run_vol(tag, torch.float32)
    s = int(g["stride"])
    B, NV = c["B"], c["NV"]
    assert kp.shape == (B, 17, 3) and feats.shape[:3] == (B, NV, 32) and vols.shape == (B, 17, c["V"], c["V"], c["V"])
    assert cvs.shape == (B, c["V"], c["V"], c["V"], 3) and bps.shape == (B, 3) and len(cuboids) == B
    check(tag + "/coord_volumes", cvs.cpu()[:, ::s, ::s, ::s], g["cv_sub"], 2e-7)
    check(tag + "/base_points", bps.cpu(), g["base_points"], 1e-7)
    # intermediates within 1e-4 * max|ref|; features and logits are gated 5x tighter than that (measured 2e-6)
    check(tag + "/features", _sub(feats.cpu().reshape(B * NV, *feats.shape[2:]), s), g["feat_sub"], 2e-5)
    check(tag + "/volumes (softmaxed)", _sub(vols.cpu(), s), g["vol_sub"], 1e-4)
    assert np.allclose(np.stack([cb.position for cb in cuboids]), g["cuboid_pos"]) and np.allclose(cuboids[0].sides, g["cuboid_sides"][0])
    if conf is not None:
        check(tag + "/vol_confidences", conf.cpu(), g["vol_conf"], 1e-4)
    rel = np.abs(kp.cpu().numpy() - g["kp"]) / np.maximum(np.abs(g["kp"]), 1.0)
    mpjpe = float(np.sqrt(((kp.cpu().numpy() - g["kp"]) ** 2).sum(-1)).mean())
    record(tag + "/joints fp32: max rel err (1 mm floor)", float(rel.max()))
    record(tag + "/joints fp32: MPJPE vs reference (mm)", mpjpe)
    # The north-star gate, against the reference's output AND against the exact (fp64) soft-argmax of the reference's own logits.
    # torch's fp32 softmax / einsum over V^3 voxels have a reduction error of their own (stored with the fixture: 2.9e-4 at 128^3, where
    # the reference's probabilities sum to 1 +- 2.9e-4; <= 3e-5 at 64^3): a kernel can be no closer to the reference than the
    # reference is to the exact value, so the first gate widens by that measured amount and the second one is the strict 1e-4.
    self_rel = float(g["ref_self_rel"])
    rel64 = np.abs(kp.cpu().double().numpy() - g["kp_fp64"]) / np.maximum(np.abs(g["kp_fp64"]), 1.0)
    if not c["volume_softmax"]:
        # volume_softmax: false -- the joints are the UNNORMALISED sums  sum_i relu(l_i) X_i  (op.py:91-95), whose three components cancel to very different
        # degrees (the voxel coordinates are symmetric about the cuboid centre): element-wise relative error is then a statement about the cancellation,
        # not about the kernel (the same 1e-6-of-max logit differences that leave the softmax cases at 1e-5 give 2e-4 on the most cancelled component).
        # The gate of this case is norm-wise: against the largest component of the sample's joints; the element-wise figure is recorded.
        record(tag + "/joints fp32 (ReLU volumes): element-wise max rel err vs the exact value, recorded", float(rel64.max()))
        scale = np.abs(g["kp_fp64"]).max(axis=(1, 2), keepdims=True)
        rel64 = np.abs(kp.cpu().double().numpy() - g["kp_fp64"]) / scale
        rel = np.abs(kp.cpu().numpy() - g["kp"]) / scale
    record(tag + "/joints fp32: max rel err vs the fp64 soft-argmax of the reference's logits", float(rel64.max()))
    record(tag + "/reference's own fp32 reduction error (max rel)", self_rel)
    # ref_order_rel (stored with c2_b8_sharp, round 6): how far the exact soft-argmax of the REFERENCE's logits moves when only the order of the reference's
    # fp32 sums changes (1 thread vs 8; oracle/make_golden.py, oracle/ref_noise.py) -- 3.0e-5 on that fixture, whose 408 coordinates with a 1 mm floor make it
    # the tightest of the set: the exact-fp32 kernels measured 0.80e-4 with one K order of V2V's 7^3 layer, 1.01e-4 and 1.05e-4 with two other orders of the SAME
    # products (the two-phase halo kernel flushing to fp64 every 4 / every 8 taps).  The reference's logits are defined up to that noise, so the strict gate widens by it where it was measured (0 elsewhere).
    order_rel = float(g["ref_order_rel"]) if "ref_order_rel" in g.files else 0.0
    if order_rel:
        record(tag + "/reference's own fp32 ORDER noise, 1 vs 8 threads (max rel of the exact soft-argmax of its logits)", order_rel)
    assert rel64.max() <= 1e-4 + order_rel, "joints vs exact soft-argmax of the reference logits: max rel %.3e" % rel64.max()
    assert rel.max() <= 1e-4 + self_rel + order_rel, "joints max rel %.3e (reference self error %.3e)" % (rel.max(), self_rel)
    # intermediates that the API does not return: unprojected volume and V2V logits, from the plan's buffers
    P = list(m._plans.values())[0]
    # graph replay == eager, and a second call is bit-identical (determinism)
    out2 = m(inp["images"].to(DEV), None, batch) if not c["rotate"] else None
    if out2 is not None:
        assert torch.equal(out2[0], kp) and torch.equal(out2[2], vols)
    logits = P["logits"].t.permute(0, 4, 1, 2, 3).float().cpu()
    check(tag + "/v2v logits", _sub(logits, s), g["logits_sub"], 2e-5)


# bf16 throughput mode against the REFERENCE's stored outputs: per-fixture gates (VERDICT r5 "next" 1a: the 60 mm "same skeleton, not garbage" bound this
# replaces said nothing about the kernels).  Round 6 measured every figure TWICE with the same kernel set: once with the BatchNorm fold's constants from torch's
# vectorised sqrt, once from the IEEE sqrt (a last-bit change of ~0.5 % of the scales, lt_engine.fold_bn).  Through 152 bf16 layers and a sharpened
# soft-argmax that moved the single-fixture figures by up to +53 % (small_softmax max abs 0.45 -> 0.69 mm, c4_sharp MPJPE 1.60 -> 2.19 mm) and others down
# (c2_sharp 3.51 -> 3.19 mm): the deviation of a bf16 forward from the fp32 reference is itself a noisy quantity.  Gates: 1.5 x the LARGER of the two measurements
# (profiles/r06_parity_report.json holds the final run).  Per tag:
# (joints MPJPE mm, joints max abs mm, features max|d|/max|ref|, V2V logits max|d|/max|ref|, softmaxed volumes max|d|/max|ref|)
BF16_GATES = {                  # the two measurements behind each row
    "small_softmax": (0.405, 1.03, 0.0177, 0.0135, 0.0444),       # 0.193 | 0.270 mm, 0.448 | 0.688 mm, 1.17e-2, 8.98e-3 | 8.75e-3, 2.28e-2 | 2.96e-2
    "c2_default": (0.0037, 0.0075, 0.0262, 0.0172, 0.00115),      # 0.0024 | 0.0023 mm, 0.0050 | 0.0035 mm, 1.67e-2 | 1.75e-2, 1.14e-2 | 1.08e-2, 7.6e-4 | 7.3e-4
    "c2_sharp": (5.26, 44.1, 0.0262, 0.0186, 0.104),              # 3.51 | 3.19 mm, 29.4 | 24.1 mm (one joint of a near-argmax volume), 1.75e-2, 1.24e-2 | 1.18e-2, 5.3e-2 | 6.9e-2
    "c2_b4": (2.51, 11.4, 0.0281, 0.0223, 0.0833),                # 1.67 | 1.49 mm, 5.79 | 7.56 mm, 1.87e-2 | 1.75e-2, 1.31e-2 | 1.49e-2, 5.0e-2 | 5.6e-2
    "c2_b8_sharp": (1.10, 5.6, 0.0283, 0.0172, 0.118),            # 0.73 | 0.64 mm, 3.57 | 3.73 mm, 1.88e-2 | 1.81e-2, 1.14e-2 | 1.02e-2, 4.8e-2 | 7.9e-2 (plain and all-cat2 runs)
    "c4_sharp": (3.28, 12.1, 0.0257, 0.0284, 0.098),              # 1.60 | 2.19 mm, 5.28 | 8.06 mm, 1.71e-2 | 1.65e-2, 1.68e-2 | 1.89e-2, 6.5e-2 | 6.5e-2
}


def _plan_conv_kernels(m):
    """What the recorded plan of ``m`` is made of, from the op labels and the convolution descriptors it keeps alive: {kind: count}."""
    P = list(m._plans.values())[0]
    labels = [meta["label"] for _, meta in P["plan"].ops]
    halo2d = sum(1 for k in P["plan"].keep if isinstance(k, H.ConvDesc) and k.D == 1 and k.Cin == 256 and k.Cout == 256 and k.phase[0].weight_frag_layout == 2)
    return {"xr": sum(l.startswith("xr ") for l in labels), "bneck": sum(l.startswith("bneck ") for l in labels),
            "bneck_ds": sum(l.startswith("bneck-ds") for l in labels), "cat2": sum("downsample)" in l for l in labels), "conv2d_halo": halo2d,
            "pwchain": sum(l.startswith("pwchain") for l in labels), "stem": sum(l.startswith("stem ") for l in labels)}


def _bf16_vs_reference(golden_dir, tag, monkeypatch=None, env=()):
    for k, v in env:
        monkeypatch.setenv(k, v)
    g = np.load(os.path.join(golden_dir, "vol_%s.npz" % tag))
    m, (kp, feats, vols, conf, cuboids, cvs, bps), inp, c, batch = _run_vol(tag, torch.bfloat16)
    s = int(g["stride"])
    B, NV = c["B"], c["NV"]
    d = kp.cpu().numpy() - g["kp"]
    mpjpe, mabs = float(np.sqrt((d ** 2).sum(-1)).mean()), float(np.abs(d).max())
    P = list(m._plans.values())[0]
    logits = P["logits"].t.permute(0, 4, 1, 2, 3).float().cpu()
    name = tag + ("" if not env else " [" + " ".join("%s=%s" % kv for kv in env) + "]")
    e_f = rel_err(_sub(feats.cpu().reshape(B * NV, *feats.shape[2:]), s), g["feat_sub"])
    e_l = rel_err(_sub(logits, s), g["logits_sub"])
    e_v = rel_err(_sub(vols.cpu(), s), g["vol_sub"])
    record(name + "/joints bf16: MPJPE vs reference (mm)", mpjpe)
    record(name + "/joints bf16: max abs vs reference (mm)", mabs)
    record(name + "/joints bf16: max rel err (1 mm floor)", float((np.abs(d) / np.maximum(np.abs(g["kp"]), 1.0)).max()))
    record(name + "/features bf16 vs reference (max|d|/max|ref|)", e_f)
    record(name + "/v2v logits bf16 vs reference (max|d|/max|ref|)", e_l)
    record(name + "/volumes bf16 vs reference (max|d|/max|ref|)", e_v)
    kernels = _plan_conv_kernels(m)
    record(name + "/bf16 plan kernels", kernels)
    assert np.isfinite(kp.cpu().numpy()).all()
    for what, got, gate in zip(("MPJPE mm", "max abs mm", "features", "logits", "volumes"), (mpjpe, mabs, e_f, e_l, e_v), BF16_GATES[tag]):
        assert gate is None or got <= gate, "%s: %s = %.4g > gate %.4g (1.5 x the larger of the two round-6 measurements)" % (name, what, got, gate)
    return kernels


@pytest.mark.parametrize("tag", list(BF16_GATES))
def test_volumetric_forward_bf16_deviation(golden_dir, tag):
    """bf16 throughput mode against the reference's fp32 outputs (not held to the 1e-4 gate -- SURVEY 8d: "bf16: report deviation"), gated per fixture at
    1.5 x what this kernel set measures: joints (MPJPE and max abs, mm), features, V2V logits, softmaxed volumes.  ``c2_b8_sharp`` (round 6) is the smallest
    batch whose bf16 plan is made of the kernels the timed forward runs: asserted below."""
    kernels = _bf16_vs_reference(golden_dir, tag)
    if tag == "c2_b8_sharp":       # 32 images: the 2D halo kernel (35 3x3 of layer3 + 2 transposed), the 34 seams, layer1's fused first block, 9 whole bottlenecks
        assert kernels["conv2d_halo"] >= 37 and kernels["xr"] == 34 and kernels["bneck_ds"] == 1 and kernels["bneck"] == 9 and kernels["cat2"] >= 2, kernels
        assert kernels["pwchain"] == 1 and kernels["stem"] == 1, kernels


def test_volumetric_forward_bf16_every_first_block_fused(golden_dir, monkeypatch):
    """The same B = 8 reference golden with lt_conv_cat2_fwd forced on for ALL THREE strided first blocks (LT_CAT2_ANY_SIZE=1: layer4's has 128 tiles at 32
    images, below the builder's 200-tile rule, and would otherwise only ever run at the benchmark's batch): same gates."""
    kernels = _bf16_vs_reference(golden_dir, "c2_b8_sharp", monkeypatch, (("LT_CAT2_ANY_SIZE", "1"),))
    assert kernels["cat2"] == 3, kernels


def test_eager_equals_graph_and_batch_independence():
    """Size-independent properties at a real shape: (a) hipGraph replay == eager launches bit for bit; (b) a sample's result
    does not depend on what else is in the batch (samples are independent units -> sharding across ranks is exact)."""
    from mvn.models.triangulation import VolumetricTriangulationNet
    # moderately sharp soft-argmax (logit std ~1): sensitive to errors but away from the noise floor the sharpened goldens probe
    cfg = synth.vol_config(50, 32, "softmax", multiplier=50.0)
    sd = synth.make_state_dict(spec.vol_net_spec(50, 17), seed=21, sharpen=False)
    inp = synth.make_inputs(8, 4, 128, seed=21)
    outs = {}
    for name, graph, idx in (("graph8", True, slice(0, 8)), ("eager8", False, slice(0, 8)), ("graph_2", True, slice(2, 3))):
        m = VolumetricTriangulationNet(cfg, device=DEV); m.load_state_dict(sd); m.eval(); m.use_graph = graph
        B = idx.stop - idx.start
        batch = {"cameras": _cameras(inp, B), "pred_keypoints_3d": inp["pred_keypoints_3d"][idx]}
        outs[name] = m(inp["images"][idx].to(DEV), None, batch)
    assert torch.equal(outs["graph8"][0], outs["eager8"][0]) and torch.equal(outs["graph8"][2], outs["eager8"][2])
    # B=1 and B=8 may pick different tiles / kernels for the same layer (different summation order): equal to rounding, not bitwise
    d21 = float((outs["graph8"][0][2:3] - outs["graph_2"][0]).abs().max())
    record("batch independence: sample 2 in B=8 vs alone, joints max abs diff (mm)", d21)
    assert d21 <= 1e-2, "sample 2 differs between B=8 and B=1"
    o = O.volumetric_forward(sd, cfg, inp["images"][:2], inp["K"], inp["R"], inp["t"], inp["pred_keypoints_3d"][:2], stages=True)
    rel = ((outs["graph8"][0][:2].cpu() - o["keypoints_3d"]).abs() / o["keypoints_3d"].abs().clamp(min=1.0)).max()
    record("B=8 XCD-pinned path vs oracle: joints max rel", float(rel))
    record("B=8 XCD-pinned path vs oracle: joints max abs (mm)", float((outs["graph8"][0][:2].cpu() - o["keypoints_3d"]).abs().max()))
    record("B=8 XCD-pinned path vs oracle: features rel", rel_err(outs["graph8"][1][:2].cpu(), o["features"]))
    record("B=8 XCD-pinned path vs oracle: volumes rel", rel_err(outs["graph8"][2][:2].cpu(), o["volumes"]))
    record("B=8 XCD-pinned path vs oracle: coord rel", rel_err(outs["graph8"][5][:2].cpu(), o["coord_volumes"]))
    # near-uniform soft-argmax: joints sit within a few mm of the world origin, where a RELATIVE gate is meaningless
    # (fp32 sum of 32768 terms of magnitude ~1e3 mm): gate the absolute deviation at 0.01 mm = 4e-6 of the cuboid side
    assert float((outs["graph8"][0][:2].cpu() - o["keypoints_3d"]).abs().max()) <= 1e-2
    p = outs["graph8"][2]
    assert float((p.sum(dim=(2, 3, 4)) - 1).abs().max()) < 1e-4                      # probabilities sum to one
    lo = outs["graph8"][5].amin(dim=(1, 2, 3)); hi = outs["graph8"][5].amax(dim=(1, 2, 3))
    assert bool(((outs["graph8"][0] >= lo[:, None]) & (outs["graph8"][0] <= hi[:, None])).all())  # joints inside the cuboid


def _dev_err(a, ref):
    """(max|d| / max|ref|, rms(d) / rms(ref)) of two equally shaped device tensors, evaluated on the device in fp64 sample by sample (the benchmark's
    17 x 64^3 tensors at 64 samples are 1.1 GB each)."""
    mx = mr = sd = sr = 0.0
    for i in range(a.shape[0]):
        x, r = a[i].double(), ref[i].double()
        mx = max(mx, float((x - r).abs().max())); mr = max(mr, float(r.abs().max()))
        sd += float(((x - r) ** 2).sum()); sr += float((r ** 2).sum())
    return mx / max(mr, 1e-30), (sd / max(sr, 1e-300)) ** 0.5


# the timed kernel set against the fp32 parity kernels at the benchmark's batch, SHARPENED weights (what bench.py times): gates at 1.5 x the round-6
# measurement -- the larger of the two described at BF16_GATES (profiles/r06_parity_report.json).  Per batch: (MPJPE mm, max abs mm, features max-rel, features rms, logits max-rel, logits rms, volumes max-rel)
BENCH_SHAPE_GATES = {32: (4.33, 43.4, 0.0319, 0.0187, 0.0210, 0.0111, 0.089),          # 2.88 | 2.86 mm, 23.1 | 28.9 mm, 1.78e-2 | 2.12e-2, 1.24e-2, 1.39e-2, 7.4e-3, 4.5e-2 | 5.9e-2
                     64: (4.37, 48.2, 0.0315, 0.0187, 0.0206, 0.0111, 0.105)}          # 2.91 | 2.87 mm, 22.4 | 32.1 mm, 1.77e-2 | 2.10e-2, 1.24e-2, 1.37e-2 | 1.30e-2, 7.4e-3, 7.0e-2 | 6.5e-2


@pytest.mark.parametrize("B", [32, 64])
def test_bf16_kernel_set_of_the_benchmark_vs_fp32_kernels(B):
    """The exact kernel set bench.py times (bf16, B samples x 4 views x 384^2, ResNet-152, 64^3 volume -- 64 is the driver line's batch since round 4,
    32 rounds 1-3's and still in its batch_sweep: fused stem reading the fp32 images, whole-bottleneck launches in layer1 / layer2, the expand + reduce seam
    launches and the 2D halo kernel of layer3, conv_cat2, conv_igemm7/6/3/2, conv_pw, column-walk 3^3, kd-blocked 7^3, quad unproject, pwchain with planar
    logits, vectorised soft-argmax as tail op) against the fp32 parity kernels on the same inputs and weights.

    Round 6 (VERDICT r5 "next" 1a): SHARPENED weights -- the soft-argmax then depends on the input (with the default output gain the joints are the cube's
    centroid whatever the backbone computes) -- and every stage downstream of the backbone is gated: V2V logits and softmaxed volumes (max-rel and rms) as well
    as the features, joints in mm; plus one sample of the batch run alone (B = 1 picks different kernels for most layers)."""
    from mvn.models.triangulation import VolumetricTriangulationNet
    cfg = synth.vol_config(152, 64, "softmax", 1.0)
    sd = synth.make_state_dict(spec.vol_net_spec(152, 17), seed=0, sharpen=True)
    inp = synth.make_inputs(B, 4, 384, seed=5)
    images = inp["images"].to(DEV)
    outs, logits = {}, {}
    for name, dtype, idx in (("f32", torch.float32, slice(0, B)), ("bf16", torch.bfloat16, slice(0, B)), ("bf16_one", torch.bfloat16, slice(5, 6))):
        m = VolumetricTriangulationNet(cfg, device=DEV); m.load_state_dict(sd); m.eval(); m.compute_dtype = dtype
        n = idx.stop - idx.start
        batch = {"cameras": _cameras(inp, n), "pred_keypoints_3d": inp["pred_keypoints_3d"][idx]}
        o = m(images[idx], None, batch)
        o2 = m(images[idx], None, batch)     # second call = graph replay into fresh output tensors
        assert torch.equal(o[0], o2[0]) and o[2].data_ptr() != o2[2].data_ptr()
        outs[name] = o
        logits[name] = list(m._plans.values())[0]["logits"].t.permute(0, 4, 1, 2, 3).float().clone()
        if name == "bf16":
            kernels = _plan_conv_kernels(m)
            record("bench shape B=%d: bf16 plan kernels" % B, kernels)
            assert kernels["conv2d_halo"] == 37 and kernels["xr"] == 34 and kernels["bneck_ds"] == 1 and kernels["bneck"] == 9 and kernels["cat2"] == 3, kernels
        del m, o2
        torch.cuda.empty_cache()
    kp32, kp16, kp1 = outs["f32"][0], outs["bf16"][0], outs["bf16_one"][0]
    mpjpe = float((kp16 - kp32).norm(dim=-1).mean())
    mabs = float((kp16 - kp32).abs().max())
    f_max, f_rms = _dev_err(outs["bf16"][1], outs["f32"][1])
    l_max, l_rms = _dev_err(logits["bf16"], logits["f32"])
    v_max, v_rms = _dev_err(outs["bf16"][2], outs["f32"][2])
    one = float((kp16[5:6] - kp1).abs().max())
    record("bench shape B=%d (sharpened): joints bf16 kernels vs fp32 kernels, MPJPE (mm)" % B, mpjpe)
    record("bench shape B=%d (sharpened): joints bf16 kernels vs fp32 kernels, max abs (mm)" % B, mabs)
    record("bench shape B=%d (sharpened): features bf16 vs fp32 (max|d|/max|ref|, rms)" % B, [f_max, f_rms])
    record("bench shape B=%d (sharpened): v2v logits bf16 vs fp32 (max|d|/max|ref|, rms)" % B, [l_max, l_rms])
    record("bench shape B=%d (sharpened): volumes bf16 vs fp32 (max|d|/max|ref|, rms)" % B, [v_max, v_rms])
    record("bench shape (sharpened): sample 5 in B=%d vs alone (bf16), joints max abs (mm)" % B, one)
    assert torch.isfinite(kp16).all()
    assert float((outs["bf16"][2].sum(dim=(2, 3, 4)) - 1).abs().max()) < 1e-3
    for what, got, gate in zip(("MPJPE mm", "max abs mm", "features max", "features rms", "logits max", "logits rms", "volumes max"),
                               (mpjpe, mabs, f_max, f_rms, l_max, l_rms, v_max), BENCH_SHAPE_GATES[B]):
        assert gate is None or got <= gate, "bench shape B=%d: %s = %.4g > gate %.4g (1.5 x the round-6 measurement)" % (B, what, got, gate)
    # one sample of the batch alone: other kernels for most layers, the same bf16 arithmetic up to summation order -- within the batch's own deviation from fp32
    assert one <= max(mabs, 1e-3), (one, mabs)


def test_panoptic_shape_8_views_128_cube():
    """BASELINE config 4 (8 synthetic views, 128^3 voxel cube): every volumetric kernel at 8x the voxel count and with the generic
    (not the 4-view quad) unprojection; bf16 kernels against the fp32 parity kernels, ResNet-50 backbone to keep the test short."""
    from mvn.models.triangulation import VolumetricTriangulationNet
    B, NV, V = 2, 8, 128
    cfg = synth.vol_config(50, V, "softmax", 1.0)
    sd = synth.make_state_dict(spec.vol_net_spec(50, 17), seed=4, sharpen=False)
    inp = synth.make_inputs(B, NV, 256, seed=9)
    images = inp["images"].to(DEV)
    outs = {}
    for name, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m = VolumetricTriangulationNet(cfg, device=DEV); m.load_state_dict(sd); m.eval(); m.compute_dtype = dtype
        batch = {"cameras": _cameras(inp, B), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
        o = m(images, None, batch)
        outs[name] = [t.float().cpu() if torch.is_tensor(t) else t for t in o]
        del m
        torch.cuda.empty_cache()
    kp32, kp16 = outs["f32"][0], outs["bf16"][0]
    assert tuple(outs["bf16"][2].shape) == (B, 17, V, V, V)
    record("8 views / 128^3: joints bf16 kernels vs fp32 kernels, MPJPE (mm)", float((kp16 - kp32).norm(dim=-1).mean()))
    record("8 views / 128^3: volumes bf16 vs fp32 (max|d|/max|ref|)", rel_err(outs["bf16"][2], outs["f32"][2]))
    assert torch.isfinite(kp16).all() and float((kp16 - kp32).norm(dim=-1).mean()) < 0.1
    assert float((outs["bf16"][2].sum(dim=(2, 3, 4)) - 1).abs().max()) < 1e-3


def test_config4_at_32_samples_is_one_plan():
    """BASELINE config 4 (8 views, 128^3 voxels) at 32 samples per step: the 32-channel volumes have exactly 2^31 elements.  Until round 5 forward() cut such a
    batch into sub-batches of the MODEL; now it is one plan (one captured graph), and the convolution entry points walk the oversize tensors in sample chunks
    with 64-bit base pointers.  fp32 kernels (the arithmetic that is exact up to summation order), ResNet-18 backbone to keep it short, sharpened
    weights: every sample of the 32 equals the same sample run in a batch of 16 -- joints 1e-4 relative, features 1e-5 and softmaxed volumes 1e-4 of max (SURVEY 8d's gates)."""
    import warnings
    from mvn.models.triangulation import VolumetricTriangulationNet
    B, NV, V = 32, 8, 128
    cfg = synth.vol_config(18, V, "softmax", 1.0)
    sd = synth.make_state_dict(spec.vol_net_spec(18, 17), seed=12, sharpen=157.0, basic_block=True)
    inp = synth.make_inputs(B, NV, 128, seed=12)
    images = inp["images"].to(DEV)
    m = VolumetricTriangulationNet(cfg, device=DEV); m.load_state_dict(sd); m.eval(); m.compute_dtype = torch.float32
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # the old path announced its sub-batches with a warning: there must be none
        o32 = m(images, None, {"cameras": _cameras(inp, B), "pred_keypoints_3d": inp["pred_keypoints_3d"]})
    assert len(m._plans) == 1 and list(m._plans)[0][0] == B, "32 samples must be ONE plan"
    kp32, f32_, v32 = o32[0].clone(), o32[1].clone(), o32[2]
    assert torch.isfinite(kp32).all() and tuple(v32.shape) == (B, 17, V, V, V)
    worst = 0.0
    for lo in (0, 16):
        o16 = m(images[lo:lo + 16], None, {"cameras": _cameras(inp, 16), "pred_keypoints_3d": inp["pred_keypoints_3d"][lo:lo + 16]})
        rel = float(((o16[0] - kp32[lo:lo + 16]).abs() / kp32[lo:lo + 16].abs().clamp(min=1.0)).max())
        worst = max(worst, rel)
        assert rel <= 1e-4, "samples %d..: joints of the 32-sample call differ from the 16-sample call by %.3e" % (lo, rel)
        fe = float((o16[1] - f32_[lo:lo + 16]).abs().max() / f32_.abs().max())
        ve = max(float((o16[2][i] - v32[lo + i]).abs().max()) for i in range(16)) / float(v32.max())
        assert fe <= 1e-5 and ve <= 1e-4, (fe, ve)
        del o16
    record("config 4 at 32 samples (one plan, conv entry points chunk the 2^31-element tensors) vs 16-sample calls: joints max rel", worst)


@pytest.mark.parametrize("nl,hw", [(50, 128), (18, 64)])
def test_pose_resnet_caffe_style_vs_reference_golden(golden_dir, nl, hw):
    """style == 'caffe' (reference pose_resnet.py:98-137, :322-324): stride on the first 1x1; at depth 18 the reference still
    builds expansion-4 bottlenecks.  fp32 kernels against the reference's outputs."""
    from mvn.models import pose_resnet
    g = np.load(os.path.join(golden_dir, "nets_caffe.npz"))
    gen = torch.Generator().manual_seed(31)
    for n2, h2 in ((50, 128), (18, 64)):
        x = torch.randn(2, 3, h2, h2, generator=gen)
        if n2 == nl:
            break
    cfg = synth.AttrDict(num_layers=nl, style="caffe", num_joints=17, alg_confidences=False, vol_confidences=False, init_weights=False, checkpoint="")
    m = pose_resnet.get_pose_net(cfg, device=DEV)
    m.load_state_dict(synth.make_state_dict(spec.pose_resnet_spec(nl, 17, False, False, "", caffe=True), seed=700 + nl), strict=True)
    m.eval()
    hm, ft, _, _ = m(x.to(DEV))
    check("caffe resnet%d features fp32 vs reference" % nl, _sub(ft.cpu(), 2), g["rn%d_feat_s2" % nl], 1e-4)
    check("caffe resnet%d heatmaps fp32 vs reference" % nl, hm.cpu(), g["rn%d_hm" % nl], 1e-4)


def test_algebraic_tail_on_rendered_heatmaps_vs_reference(golden_dir):
    """Well-conditioned algebraic pipeline, END TO END gate on the 3D joints: heatmaps rendered from a projected
    skeleton -> lt_softargmax2d_fwd (x100, softmax) -> heatmap->image scaling -> confidence-weighted lt_triangulate_dlt, against the
    reference's op.integrate_tensor_2d + multiview.triangulate_batch_of_points on the same heatmaps."""
    from mvn.utils import multiview, op
    g = np.load(os.path.join(golden_dir, "pipe2d.npz"))
    hm = torch.from_numpy(g["hm"]).to(DEV)
    B, NV, J, h, _ = hm.shape
    kp2d, hm_sm = op.integrate_tensor_2d(hm.reshape(B * NV, J, h, h) * 100.0, True)
    conf = torch.from_numpy(g["conf"]).to(DEV)
    conf = conf / conf.sum(dim=1, keepdim=True) + 1e-5
    kp2d = kp2d.reshape(B, NV, J, 2) * (256 / h)
    kp3d = multiview.triangulate_batch_of_points(torch.from_numpy(g["P"]).to(DEV), kp2d, confidences_batch=conf)
    check("pipe2d/keypoints_2d (image px)", kp2d.cpu(), g["kp2d"], 1e-5)
    check("pipe2d/heatmaps after softmax", _sub(hm_sm.cpu(), 4), g["hm_sm_sub"], 1e-4)
    rel = np.abs(kp3d.cpu().numpy() - g["kp3d"]) / np.maximum(np.abs(g["kp3d"]), 1.0)
    record("pipe2d/keypoints_3d end to end: max rel (1 mm floor)", float(rel.max()))
    record("pipe2d/keypoints_3d vs the true skeleton (mm)", float(np.abs(kp3d.cpu().numpy() - g["X"]).max()))
    assert rel.max() <= 1e-3, rel.max()


def test_algebraic_c1_vs_reference_golden(golden_dir):
    """BASELINE config 1: algebraic triangulation, 4 x 256^2, ResNet-50 with confidences."""
    from mvn.models.triangulation import AlgebraicTriangulationNet
    g = np.load(os.path.join(golden_dir, "alg_c1.npz"))
    cfg = synth.alg_config(50, True)
    m = AlgebraicTriangulationNet(cfg, device=DEV)
    m.load_state_dict(synth.make_state_dict(spec.alg_net_spec(50, 17, True), seed=50), strict=True)
    m.eval()
    inp = synth.make_inputs(2, 4, 256, seed=1)
    P = torch.from_numpy(inp["K"] @ np.concatenate([inp["R"], inp["t"]], -1)).float()[None].repeat(2, 1, 1, 1)
    kp3, kp2, hm, conf = m(inp["images"].to(DEV), P.to(DEV), {})
    from mvn.utils import multiview
    check("alg/keypoints_2d", kp2.cpu(), g["kp2"], 1e-4)
    check("alg/confidences", conf.cpu(), g["conf"], 1e-4)
    check("alg/heatmaps", _sub(hm.cpu().reshape(8, 17, 64, 64), 4), g["hm_sub"], 2e-3)
    # With random weights the four views' 2D keypoints are mutually inconsistent, so the DLT systems are ill conditioned
    # (sigma3/sigma4 ~ 1.04..1.26, |X| up to 1.4e5 mm): a 5e-5 relative change of the 2D inputs moves the 3D output by 1.3e-2, and
    # the reference's own fp32 SVD is 3.3e-4 away from the fp64 solution of ITS inputs.  So: (a) the DLT kernel is gated on the
    # reference's 2D keypoints / confidences, (b) the end-to-end 3D deviation is recorded.
    k3 = multiview.triangulate_batch_of_points(P.to(DEV), torch.from_numpy(g["kp2"]).to(DEV), torch.from_numpy(g["conf"]).to(DEV))
    check("alg/keypoints_3d from the reference's 2D keypoints", k3.cpu(), g["kp3"], 1e-3)
    record("alg/keypoints_3d end-to-end deviation (ill-conditioned, see test)", rel_err(kp3.cpu(), g["kp3"]))
    assert torch.isfinite(kp3).all()


def test_algebraic_relu_heatmaps_without_confidences_vs_reference_golden(golden_dir):
    """AlgebraicTriangulationNet with use_confidences false (uniform weights) and heatmap_softmax false (ReLU heatmaps normalised by their mass,
    op.py:11-18): 2D keypoints, heatmaps and confidences against the reference; the DLT on the reference's 2D keypoints (ill-conditioned end to end at
    random init, as in the c1 case)."""
    from mvn.models.triangulation import AlgebraicTriangulationNet
    from mvn.utils import multiview
    g = np.load(os.path.join(golden_dir, "alg_relu_noconf.npz"))
    cfg = synth.alg_config(18, False)
    cfg.model.heatmap_softmax = False
    cfg.model.heatmap_multiplier = 1.0
    cfg.model["heatmap_softmax"] = False
    cfg.model["heatmap_multiplier"] = 1.0
    m = AlgebraicTriangulationNet(cfg, device=DEV)
    m.load_state_dict(synth.make_state_dict(spec.alg_net_spec(18, 17, False), seed=51, basic_block=True), strict=True)
    m.eval()
    inp = synth.make_inputs(2, 3, 128, seed=9)
    P = torch.from_numpy(inp["K"] @ np.concatenate([inp["R"], inp["t"]], -1)).float()[None].repeat(2, 1, 1, 1)
    kp3, kp2, hm, conf = m(inp["images"].to(DEV), P.to(DEV), {})
    check("alg-relu/keypoints_2d", kp2.cpu(), g["kp2"], 1e-4)
    check("alg-relu/heatmaps", _sub(hm.cpu().reshape(6, 17, 32, 32), 2), g["hm_sub"], 2e-3)
    check("alg-relu/confidences (ones, normalised over the views, + 1e-5)", conf.cpu(), g["conf"], 1e-6)
    k3 = multiview.triangulate_batch_of_points(P.to(DEV), torch.from_numpy(g["kp2"]).to(DEV), torch.from_numpy(g["conf"]).to(DEV))
    check("alg-relu/keypoints_3d from the reference's 2D keypoints", k3.cpu(), g["kp3"], 1e-3)
    record("alg-relu/keypoints_3d end-to-end deviation (ill-conditioned)", rel_err(kp3.cpu(), g["kp3"]))
    assert torch.isfinite(kp3).all()


def test_loud_failure_modes():
    from mvn.models.triangulation import VolumetricTriangulationNet
    m = VolumetricTriangulationNet(synth.vol_config(18, 32, "softmax"), device=DEV)
    inp = synth.make_inputs(1, 2, 64)
    batch = {"cameras": _cameras(inp, 1), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    m.train()
    with pytest.raises(RuntimeError, match="move the model"):            # training updates the parameters in place: they must live on the GPU
        m(inp["images"].to(DEV), None, batch)
    m.to(DEV)
    with pytest.raises(RuntimeError, match="GPU"):                        # no CPU fallback anywhere
        m.eval()(inp["images"], None, batch)
    info = H.device_info()
    record("device", info)
    assert info["arch"].startswith("gfx950"), info
