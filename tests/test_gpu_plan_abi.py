"""GPU (-m gpu): the PLAN-LEVEL C ABI (include/lt_hip.h: lt_plan_create_vol / lt_plan_forward_vol / lt_plan_info / lt_plan_destroy; SURVEY.md section 8b,
VERDICT r5 "next" 5) driven through ctypes ALONE -- this file imports neither lt_engine nor the mvn package for the runs that are gated against the
REFERENCE's golden outputs: the state dict goes in as names + host fp32 arrays, the cameras as host fp64 K / R / t, the images as a device pointer, and layer ->
kernel selection, weight packing, the BatchNorm fold, buffer reuse and the hipGraph are the library's.  torch appears as the owner of device memory only.

Gates: the fp32 plans at the gates of tests/test_gpu_models.py (joints 1e-4 relative against the exact soft-argmax of the reference's logits and
against the reference's own output widened by its measured reduction error, features / logits 2e-5, volumes 1e-4, coordinates 2e-7); the bf16 plan of the
benchmark shape at B = 8 at the per-fixture gates of the Python path (BF16_GATES), and -- last test -- bit-identical to the Python host's plan."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

import lt_hip as H
from gpu_util import check, record, rel_err
from oracle import spec, synth
from test_oracle_golden import VOL_CASES, build_vol_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sub(t, s):
    sl = (slice(None), slice(None)) + tuple(slice(None, None, s) for _ in range(t.dim() - 2))
    return t[sl]


def _rotation(axis, theta):          # Euler-Rodrigues, as volumetric.get_rotation_matrix of the reference (mvn/utils/volumetric.py:87-99)
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / math.sqrt(float(ax @ ax))
    a = math.cos(theta / 2.0)
    b, c, d = (-ax * math.sin(theta / 2.0)).tolist()
    return np.array([[a * a + b * b - c * c - d * d, 2.0 * (b * c + a * d), 2.0 * (b * d - a * c)],
                     [2.0 * (b * c - a * d), a * a + c * c - b * b - d * d, 2.0 * (c * d + a * b)],
                     [2.0 * (b * d + a * c), 2.0 * (c * d - a * b), a * a + d * d - b * b - c * c]], dtype=np.float64)


class CPlan:
    """lt_plan_* through ctypes: what a host in any language does."""

    def __init__(self, tag, dtype, use_graph=True):
        cfg, sd, inp, c = build_vol_case(tag)
        self.c, self.inp = c, inp
        lib = H.lib()
        m = cfg.model
        pc = H.VolPlanConfig()
        pc.dtype = H.LT_F32 if dtype == torch.float32 else H.LT_BF16
        pc.num_layers, pc.style_caffe, pc.num_joints = m.backbone.num_layers, int(m.backbone.style == "caffe"), 17
        pc.B, pc.NV, pc.H, pc.W = c["B"], c["NV"], c["H"], c["H"]
        pc.volume_size, pc.cuboid_side, pc.volume_multiplier = m.volume_size, m.cuboid_side, m.volume_multiplier
        pc.volume_softmax, pc.aggregation = int(bool(m.volume_softmax)), H.AGG[m.volume_aggregation_method]
        pc.transfer_cmu_to_human36m, pc.use_graph = int(bool(m.get("transfer_cmu_to_human36m", False))), int(use_graph)
        self.kind = m.kind
        keep = []
        arr = (H.NamedTensor * len(sd))()
        for i, (k, v) in enumerate(sd.items()):
            t = v.detach().float().contiguous()
            keep.append(t)
            arr[i].name, arr[i].data, arr[i].ndim = k.encode(), t.data_ptr(), max(1, t.dim())
            for j, n in enumerate(t.shape if t.dim() else (1,)):
                arr[i].shape[j] = n
        self.plan = C.c_void_p()
        H.check(lib.lt_plan_create_vol(C.byref(pc), arr, len(sd), C.byref(self.plan)), "lt_plan_create_vol")
        del keep, arr                      # the host arrays are read during create only
        self.pc = pc
        self.info = H.PlanInfo()
        H.check(lib.lt_plan_info(self.plan, C.byref(self.info)), "lt_plan_info")

    def forward(self, thetas=None):
        c, inp, pc, lib = self.c, self.inp, self.pc, H.lib()
        B, NV, V, J = pc.B, pc.NV, pc.volume_size, pc.num_joints
        h, w = self.info.heatmap_h, self.info.heatmap_w
        K = np.ascontiguousarray(np.broadcast_to(inp["K"][None], (B, NV, 3, 3)), dtype=np.float64)
        R = np.ascontiguousarray(np.broadcast_to(inp["R"][None], (B, NV, 3, 3)), dtype=np.float64)
        t = np.ascontiguousarray(np.broadcast_to(inp["t"].reshape(NV, 3)[None], (B, NV, 3)), dtype=np.float64)
        k3 = np.asarray(inp["pred_keypoints_3d"], dtype=np.float64)
        base = np.ascontiguousarray((k3[:, 11, :3] + k3[:, 12, :3]) / 2 if self.kind == "coco" else k3[:, 6, :3])          # triangulation.py:284-296
        rot = None
        if thetas is not None:
            axis = [0, 1, 0] if self.kind == "coco" else [0, 0, 1]
            rot = np.ascontiguousarray(np.stack([_rotation(axis, th) for th in thetas]).reshape(B, 9))
        images = inp["images"].to(DEV).contiguous()
        out = {"kp": torch.empty(B, J, 3, device=DEV), "vols": torch.empty(B, J, V, V, V, device=DEV), "feats": torch.empty(B, NV, 32, h, w, device=DEV),
               "coords": torch.empty(B, V, V, V, 3, device=DEV)}
        conf = torch.empty(B, NV, 32, device=DEV) if pc.aggregation in (H.AGG["conf"], H.AGG["conf_norm"]) else None
        st = torch.cuda.current_stream().cuda_stream
        dp = lambda a: a.ctypes.data_as(C.c_void_p)
        H.check(lib.lt_plan_forward_vol(self.plan, images.data_ptr(), dp(K), dp(R), dp(t), dp(base), None if rot is None else dp(rot), out["kp"].data_ptr(),
                                        out["vols"].data_ptr(), out["feats"].data_ptr(), out["coords"].data_ptr(), None if conf is None else conf.data_ptr(), st),
                "lt_plan_forward_vol")
        torch.cuda.synchronize()
        H.check(lib.lt_plan_info(self.plan, C.byref(self.info)), "lt_plan_info")
        n = B * J * V ** 3
        lg = torch.empty(n, device=DEV)
        # device -> device copy out of the plan's buffer through the library itself (fp32 -> fp32 "conversion" of n / 32 rows of 32)
        H.check(lib.lt_convert_pad(H.LT_F32, self.info.logits, H.LT_F32, lg.data_ptr(), n // 32, 32, 32, st), "lt_convert_pad")
        torch.cuda.synchronize()
        out["logits"] = lg.reshape(B, J, V, V, V) if self.info.logits_planar else lg.reshape(B, V, V, V, J).permute(0, 4, 1, 2, 3)
        out["conf"] = conf
        return out

    def close(self):
        if self.plan:
            H.lib().lt_plan_destroy(self.plan)
            self.plan = None


@pytest.mark.parametrize("tag", ["c2_sharp", "small_softmax", "small_sum_coco", "small_conf", "small_max", "small_relu_conf", "c2_b4"])
def test_plan_abi_fp32_vs_reference_golden(golden_dir, tag):
    """The fp32 plan built and run through lt_plan_* alone against the reference's stored outputs: every backbone depth of the fixtures (18 / 50 / 152), the
    four aggregation methods with the confidence head, kind 'coco' with the CMU transfer, rotated cuboids, a camera inside the cuboid, ReLU volumes."""
    g = np.load(os.path.join(golden_dir, "vol_%s.npz" % tag))
    P = CPlan(tag, torch.float32)
    try:
        c = P.c
        o = P.forward(g["thetas"] if c["rotate"] else None)
        s = int(g["stride"])
        B, NV = c["B"], c["NV"]
        check("plan-abi " + tag + "/coord_volumes", o["coords"].cpu()[:, ::s, ::s, ::s], g["cv_sub"], 2e-7)
        check("plan-abi " + tag + "/features", _sub(o["feats"].cpu().reshape(B * NV, *o["feats"].shape[2:]), s), g["feat_sub"], 2e-5)
        check("plan-abi " + tag + "/v2v logits", _sub(o["logits"].float().cpu(), s), g["logits_sub"], 2e-5)
        check("plan-abi " + tag + "/volumes", _sub(o["vols"].cpu(), s), g["vol_sub"], 1e-4)
        if o["conf"] is not None:
            conf = o["conf"].cpu()
            if c["method"] == "conf_norm":
                conf = conf / conf.sum(dim=1, keepdim=True)          # the reference RETURNS the normalised confidences (triangulation.py:268-269)
            check("plan-abi " + tag + "/vol_confidences", conf, g["vol_conf"], 1e-4)
        kp = o["kp"].cpu().double().numpy()
        self_rel = float(g["ref_self_rel"])
        if c["volume_softmax"]:
            rel64 = np.abs(kp - g["kp_fp64"]) / np.maximum(np.abs(g["kp_fp64"]), 1.0)
            rel = np.abs(kp - g["kp"]) / np.maximum(np.abs(g["kp"]), 1.0)
        else:                                  # unnormalised ReLU volumes: norm-wise, see tests/test_gpu_models.py
            scale = np.abs(g["kp_fp64"]).max(axis=(1, 2), keepdims=True)
            rel64, rel = np.abs(kp - g["kp_fp64"]) / scale, np.abs(kp - g["kp"]) / scale
        record("plan-abi " + tag + "/joints fp32: max rel vs the exact soft-argmax of the reference's logits", float(rel64.max()))
        order_rel = float(g["ref_order_rel"]) if "ref_order_rel" in g.files else 0.0          # the reference's own fp32 order noise where the fixture stores it (tests/test_gpu_models.py)
        assert rel64.max() <= 1e-4 + order_rel and rel.max() <= 1e-4 + self_rel + order_rel, (float(rel64.max()), float(rel.max()), self_rel, order_rel)
        o2 = P.forward(g["thetas"] if c["rotate"] else None)          # the second call replays the captured graph
        assert P.info.graph_captured == 1 and torch.equal(o2["kp"], o["kp"]) and torch.equal(o2["vols"], o["vols"])
        assert P.info.n_expand_reduce == 0 and P.info.n_bottleneck == 0 and P.info.n_stem_pool == 0          # fp32 plans record no bf16-only fusion
    finally:
        P.close()


def test_plan_abi_bf16_benchmark_kernel_set_vs_reference_golden(golden_dir):
    """The bf16 plan of the benchmark shape at B = 8 through lt_plan_* alone: it records the kernel set of the timed forward (asserted from lt_plan_info) and
    meets the per-fixture gates of the Python path (tests/test_gpu_models.py BF16_GATES, 1.5 x the measured deviation from the REFERENCE's fp32 outputs)."""
    from test_gpu_models import BF16_GATES
    tag = "c2_b8_sharp"
    g = np.load(os.path.join(golden_dir, "vol_%s.npz" % tag))
    P = CPlan(tag, torch.bfloat16)
    try:
        c = P.c
        o = P.forward()
        i = P.info
        kernels = {"xr": i.n_expand_reduce, "bneck": i.n_bottleneck, "bneck_ds": i.n_bottleneck_ds, "cat2": i.n_conv_cat2, "conv2d_halo": i.n_conv2d_halo, "pwchain": i.n_pwchain,
                   "stem": i.n_stem_pool, "splitk": i.n_splitk, "conv_skip": i.n_conv_skip}
        record("plan-abi c2_b8_sharp/bf16 plan kernels", kernels)
        assert kernels["xr"] == 34 and kernels["bneck"] == 9 and kernels["bneck_ds"] == 1 and kernels["cat2"] >= 2 and kernels["conv2d_halo"] == 37, kernels
        assert kernels["pwchain"] == 1 and kernels["stem"] == 1 and kernels["splitk"] == 18 and kernels["conv_skip"] == 1, kernels
        s = int(g["stride"])
        B, NV = c["B"], c["NV"]
        d = o["kp"].cpu().numpy() - g["kp"]
        got = (float(np.sqrt((d ** 2).sum(-1)).mean()), float(np.abs(d).max()), rel_err(_sub(o["feats"].cpu().reshape(B * NV, *o["feats"].shape[2:]), s), g["feat_sub"]),
               rel_err(_sub(o["logits"].float().cpu(), s), g["logits_sub"]), rel_err(_sub(o["vols"].cpu(), s), g["vol_sub"]))
        record("plan-abi c2_b8_sharp/bf16: MPJPE mm, max abs mm, features, logits, volumes vs reference", list(got))
        for what, v, gate in zip(("MPJPE mm", "max abs mm", "features", "logits", "volumes"), got, BF16_GATES[tag]):
            assert gate is None or v <= gate, "plan-abi %s: %s = %.4g > gate %.4g" % (tag, what, v, gate)
        check("plan-abi c2_b8_sharp/coord_volumes", o["coords"].cpu()[:, ::s, ::s, ::s], g["cv_sub"], 2e-7)
    finally:
        P.close()


@pytest.mark.parametrize("tag,dtype", [("c2_b8_sharp", torch.bfloat16), ("c2_b4", torch.bfloat16), ("small_conf", torch.float32), ("c2_sharp", torch.float32)])
def test_plan_abi_equals_the_python_hosts_plan_bit_for_bit(tag, dtype):
    """The C plan and the plan lt_engine.py / mvn/models record for the same model are the SAME launches over the same packed weights: identical outputs, bit for
    bit (joints, volumes, features, coordinates) -- the selection rules, thresholds, BatchNorm fold and packing orders of the two hosts agree."""
    from mvn.models.triangulation import VolumetricTriangulationNet
    from mvn.utils.multiview import Camera
    P = CPlan(tag, dtype)
    try:
        o = P.forward()
        cfg, sd, inp, c = build_vol_case(tag)
        m = VolumetricTriangulationNet(cfg, device=DEV)
        m.load_state_dict(sd, strict=True)
        m.eval()
        m.compute_dtype = dtype
        cams = [[Camera(inp["R"][v], inp["t"][v], inp["K"][v]) for _ in range(c["B"])] for v in range(c["NV"])]
        kp, feats, vols, conf, _, cvs, _ = m(inp["images"].to(DEV), None, {"cameras": cams, "pred_keypoints_3d": inp["pred_keypoints_3d"]})
        torch.cuda.synchronize()
        for name, a, b in (("coord_volumes", o["coords"], cvs), ("features", o["feats"], feats), ("volumes", o["vols"], vols), ("joints", o["kp"], kp)):
            assert torch.equal(a, b), "%s %s: C plan != Python plan (max |d| %.3e)" % (tag, name, float((a - b).abs().max()))
        npy = len(list(m._plans.values())[0]["plan"].ops)
        record("plan-abi %s: launches of the C plan | of the Python plan" % tag, [P.info.launches, npy])
        assert P.info.launches == npy + (1 if dtype == torch.float32 else 0)          # the fp32 C plan counts its image layout pass (the Python host launches it outside its plan)
    finally:
        P.close()


def test_plan_abi_runs_in_a_process_that_never_imports_the_python_host(golden_dir):
    """The proof that the plan level needs nothing of the Python host: a FRESH interpreter builds the c2_sharp plan (ResNet-152, 4 x 384^2, 64^3) through
    lt_plan_* and checks its joints against the reference's stored output at the fp32 gate -- and at the end neither ``lt_engine`` nor any ``mvn`` module has
    been imported into that process (this test session itself imports them for other tests)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, os
import numpy as np, torch
root = %r
for p in (root, os.path.join(root, "learnable-triangulation-pytorch_amd"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import test_gpu_plan_abi as T
g = np.load(os.path.join(%r, "vol_c2_sharp.npz"))
P = T.CPlan("c2_sharp", torch.float32)
o = P.forward()
kp = o["kp"].cpu().double().numpy()
rel = float((np.abs(kp - g["kp_fp64"]) / np.maximum(np.abs(g["kp_fp64"]), 1.0)).max())
P.close()
host = sorted(m for m in sys.modules if m == "lt_engine" or m == "lt_train" or m == "mvn" or m.startswith("mvn."))
print("REL %%.3e HOST %%s" %% (rel, host))
assert rel <= 1e-4 and not host
''' % (root, golden_dir)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("REL")][0]
    record("plan-abi c2_sharp in a process without lt_engine / mvn: joints max rel vs the exact value | host modules imported", line)
