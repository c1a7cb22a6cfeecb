"""CPU: host-side logic of the product package -- weight packing / BN folding / transposed-conv phases
(through the ConvSpec interpreter in tests/emul.py), config + camera helpers against the golden
fixtures, state_dict parity with the reference's key set, loud failure without a GPU, and the C ABI
(library loads, exports every symbol include/lt_hip.h declares; no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import lt_engine as E
import lt_hip as H
from emul import emulate_conv
from oracle import spec as ospec
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cl(x):  # N,C,(D),H,W -> N,D,H,W,C
    if x.dim() == 4:
        x = x.unsqueeze(2)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _ncdhw(y, nd):
    y = y.permute(0, 4, 1, 2, 3)
    return y[:, :, 0] if nd == 2 else y


def _rand_bn(c, g):
    return (0.5 + torch.rand(c, generator=g), torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, 0.5 + torch.rand(c, generator=g))


def _bn_ref(x, bn):
    return F.batch_norm(x, bn[2], bn[3], bn[0], bn[1], False, 0.1, 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["stem7x7s2", "3x3s1", "3x3s2", "1x1s2", "1x1"])
def test_conv2d_spec(case, dtype):
    g = torch.Generator().manual_seed(1)
    cin, cout, k, s, p, hw = {"stem7x7s2": (3, 64, 7, 2, 3, 21), "3x3s1": (16, 24, 3, 1, 1, 9), "3x3s2": (8, 130, 3, 2, 1, 10),
                              "1x1s2": (32, 20, 1, 2, 0, 9), "1x1": (64, 17, 1, 1, 0, 5)}[case]
    x = torch.randn(2, cin, hw, hw + 1, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.2
    bn = _rand_bn(cout, g)
    cpad = max(cin, E.min_cin_of(dtype)) if cin == 3 else cin
    xcl = _cl(x)
    if cpad > cin:
        xcl = torch.cat([xcl, torch.zeros(*xcl.shape[:-1], cpad - cin)], dim=-1)
    ref = _bn_ref(F.conv2d(x, w, None, s, p), bn)
    res = torch.randn(ref.shape, generator=g)
    sp = E.make_conv_spec(w, None, bn, tuple(xcl.shape), s, p, dtype, flags=H.EPI_RELU_POST)
    assert sp.k_pad % E.k_step_of(dtype) == 0 and sp.cout_pad % 16 == 0 and sp.cout_pad >= cout
    out = emulate_conv(sp, xcl, _cl(res))
    assert torch.allclose(_ncdhw(out, 2), torch.relu(ref + res), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("k", [7, 3, 1])
def test_conv3d_spec(k):
    g = torch.Generator().manual_seed(2)
    cin, cout = (32, 16) if k == 7 else (16, 32)
    x = torch.randn(1, cin, 6, 5, 7, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    bn = _rand_bn(cout, g)
    ref = torch.relu(_bn_ref(F.conv3d(x, w, bias, 1, (k - 1) // 2), bn))
    sp = E.make_conv_spec(w, bias, bn, tuple(_cl(x).shape), 1, (k - 1) // 2, torch.float32, flags=H.EPI_RELU_POST)
    assert torch.allclose(_ncdhw(emulate_conv(sp, _cl(x)), 3), ref, atol=3e-5, rtol=1e-5)


def test_deconv2d_spec():
    """ConvTranspose2d 4x4 s2 p1 (pose_resnet deconv_layers) = 4 parity phases x 4 taps."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 5, 6, generator=g)
    w = torch.randn(16, 24, 4, 4, generator=g) * 0.1
    bn = _rand_bn(24, g)
    ref = torch.relu(_bn_ref(F.conv_transpose2d(x, w, None, 2, 1), bn))
    sp = E.make_conv_spec(w, None, bn, tuple(_cl(x).shape), 2, 1, torch.float32, transposed=True, flags=H.EPI_RELU_POST)
    assert len(sp.phases) == 4 and all(p.taps.shape[0] == 4 for p in sp.phases)
    assert torch.allclose(_ncdhw(emulate_conv(sp, _cl(x)), 2), ref, atol=2e-5, rtol=1e-5)


def test_deconv3d_spec():
    """ConvTranspose3d 2^3 s2 (Upsample3DBlock) = 8 parity phases x 1 tap, ReLU BEFORE the skip add."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 16, 3, 4, 2, generator=g)
    w = torch.randn(16, 8, 2, 2, 2, generator=g) * 0.2
    bias = torch.randn(8, generator=g)
    bn = _rand_bn(8, g)
    skip = torch.randn(1, 8, 6, 8, 4, generator=g)
    ref = torch.relu(_bn_ref(F.conv_transpose3d(x, w, bias, 2), bn)) + skip
    sp = E.make_conv_spec(w, bias, bn, tuple(_cl(x).shape), 2, 0, torch.float32, transposed=True, flags=H.EPI_RELU_PRE)
    assert len(sp.phases) == 8 and all(p.taps.shape[0] == 1 for p in sp.phases)
    assert torch.allclose(_ncdhw(emulate_conv(sp, _cl(x), _cl(skip)), 3), ref, atol=2e-5, rtol=1e-5)


def test_linear_as_conv_sigmoid():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 256, generator=g)
    w = torch.randn(17, 256, generator=g) * 0.1
    b = torch.randn(17, generator=g)
    sp = E.make_conv_spec(w[:, :, None, None], b, None, (1, 1, 1, 6, 256), 1, 0, torch.float32, flags=H.EPI_SIGMOID)
    out = emulate_conv(sp, x.reshape(1, 1, 1, 6, 256))
    assert torch.allclose(out.reshape(6, 17), torch.sigmoid(F.linear(x, w, b)), atol=1e-5)


def test_state_dict_keys_match_reference_spec():
    from mvn.models.triangulation import AlgebraicTriangulationNet, VolumetricTriangulationNet
    for nl, method in ((152, "softmax"), (50, "conf_norm"), (18, "sum")):
        m = VolumetricTriangulationNet(synth.vol_config(nl, 32, method), device="cpu")
        sp = ospec.vol_net_spec(nl, 17, method.startswith("conf"))
        sd = m.state_dict()
        assert list(sd) == list(sp) and all(tuple(sd[k].shape) == sp[k][0] for k in sp)
        assert not m.backbone.final_layer.weight.requires_grad
        m.load_state_dict(synth.make_state_dict(sp, seed=1), strict=True)
    m = AlgebraicTriangulationNet(synth.alg_config(50, True), device="cpu")
    assert list(m.state_dict()) == list(ospec.alg_net_spec(50, 17, True))
    cfg = synth.vol_config(50, 32, "conf")
    VolumetricTriangulationNet(cfg, device="cpu")
    assert cfg.model.backbone.vol_confidences is True and cfg.model.backbone.alg_confidences is False  # ctor mutates config
    with pytest.raises(ValueError):
        VolumetricTriangulationNet(synth.vol_config(18, 32, "bogus"), device="cpu")


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU -- it never routes through torch CPU ops or the oracle."""
    from mvn.models.triangulation import VolumetricTriangulationNet
    from mvn.utils import op
    m = VolumetricTriangulationNet(synth.vol_config(18, 32), device="cpu").eval()
    inp = synth.make_inputs(1, 2, 64)
    with pytest.raises(RuntimeError, match="GPU"):
        m(inp["images"], None, {"cameras": [], "pred_keypoints_3d": inp["pred_keypoints_3d"]})
    with pytest.raises(RuntimeError, match="GPU"):
        op.unproject_heatmaps(torch.zeros(1, 2, 4, 8, 8), torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 2, 2, 3), "sum")
    with pytest.raises(ValueError, match="Unknown volume_aggregation_method: bogus"):
        op.unproject_heatmaps(torch.zeros(1, 2, 4, 8, 8), torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 2, 2, 3), "bogus")
    with pytest.raises(RuntimeError, match="GPU"):
        op.integrate_tensor_3d_with_coordinates(torch.zeros(1, 2, 2, 2, 2), torch.zeros(1, 2, 2, 2, 3))
    for mod in ("lt_hip", "lt_engine", "mvn.models.triangulation", "mvn.utils.op"):
        import importlib
        src = open(importlib.import_module(mod).__file__).read()
        assert "oracle" not in src, mod + " must not touch oracle/"


def test_geometry_helpers(golden_dir):
    from mvn.utils import multiview, volumetric
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    for i in range(4):
        a = g["rot_%d_arg" % i]
        assert np.abs(volumetric.get_rotation_matrix(a[:3], a[3]) - g["rot_%d" % i]).max() < 1e-15
    ci = g["cam_in"]
    cam = multiview.Camera(ci[:9].reshape(3, 3), ci[9:12], ci[12:].reshape(3, 3))
    cam.update_after_crop((10, 20, 200, 220))
    cam.update_after_resize((200, 190), (96, 96))
    assert np.abs(cam.K - g["cam_K_after"]).max() < 1e-12 and np.abs(cam.projection - g["cam_P_after"]).max() < 1e-9
    K, R, t = synth.ring_cameras(3, 128)
    cams = [[multiview.Camera(R[v], t[v], K[v]) for _ in range(2)] for v in range(3)]
    Ks, Rs, ts = multiview.stack_cameras(cams)
    P = multiview.resized_projections(Ks, Rs, ts, (128, 128), (32, 32))
    for v in range(3):
        c = multiview.Camera(R[v], t[v], K[v]); c.update_after_resize((128, 128), (32, 32))
        assert np.abs(P[1, v] - c.projection).max() < 1e-9
    with pytest.raises(TypeError):
        multiview.euclidean_to_homogeneous([1, 2, 3])
    X = np.array([[1.0, 2.0, 3.0]])
    assert np.allclose(multiview.project_3d_points_to_image_plane_without_distortion(P[0, 0], X),
                       (np.append(X[0], 1) @ P[0, 0].T)[:2] / (np.append(X[0], 1) @ P[0, 0].T)[2])


def test_config_surface(tmp_path):
    from mvn.utils import cfg
    p = tmp_path / "c.yaml"
    p.write_text("title: t\nimage_shape: [384, 384]\nmodel:\n  name: vol\n  backbone:\n    num_layers: 152\n    num_joints: 17\n")
    c = cfg.load_config(str(p))
    assert c.model.backbone.num_layers == 152 and c.image_shape == [384, 384] and c["model"]["name"] == "vol"
    c.model.backbone.alg_confidences = False
    assert c.model.backbone.alg_confidences is False and not hasattr(c.model, "transfer_cmu_to_human36m")


def test_c_abi_exports_every_declared_symbol():
    """liblt_hip.so loads without a GPU and exports exactly the functions include/lt_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "lt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(H.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
    assert declared == set(H.SIGNATURES), (declared ^ set(H.SIGNATURES))
    l = H.lib()
    assert l.lt_abi_version() == 1
    assert [l.lt_conv_cout_pad(c) for c in (1, 16, 17, 32, 33, 64, 65, 128, 129, 2048)] == [E.cout_pad_of(c) for c in (1, 16, 17, 32, 33, 64, 65, 128, 129, 2048)]
    # argument validation happens before any device work: error codes + messages without a GPU
    d = H.ConvDesc()
    assert l.lt_conv_fwd(ctypes.byref(d), None, None, None, None, None, None, None) == -1
    assert b"null" in l.lt_last_error()
    assert l.lt_unproject_fwd(0, 1, 1, 1, None, 1, 1, 1, 4, 8, 8, 2, 2, 2, 9, None) == -1 and b"aggregation" in l.lt_last_error()
    assert l.lt_softargmax3d_fwd(1, 1, 1.0, 1, 1, 40, 1, None, 1, 40, 8, 1, None) == -2 and b"J=40" in l.lt_last_error()
    assert l.lt_softargmax3d_workspace(2, 17, 64 ** 3) == (2 * 17 * 128 * 5 + 2 * 17 * 2) * 4
    # lt_pwchain_fwd: bf16 only, 32 input channels, rows % 64 == 0, inner widths 32, exactly the last layer stores fp32
    pd = H.PwChainDesc()
    pd.dtype, pd.nlayers, pd.rows, pd.cin, pd.ldy = 0, 1, 64, 32, 17
    assert l.lt_pwchain_fwd(ctypes.byref(pd), 1, 1, None) == -2 and b"bf16" in l.lt_last_error()
    pd.dtype, pd.rows = 1, 100
    assert l.lt_pwchain_fwd(ctypes.byref(pd), 1, 1, None) == -2 and b"multiple of 64" in l.lt_last_error()
    pd.rows, pd.nlayers = 64, 4
    assert l.lt_pwchain_fwd(ctypes.byref(pd), 1, 1, None) == -1 and b"nlayers" in l.lt_last_error()
    pd.nlayers = 2
    pd.cout[0], pd.cout[1], pd.k_pad[0], pd.k_pad[1], pd.weight[0], pd.weight[1] = 16, 17, 64, 64, 1, 1
    assert l.lt_pwchain_fwd(ctypes.byref(pd), 1, 1, None) == -2 and b"inner layers" in l.lt_last_error()
    pd.cout[0] = 32
    assert l.lt_pwchain_fwd(ctypes.byref(pd), 1, 1, None) == -2 and b"last layer stores fp32" in l.lt_last_error()
    # lt_stem_pool_fwd: bf16, 8 -> 64 channels, weights packed by lt_stem_pack_weights from [64][k_pad >= 392]
    sd = H.StemDesc()
    sd.dtype, sd.N, sd.H, sd.W, sd.Cin, sd.Cout = 0, 1, 32, 32, 8, 64
    assert l.lt_stem_pool_fwd(ctypes.byref(sd), 1, 1, None) == -2 and b"bf16" in l.lt_last_error()
    sd.dtype, sd.Cin = 1, 3
    assert l.lt_stem_pool_fwd(ctypes.byref(sd), 1, 1, None) == -2 and b"channels" in l.lt_last_error()
    sd.Cin, sd.weight = 8, 8
    assert l.lt_stem_pool_fwd(ctypes.byref(sd), 1, 1, None) == -1 and b"packed weights" in l.lt_last_error()
    sd.x_layout = 1       # fp32 (N, 3, H, W) images: Cin must then be 3
    assert l.lt_stem_pool_fwd(ctypes.byref(sd), 1, 1, None) == -2 and b"channels" in l.lt_last_error()
    sd.x_layout = 2
    assert l.lt_stem_pool_fwd(ctypes.byref(sd), 1, 1, None) == -1 and b"x_layout" in l.lt_last_error()
    assert l.lt_stem_packed_bytes() == 2 * 28 * 64 * 16
    assert l.lt_stem_pack_weights(1, 64, 1, None) == -1 and b"k_pad" in l.lt_last_error()
    assert l.lt_conv_pack_weights(1, 100, 64, 1, None) == -1 and b"cout_pad" in l.lt_last_error()
    assert l.lt_conv_pack_weights_t32(1, 64, 1728, 64, 28, 1, None) == -1 and b"ntaps" in l.lt_last_error()
    # planar output: voxels per sample must be a multiple of 64 that divides rows
    pd.flags[1], pd.rows, pd.plane = H.EPI_STORE_F32, 128, 96
    assert l.lt_pwchain_fwd(ctypes.byref(pd), 1, 1, None) == -1 and b"plane" in l.lt_last_error()


REF_EXPERIMENTS = "/root/reference/experiments/human36m"


@pytest.mark.skipif(not os.path.isdir(REF_EXPERIMENTS), reason="the reference checkout is only present in the build container")
def test_reference_experiment_yamls_build_the_models():
    """north_star: "keeps ... the experiments/*.yaml config surface".  Every vol / alg experiment file of the reference (train and eval), read with
    mvn.utils.cfg.load_config exactly as train.py:397 does, constructs the model class train.py:400-404 picks -- with the keys the constructors read
    (volume_aggregation_method, volume_softmax, volume_multiplier, volume_size, cuboid_side, kind, use_gt_pelvis, use_confidences, heatmap_*,
    backbone.*) -- and the state dict has the reference layout (oracle/spec.py, pinned against the reference's own modules by make_golden).  Only
    ``init_weights`` is switched off (the checkpoints are not part of the repository); the ransac file is SURVEY's out-of-scope CPU baseline."""
    import glob
    from mvn.utils import cfg
    from mvn.models.triangulation import AlgebraicTriangulationNet, VolumetricTriangulationNet
    from oracle import spec
    files = sorted(glob.glob(os.path.join(REF_EXPERIMENTS, "*", "*.yaml")))
    assert len(files) >= 5
    seen = set()
    for f in files:
        c = cfg.load_config(f)
        name = c.model.name
        if name == "ransac":
            continue
        c.model.init_weights = False
        c.model.backbone.init_weights = False
        nl, nj = c.model.backbone.num_layers, c.model.backbone.num_joints
        if name == "vol":
            m = VolumetricTriangulationNet(c, device="cpu")
            assert m.volume_aggregation_method == c.model.volume_aggregation_method and m.volume_size == c.model.volume_size
            assert m.volume_softmax == c.model.volume_softmax and m.volume_multiplier == c.model.volume_multiplier
            assert m.cuboid_side == c.model.cuboid_side and m.kind == c.model.kind and m.use_gt_pelvis == c.model.use_gt_pelvis
            want = spec.vol_net_spec(nl, nj, c.model.volume_aggregation_method.startswith("conf"))
        else:
            assert name == "alg"
            m = AlgebraicTriangulationNet(c, device="cpu")
            assert m.use_confidences == c.model.use_confidences and m.heatmap_softmax == c.model.heatmap_softmax
            assert m.heatmap_multiplier == c.model.heatmap_multiplier
            want = spec.alg_net_spec(nl, nj, c.model.use_confidences)
        sd = m.state_dict()
        assert list(sd.keys()) == list(want.keys()), os.path.basename(f)
        assert all(tuple(sd[k].shape) == tuple(want[k][0]) for k in want), os.path.basename(f)
        assert c.opt.scale_keypoints_3d == 0.1 and c.image_shape == [384, 384]
        seen.add((name, os.path.basename(os.path.dirname(f))))
    assert {("vol", "train"), ("vol", "eval"), ("alg", "train"), ("alg", "eval")} <= seen
