"""GPU (-m gpu): the non-convolution backward kernels and training-mode BatchNorm statistics
against gradients computed by torch.autograd THROUGH THE REFERENCE ITSELF (tests/golden/grads.npz, oracle/make_golden.py
``grad``), through the product's autograd-facing API (mvn.utils.op / mvn.models.loss -> lt_unproject_bwd, lt_softargmax3d_bwd,
lt_volumetric_ce_fwd, lt_bn_stats_fwd).  Gate: max|d| <= 1e-4 * max|ref| (fp32 kernels)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import check, record
from oracle import spec, synth
from oracle import vol_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def grads(golden_dir):
    return np.load(os.path.join(golden_dir, "grads.npz"))


@pytest.fixture(scope="module")
def ops(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


@pytest.mark.parametrize("method", ["sum", "max", "softmax", "conf"])
def test_unproject_backward_vs_reference_autograd(ops, grads, method):
    from mvn.utils import op
    hm = torch.from_numpy(ops["unproj_hm_C8"]).to(DEV).requires_grad_(True)
    conf = torch.from_numpy(ops["unproj_cin_C8"]).to(DEV).requires_grad_(True)
    P, cv = torch.from_numpy(ops["unproj_P"]).to(DEV), torch.from_numpy(ops["unproj_cv"]).to(DEV)
    G = torch.from_numpy(grads["u_G"]).to(DEV)
    vol = op.unproject_heatmaps(hm, P, cv, method, conf)
    check("bwd/unproject %s forward (autograd path)" % method, vol.detach().cpu(), ops["unproj_%s_C8" % method], 1e-5)
    (vol * G).sum().backward()
    check("bwd/unproject %s d/d heatmaps" % method, hm.grad.cpu(), grads["u_ghm_" + method], 1e-4)
    if method == "conf":
        check("bwd/unproject conf d/d confidences", conf.grad.cpu(), grads["u_gconf"], 1e-4)
    else:
        assert conf.grad is None


def test_softargmax3d_backward_and_losses_vs_reference_autograd(ops, grads):
    from mvn.models import loss as L
    from mvn.utils import op
    cvs = torch.from_numpy(ops["int3d_cv"]).to(DEV)
    Gk, Gp = torch.from_numpy(grads["s_Gk"]).to(DEV), torch.from_numpy(grads["s_Gp"]).to(DEV)
    gt, val = torch.from_numpy(grads["s_gt"]).to(DEV), torch.from_numpy(grads["s_val"]).to(DEV)
    for sm in (True, False):     # dense gradients on both outputs (coordinates and the returned volumes)
        v = torch.from_numpy(ops["int3d_in"]).to(DEV).requires_grad_(True)
        c, pv = op.integrate_tensor_3d_with_coordinates(v, cvs, softmax=sm)
        ((c * Gk).sum() + (pv * Gp).sum()).backward()
        check("bwd/softargmax3d softmax=%d dense" % sm, v.grad.cpu(), grads["s_glogits_dense_%d" % sm], 1e-4)
    # the training loss of train.py:217-230: MAE on the scaled joints + 0.01 x VolumetricCELoss on the returned volumes
    v = torch.from_numpy(ops["int3d_in"]).to(DEV).requires_grad_(True)
    c, pv = op.integrate_tensor_3d_with_coordinates(v, cvs, softmax=True)
    mae = L.KeypointsMAELoss()(c * 0.1, gt * 0.1, val)
    ce = L.VolumetricCELoss()(cvs, pv, gt, val)
    assert float(mae) == pytest.approx(float(grads["s_mae"]), rel=1e-5) and float(ce) == pytest.approx(float(grads["s_ce"]), rel=1e-5)
    (mae + 0.01 * ce).backward()
    check("bwd/softargmax3d MAE + 0.01 CE (sparse CE gradient)", v.grad.cpu(), grads["s_glogits_loss"], 1e-4)
    for name, cls in (("mse", L.KeypointsMSELoss), ("mse_smooth", L.KeypointsMSESmoothLoss), ("l2", L.KeypointsL2Loss)):
        assert float(cls()(c.detach() * 0.1, gt * 0.1, val)) == pytest.approx(float(grads["s_" + name]), rel=1e-5), name
    # CE on volumes that do not come from our soft-argmax node: dense scatter fallback, same numbers
    pv2 = pv.detach().clone().requires_grad_(True)
    L.VolumetricCELoss()(cvs, pv2, gt, val).backward()
    assert int((pv2.grad != 0).sum()) <= pv2.shape[0] * pv2.shape[1]


def test_uniform_gradient_on_the_returned_volumes_is_a_real_gradient(ops):
    """``volumes.sum()`` / ``.mean()`` hand the soft-argmax node an EXPANDED (all strides zero) gradient -- the same shape of tensor as the
    placeholder VolumetricCELoss leaves next to its sparse gradient.  It must be treated as the dense gradient it is: with softmax=False (ReLU volumes,
    op.py:90-91) d sum(relu(v)) / dv = [v > 0] is not zero.  Reference: torch autograd over the reference's expressions (op.py:84-96) on the CPU."""
    from mvn.models import loss as L
    from mvn.utils import op
    cvs = torch.from_numpy(ops["int3d_cv"]).to(DEV)
    x = torch.from_numpy(ops["int3d_in"])
    for sm in (False, True):
        v = x.clone().to(DEV).requires_grad_(True)
        c, pv = op.integrate_tensor_3d_with_coordinates(v, cvs, softmax=sm)
        (c.sum() * 0.01 + pv.sum() * 0.5).backward()
        vr = x.clone().double().requires_grad_(True)
        flat = vr.reshape(*vr.shape[:2], -1)
        pr = (torch.softmax(flat, dim=2) if sm else torch.relu(flat)).reshape(vr.shape)
        cr = torch.einsum("bnxyz,bxyzc->bnc", pr, cvs.cpu().double())
        (cr.sum() * 0.01 + pr.sum() * 0.5).backward()
        assert sm or float(vr.grad.abs().max()) > 0.4          # the ReLU case really has a uniform 0.5 where v > 0
        check("bwd/softargmax3d softmax=%d uniform (zero-stride) gradient on the volumes" % sm, v.grad.cpu(), vr.grad, 1e-4)
    # and next to VolumetricCELoss's sparse gradient the placeholder is still recognised (nothing is added twice, nothing dropped)
    gt = torch.randn(x.shape[0], x.shape[1], 3) * 200
    val = torch.ones(x.shape[0], x.shape[1], 1)
    v = x.clone().to(DEV).requires_grad_(True)
    c, pv = op.integrate_tensor_3d_with_coordinates(v, cvs, softmax=True)
    L.VolumetricCELoss()(cvs, pv, gt.to(DEV), val.to(DEV)).backward()
    g_sparse = v.grad.clone()
    v2 = x.clone().to(DEV).requires_grad_(True)
    c2, pv2 = op.integrate_tensor_3d_with_coordinates(v2, cvs, softmax=True)
    pv3 = pv2.detach().clone().requires_grad_(True)          # the same loss through the dense-scatter fallback, fed back as a dense gradient
    L.VolumetricCELoss()(cvs, pv3, gt.to(DEV), val.to(DEV)).backward()
    (pv2 * pv3.grad).sum().backward()
    check("bwd/softargmax3d CE sparse (placeholder) vs the same gradient handed over dense", g_sparse.cpu(), v2.grad.cpu(), 1e-5)


def test_pipeline_shape_gradients_vs_reference_autograd(golden_dir, grads):
    """small_softmax whole-pipeline case (2 samples, 3 views, camera 0 inside the cube, rotated cuboids, 32 channels, 32^3 voxels):
    d loss / d features through lt_unproject_bwd for a random upstream gradient, and d (MAE + 0.01 CE) / d logits."""
    from mvn.models import loss as L
    from mvn.utils import op
    gs = np.load(os.path.join(golden_dir, "vol_small_softmax.npz"))
    cfg = synth.vol_config(18, 32, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(18, 17, False), seed=2, sharpen=True, basic_block=True)
    inp = synth.make_inputs(2, 3, 128, seed=2, inside=True)
    o = O.volumetric_forward(sd, cfg, inp["images"], inp["K"], inp["R"], inp["t"], inp["pred_keypoints_3d"], thetas=gs["thetas"], stages=True)
    g = torch.Generator().manual_seed(41)       # replay the generator of oracle/make_golden.py gen_grad up to GV
    torch.randn(2, 8, 7, 7, 7, generator=g); torch.randn(2, 5, 3, generator=g); torch.randn(2, 5, 6, 7, 8, generator=g)
    torch.randn(2, 5, 3, generator=g); torch.rand(2, 5, 1, generator=g)
    GV = torch.randn(o["unprojected"].shape, generator=g)
    f = o["features"].to(DEV).requires_grad_(True)
    vol = op.unproject_heatmaps(f, o["proj"].to(DEV), o["coord_volumes"].to(DEV), "softmax")
    (vol * GV.to(DEV)).sum().backward()
    check("bwd/pipeline d/d features (fp32)", f.grad.cpu(), grads["p_gfeat"], 1e-4)
    fb = o["features"].to(DEV).to(torch.bfloat16).requires_grad_(True)     # bf16 activations: recorded, not gated at 1e-4
    (op.unproject_heatmaps(fb, o["proj"].to(DEV), o["coord_volumes"].to(DEV), "softmax").float() * GV.to(DEV)).sum().backward()
    record("bwd/pipeline d/d features, bf16 feature maps (max|d|/max|ref|)",
           float((fb.grad.float().cpu() - torch.from_numpy(grads["p_gfeat"])).abs().max() / np.abs(grads["p_gfeat"]).max()))
    lg = o["logits"].to(DEV).requires_grad_(True)
    cvd = o["coord_volumes"].to(DEV)
    kp, pv = op.integrate_tensor_3d_with_coordinates(lg * 1.0, cvd, softmax=True)
    gt3, val3 = torch.from_numpy(grads["p_gt"]).to(DEV), torch.from_numpy(grads["p_val"]).to(DEV)
    mae = L.KeypointsMAELoss()(kp * 0.1, gt3 * 0.1, val3)
    ce = L.VolumetricCELoss()(cvd, pv, gt3, val3)
    assert float(mae) == pytest.approx(float(grads["p_mae"]), rel=1e-4) and float(ce) == pytest.approx(float(grads["p_ce"]), rel=1e-4)
    (mae + 0.01 * ce).backward()
    e = float((lg.grad.cpu()[:, :, ::2, ::2, ::2] - torch.from_numpy(grads["p_glogits_s2"])).abs().max() / float(grads["p_glogits_absmax"]))
    record("bwd/pipeline d/d logits (max|d|/max|ref|)", e)
    assert e <= 1e-4, e


@pytest.mark.parametrize("shape,dtype", [((6, 17, 9, 11), torch.float32), ((3, 64, 24, 24), torch.bfloat16), ((2, 32, 8, 8, 8), torch.float32),
                                         ((2, 300, 5, 7), torch.float32)])
def test_batchnorm_batch_statistics_vs_torch(shape, dtype):
    """Training-mode BatchNorm: per-channel batch mean / biased variance and the running-statistics update (momentum 0.1, unbiased
    variance) against F.batch_norm(training=True) on the CPU."""
    from mvn.utils import op
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(shape, generator=g) * 2 + 0.7).to(dtype)
    C = shape[1]
    rm, rv = torch.randn(C, generator=g) * 0.1, 0.5 + torch.rand(C, generator=g)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    xf = x.float()
    F.batch_norm(xf, rm_ref, rv_ref, None, None, True, 0.1, 1e-5)
    dims = [0] + list(range(2, xf.dim()))
    mean_ref, var_ref = xf.mean(dim=dims), xf.var(dim=dims, unbiased=False)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    mean, var = op.batchnorm_batch_stats(x.to(DEV), rmd, rvd, momentum=0.1)
    tol = 1e-5
    check("bn stats mean %s %s" % (shape, dtype), mean.cpu(), mean_ref, tol)
    check("bn stats var %s %s" % (shape, dtype), var.cpu(), var_ref, tol)
    check("bn running_mean %s" % (shape,), rmd.cpu(), rm_ref, tol)
    check("bn running_var %s" % (shape,), rvd.cpu(), rv_ref, tol)


def test_unproject_backward_is_bitwise_repeatable_and_matches_the_scatter(golden_dir):
    """Round 3: lt_unproject_bwd is a gather (one workgroup owns a 16 x 16 pixel tile of one view's gradient map and adds the taps that
    fall into it in a fixed order) -- two runs are BITWISE equal, where the round-2 scatter by global float atomics
    (LT_UNPROJ_BWD_ATOMICS=1, kept as the A/B reference) is only equal up to fp32 summation order.  Pipeline shape: 2 samples, 3 views,
    camera 0 inside the cube (depth <= 0 samples), rotated cuboids, 32 channels, 32^3 voxels, 32 x 32 maps."""
    from mvn.utils import op
    gs = np.load(os.path.join(golden_dir, "vol_small_softmax.npz"))
    cfg = synth.vol_config(18, 32, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(18, 17, False), seed=2, sharpen=True, basic_block=True)
    inp = synth.make_inputs(2, 3, 128, seed=2, inside=True)
    o = O.volumetric_forward(sd, cfg, inp["images"], inp["K"], inp["R"], inp["t"], inp["pred_keypoints_3d"], thetas=gs["thetas"], stages=True)
    GV = torch.randn(o["unprojected"].shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    P, cv = o["proj"].to(DEV), o["coord_volumes"].to(DEV)

    def grad(method, conf=None):
        f = o["features"].to(DEV).requires_grad_(True)
        c = None if conf is None else conf.clone().requires_grad_(True)
        (op.unproject_heatmaps(f, P, cv, method, c) * GV).sum().backward()
        return f.grad.clone(), (None if c is None else c.grad.clone())

    conf = torch.rand(2, 3, 32, generator=torch.Generator().manual_seed(6)).to(DEV) + 0.1
    for method in ("softmax", "sum", "max", "conf"):
        cf = conf if method == "conf" else None
        g1, c1 = grad(method, cf)
        g2, c2 = grad(method, cf)
        assert torch.equal(g1, g2), method + ": the gather must be bitwise repeatable"
        if c1 is not None:
            assert torch.equal(c1, c2)
        os.environ["LT_UNPROJ_BWD_ATOMICS"] = "1"
        try:
            gs_, cs_ = grad(method, cf)
        finally:
            del os.environ["LT_UNPROJ_BWD_ATOMICS"]
        check("bwd/unproject gather vs round-2 scatter, %s: d/d features" % method, g1.cpu(), gs_.cpu(), 2e-5)
        if c1 is not None:
            check("bwd/unproject gather vs round-2 scatter, conf: d/d confidences", c1.cpu(), cs_.cpu(), 2e-5)
    assert float(g1.abs().max()) > 0


@pytest.mark.parametrize("C,hw,vshape", [(4, (20, 37), (5, 6, 7)), (16, (33, 16), (8, 8, 8)), (64, (17, 40), (9, 4, 6)), (32, (96, 96), (12, 16, 20))])
def test_unproject_backward_gather_every_channel_width_and_ragged_shapes(C, hw, vshape):
    """Every instantiation of the gather (channels per wave 1 / 4 / 16 / 8: C = 64 is the one-hit-at-a-time form without half-waves), maps
    that are not multiples of the 16 x 16 tile, coordinate grids that are not multiples of the 4 x 4 x 4 brick, a camera inside the grid
    (depth <= 0 voxels): against the round-2 scatter (itself gated against the reference's autograd), all four aggregations with gradients,
    bitwise repeatable."""
    from mvn.utils import op
    g = torch.Generator().manual_seed(100 + C)
    B, NV = 2, 3
    h, w = hw
    K, R, t = synth.ring_cameras(NV, 4 * max(h, w))
    t = t.copy(); t[0] = t[0] * 0.05                       # camera 0 almost at the origin: inside the grid
    from mvn.utils import multiview
    proj = multiview.resized_projections(np.stack([K] * B, 0), np.stack([R] * B, 0), np.stack([t] * B, 0), (4 * h, 4 * w), (h, w))
    P = torch.from_numpy(np.asarray(proj, dtype=np.float32)).reshape(B, NV, 3, 4).to(DEV)
    ax = [torch.linspace(-900.0, 900.0, n) for n in vshape]
    cv = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1)[None].repeat(B, 1, 1, 1, 1)
    cv[1] += 37.0
    cv = cv.to(DEV)
    feats = torch.randn(B, NV, C, h, w, generator=g).to(DEV)
    conf = (torch.rand(B, NV, C, generator=g) + 0.1).to(DEV)
    G = torch.randn(B, C, *vshape, generator=g).to(DEV)

    def grad(method):
        f = feats.clone().requires_grad_(True)
        c = conf.clone().requires_grad_(True) if method == "conf" else None
        (op.unproject_heatmaps(f, P, cv, method, c) * G).sum().backward()
        return f.grad.clone(), (None if c is None else c.grad.clone())

    for method in ("softmax", "sum", "max", "conf"):
        g1, c1 = grad(method)
        g2, c2 = grad(method)
        assert torch.equal(g1, g2) and (c1 is None or torch.equal(c1, c2)), "%s C=%d: not bitwise repeatable" % (method, C)
        os.environ["LT_UNPROJ_BWD_ATOMICS"] = "1"
        try:
            gs_, cs_ = grad(method)
        finally:
            del os.environ["LT_UNPROJ_BWD_ATOMICS"]
        assert float(gs_.abs().max()) > 0
        check("bwd/unproject gather vs scatter C=%d %s %s: d/d features" % (C, hw, method), g1.cpu(), gs_.cpu(), 2e-5)
        if c1 is not None:
            check("bwd/unproject gather vs scatter C=%d %s: d/d confidences" % (C, hw), c1.cpu(), cs_.cpu(), 2e-5)


def test_softargmax2d_and_dlt_backward_vs_autograd_of_the_reference_ops():
    """The algebraic model's differentiable tail (train.py:189-236): d coordinates / d heatmaps of integrate_tensor_2d (op.py:11-47) and the
    gradient of triangulate_batch_of_points (multiview.py:141-183, torch.svd in the reference) with respect to the 2D points and the
    confidences -- against torch autograd through the oracle's restatement of both ops (the same torch calls as the reference, fp64)."""
    from mvn.utils import multiview, op
    g = torch.Generator().manual_seed(17)
    hm = torch.randn(6, 5, 24, 20, generator=g) * 2.0
    G = torch.randn(6, 5, 2, generator=g)
    h64 = hm.double().requires_grad_(True)
    c64, _ = O.integrate_tensor_2d(h64, True)
    (c64 * G.double()).sum().backward()
    hd = hm.to(DEV).requires_grad_(True)
    c, p = op.integrate_tensor_2d(hd, True)
    check("bwd/softargmax2d coordinates", c.detach().cpu(), c64.detach().float(), 1e-5)
    (c * G.to(DEV)).sum().backward()
    check("bwd/softargmax2d d/d heatmaps", hd.grad.cpu(), h64.grad.float(), 1e-4)
    h64r = hm.double().requires_grad_(True)                       # ReLU mode (heatmap_softmax: false)
    c64r, _ = O.integrate_tensor_2d(h64r, False)
    (c64r * G.double()).sum().backward()
    hdr = hm.to(DEV).requires_grad_(True)
    cr, _ = op.integrate_tensor_2d(hdr, False)
    check("bwd/softargmax2d relu mode coordinates", cr.detach().cpu(), c64r.detach().float(), 1e-5)
    (cr * G.to(DEV)).sum().backward()
    check("bwd/softargmax2d relu mode d/d heatmaps", hdr.grad.cpu(), h64r.grad.float(), 1e-4)
    # DLT: 2 samples, 4 ring cameras, 7 joints near the origin, noisy 2D observations, confidences in (0.2, 1.2)
    B, NV, J = 2, 4, 7
    K, R, t = synth.ring_cameras(NV, 256)
    P = torch.from_numpy(K @ np.concatenate([R, t], -1))[None].repeat(B, 1, 1, 1)                     # (B, NV, 3, 4) fp64
    X = torch.randn(B, J, 3, generator=g).double() * 300
    Xh = torch.cat([X, torch.ones(B, J, 1, dtype=torch.float64)], -1)
    proj = torch.einsum("bvik,bjk->bvji", P, Xh)
    pts = (proj[..., :2] / proj[..., 2:3] + torch.randn(B, NV, J, 2, generator=g).double() * 2.0)
    conf = torch.rand(B, NV, J, generator=g).double() + 0.2
    GX = torch.randn(B, J, 3, generator=g)
    p64, c64 = pts.clone().requires_grad_(True), conf.clone().requires_grad_(True)
    x64 = O.triangulate_batch_of_points(P, p64, c64)
    (x64 * GX.double()).sum().backward()
    pd, cd = pts.float().to(DEV).requires_grad_(True), conf.float().to(DEV).requires_grad_(True)
    x = multiview.triangulate_batch_of_points(P.float().to(DEV), pd, cd)
    check("bwd/dlt keypoints_3d", x.detach().cpu(), x64.detach().float(), 1e-4)
    (x * GX.to(DEV)).sum().backward()
    check("bwd/dlt d/d points", pd.grad.cpu(), p64.grad.float(), 2e-3)          # fp32 inputs of an inverse problem: the forward's own rows are fp32 (multiview.py:159-161)
    check("bwd/dlt d/d confidences", cd.grad.cpu(), c64.grad.float(), 2e-3)
