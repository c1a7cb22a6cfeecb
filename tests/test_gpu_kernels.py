"""GPU (-m gpu): every liblt_hip kernel, called through the C ABI, against a plain torch-CPU fp32 statement of
the same op (floating-point kernels) and against the committed reference outputs (tests/golden).

Tolerances (max|d| / max|ref|): fp32 kernels 2e-5 (accumulation-order noise over K <= 11k terms); bf16 kernels are
compared with the fp32 op evaluated on bf16-rounded operands, 1.5e-2 (output rounding 2^-8 + fp32 accumulation)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import lt_engine as E
import lt_hip as H
from gpu_util import bf16_round, check, from_cl, record, rel_err, to_cl
from oracle import synth
from oracle import vol_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TILES = {"auto": 0, "128x128": 1, "128x64": 2, "256x32": 3, "256x16": 4, "64x64": 5, "direct": 99,
         "v2_128x128": 11, "v2_128x64": 12, "v2_256x32": 13, "v2_256x16": 14, "v2_64x64": 15}


def _bn(c, g):
    return (0.5 + torch.rand(c, generator=g), torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, 0.5 + torch.rand(c, generator=g))


def _bn_ref(x, bn):
    return F.batch_norm(x, bn[2], bn[3], bn[0], bn[1], False, 0.1, 1e-5)


def run_conv(x, w, bias, bn, stride, pad, dtype, tile, transposed=False, relu=False, relu_pre=False, residual=None, cin_pad=None):
    nd = x.dim() - 2
    b = E.PlanBuilder(DEV, dtype, tile_override=tile)
    xa = E.Act(to_cl(x, cin_pad, dtype))
    ra = None if residual is None else E.Act(to_cl(residual, None, dtype))
    y = b.conv(xa, w, bias, bn, stride=stride, pad=pad, transposed=transposed, relu=relu, relu_pre=relu_pre, residual=ra)
    b.finish().run_eager(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return from_cl(y.t, nd)


CONV_CASES = {
    # name: (nd, N, cin, cout, k, stride, pad, spatial)
    "c2d_3x3_64_128": (2, 2, 64, 128, 3, 1, 1, (20, 24)),
    "c2d_1x1_256_64": (2, 2, 256, 64, 1, 1, 0, (17, 19)),
    "c2d_3x3s2_32_192": (2, 1, 32, 192, 3, 2, 1, (23, 21)),
    "c2d_1x1s2_64_256": (2, 2, 64, 256, 1, 2, 0, (16, 18)),
    "c3d_3x3_16_32": (3, 1, 16, 32, 3, 1, 1, (9, 10, 12)),
    "c3d_3x3_32_32": (3, 2, 32, 32, 3, 1, 1, (8, 8, 8)),
    "c3d_7x7_32_16": (3, 1, 32, 16, 7, 1, 3, (10, 9, 12)),
    "c3d_1x1_32_17": (3, 1, 32, 17, 1, 1, 0, (8, 9, 10)),
    "c3d_3x3_128_128_tiny": (3, 1, 128, 128, 3, 1, 1, (2, 2, 2)),
}


def _tiles_for(cout):
    cp = E.cout_pad_of(cout)
    if cp % 128 == 0:
        t = ["128x128", "128x64", "64x64", "256x32", "256x16"]
    elif cp == 64:
        t = ["128x64", "64x64", "256x32", "256x16"]
    elif cp == 32:
        t = ["256x32", "256x16"]
    else:
        t = ["256x16"]
    return ["auto", "direct"] + t + ["v2_" + n for n in t]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", list(CONV_CASES))
def test_conv_all_tiles(case, dtype):
    nd, N, cin, cout, k, s, p, sp = CONV_CASES[case]
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, cin, *sp, generator=g)
    w = torch.randn(cout, cin, *([k] * nd), generator=g) * (1.0 / (cin * k ** nd) ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    bn = _bn(cout, g)
    conv = F.conv2d if nd == 2 else F.conv3d
    if dtype == torch.bfloat16:
        xr, wr = bf16_round(x), bf16_round(w)
    else:
        xr, wr = x, w
    pre = _bn_ref(conv(xr, wr, bias, s, p), bn)
    res = torch.randn(pre.shape, generator=g)
    resr = bf16_round(res) if dtype == torch.bfloat16 else res
    ref = torch.relu(pre + resr)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    for tname in _tiles_for(cout):
        out = run_conv(x, w, bias, bn, s, p, dtype, TILES[tname], relu=True, residual=res)
        check("conv/%s/%s/%s" % (case, "f32" if dtype == torch.float32 else "bf16", tname), out, ref, tol)


V3_CASES = {  # name: (nd, N, cin, cout, k, stride, pad, spatial, residual)
    # 288-row tile kernel (LT_TILE3_288): pointwise and uniform-tap addressing, BN = 128 and 64, ragged last M tile, strides
    "v3_1x1_256_128": (2, 3, 256, 128, 1, 1, 0, (24, 24), True),      # M = 1728 = 6 x 288
    "v3_1x1_64_256_ragged": (2, 1, 64, 256, 1, 1, 0, (25, 23), True),  # M = 575: last tile 287 rows short of... ragged
    "v3_3x3_128_128": (2, 2, 128, 128, 3, 1, 1, (24, 24), True),
    "v3_3x3_64_64": (2, 2, 64, 64, 3, 1, 1, (20, 19), True),           # BN = 64
    "v3_3x3s2_128_256": (2, 2, 128, 256, 3, 2, 1, (24, 24), False),
    "v3_1x1s2_256_512": (2, 2, 256, 512, 1, 2, 0, (24, 24), False),    # strided 1x1 (downsample): uniform-tap path
    "v3_3x3x3_64_128": (3, 1, 64, 128, 3, 1, 1, (6, 8, 10), True),
    "v3_1x1_1024_256": (2, 1, 1024, 256, 1, 1, 0, (24, 12), True),     # 16 K steps
}


@pytest.mark.parametrize("case", list(V3_CASES))
def test_conv_v3_288(case):
    """288-row / 3-stage kernels vs torch (bf16), forced with LT_TILE3_288, and AUTO agrees (the 12-wave kernel; the role-specialised
    conv_igemm4 that LT_CONV_V4=1 used to select lost its A/B inside the forward and was removed in round 2)."""
    v4 = "0"
    nd, N, cin, cout, k, s, p, sp, with_res = V3_CASES[case]
    g = torch.Generator().manual_seed(len(case) * 7 + cin)
    x = torch.randn(N, cin, *sp, generator=g)
    w = torch.randn(cout, cin, *([k] * nd), generator=g) * (1.0 / (cin * k ** nd) ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    bn = _bn(cout, g)
    conv = F.conv2d if nd == 2 else F.conv3d
    pre = _bn_ref(conv(bf16_round(x), bf16_round(w), bias, s, p), bn)
    res = torch.randn(pre.shape, generator=g) if with_res else None
    ref = torch.relu(pre + bf16_round(res)) if with_res else torch.relu(pre)
    out = run_conv(x, w, bias, bn, s, p, torch.bfloat16, H.TILE3_288, relu=True, residual=res)
    check("conv_v3/%s/v4=%s/forced" % (case, v4), out, ref, 1.5e-2)
    out2 = run_conv(x, w, bias, bn, s, p, torch.bfloat16, 0, relu=True, residual=res)
    check("conv_v3/%s/v4=%s/auto" % (case, v4), out2, ref, 1.5e-2)
    out3 = run_conv(x, w, bias, bn, s, p, torch.bfloat16, H.TILE3_288, relu=False, relu_pre=True, residual=res)   # V2V-style epilogue
    ref3 = torch.relu(pre) + bf16_round(res) if with_res else torch.relu(pre)
    check("conv_v3/%s/v4=%s/relu_pre" % (case, v4), out3, ref3, 1.5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stem_conv_padded_channels(dtype):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 37, 41, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bn = _bn(64, g)
    xr, wr = (bf16_round(x), bf16_round(w)) if dtype == torch.bfloat16 else (x, w)
    ref = torch.relu(_bn_ref(F.conv2d(xr, wr, None, 2, 3), bn))
    for tname in ("auto", "128x64", "64x64", "v2_128x64", "v2_64x64", "v2_256x16", "direct"):
        out = run_conv(x, w, None, bn, 2, 3, dtype, TILES[tname], relu=True, cin_pad=E.min_cin_of(dtype))
        check("stem/%s/%s" % ("f32" if dtype == torch.float32 else "bf16", tname), out, ref, 2e-5 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_transposed_convs(dtype):
    g = torch.Generator().manual_seed(8)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    rd = (lambda t: bf16_round(t)) if dtype == torch.bfloat16 else (lambda t: t)
    # ConvTranspose2d 4x4 s2 p1 + BN + ReLU (pose_resnet deconv_layers)
    x = torch.randn(2, 64, 6, 7, generator=g); w = torch.randn(64, 256, 4, 4, generator=g) * 0.05; bn = _bn(256, g)
    ref = torch.relu(_bn_ref(F.conv_transpose2d(rd(x), rd(w), None, 2, 1), bn))
    for tname in ("auto", "128x128", "64x64", "v2_128x128", "v2_64x64", "v2_256x32", "direct"):
        check("deconv2d/%s/%s" % (dtype, tname), run_conv(x, w, None, bn, 2, 1, dtype, TILES[tname], transposed=True, relu=True), ref, tol)
    # ConvTranspose3d 2^3 s2 + BN + ReLU, then + skip (v2v Upsample3DBlock)
    x = torch.randn(1, 128, 4, 4, 4, generator=g); w = torch.randn(128, 64, 2, 2, 2, generator=g) * 0.1
    bias = torch.randn(64, generator=g) * 0.1; bn = _bn(64, g); skip = torch.randn(1, 64, 8, 8, 8, generator=g)
    ref = torch.relu(_bn_ref(F.conv_transpose3d(rd(x), rd(w), bias, 2), bn)) + rd(skip)
    for tname in ("auto", "128x64", "64x64", "v2_128x64", "v2_64x64", "v2_256x16", "direct"):
        check("deconv3d/%s/%s" % (dtype, tname),
              run_conv(x, w, bias, bn, 2, 0, dtype, TILES[tname], transposed=True, relu_pre=True, residual=skip), ref, tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_pool_layout_avgpool(dtype):
    g = torch.Generator().manual_seed(9)
    lib = H.lib(); st = torch.cuda.current_stream().cuda_stream; code = H.dtype_code(dtype)
    rd = (lambda t: bf16_round(t)) if dtype == torch.bfloat16 else (lambda t: t)
    b = E.PlanBuilder(DEV, dtype)
    x = torch.randn(2, 64, 19, 23, generator=g)
    y = b.maxpool(E.Act(to_cl(x, None, dtype)), 3, 2, 1, nd=2)   # (the builder keeps the input tensors alive)
    x3 = torch.randn(1, 32, 6, 8, 10, generator=g)
    y3 = b.maxpool(E.Act(to_cl(x3, None, dtype)), 2, 2, 0, nd=3)
    ga = b.global_avgpool(E.Act(to_cl(x, None, dtype)))
    b.finish().run_eager(st); torch.cuda.synchronize()
    check("maxpool2d/%s" % dtype, from_cl(y.t, 2), F.max_pool2d(rd(x), 3, 2, 1), 0.0)
    check("maxpool3d/%s" % dtype, from_cl(y3.t, 3), F.max_pool3d(rd(x3), 2, 2), 0.0)
    check("avgpool/%s" % dtype, ga.t.float().cpu().reshape(2, 64), rd(x).mean(dim=(2, 3)), 1e-6 if dtype == torch.float32 else 4e-3)
    # image layout transform: N,3,H,W fp32 -> N,H,W,cpad
    img = torch.randn(3, 3, 10, 12, generator=g).to(DEV)
    cpad = E.min_cin_of(dtype)
    out = torch.empty(3, 10, 12, cpad, dtype=dtype, device=DEV)
    H.check(lib.lt_nchw_to_nhwc(code, img.data_ptr(), out.data_ptr(), 3, 3, 120, cpad, st))
    torch.cuda.synchronize()
    exp = torch.zeros(3, 10, 12, cpad); exp[..., :3] = rd(img.cpu()).permute(0, 2, 3, 1)
    check("nchw_to_nhwc/%s" % dtype, out.float().cpu(), exp, 0.0)
    src = torch.randn(2, 70, 17, generator=g).to(DEV, dtype)  # N,HW,C
    dst = torch.empty(2, 17, 70, dtype=torch.float32, device=DEV)
    H.check(lib.lt_nhwc_to_nchw_f32(code, src.data_ptr(), dst.data_ptr(), 2, 17, 70, 17, st))
    torch.cuda.synchronize()
    check("nhwc_to_nchw/%s" % dtype, dst.cpu(), src.float().cpu().permute(0, 2, 1), 0.0)


def test_coord_volumes_bit_exact_and_rotated(golden_dir):
    from mvn.utils import op, volumetric
    base = np.array([[30.0, -20.0, 10.0], [-100.123, 50.5, 80.25], [1234.5, -987.6, 55.5]])
    for V in (8, 64):
        out = op.build_coord_volumes(base, 2500.0, V, device=DEV).cpu()
        ref = torch.stack([O.coord_volume(base[b], 2500.0, V) for b in range(3)])
        check("coord_volumes/eval/V%d (bit exact)" % V, out, ref, 0.0)
    th = [0.3, 1.7, -2.2]
    out = op.build_coord_volumes(base, 2500.0, 16, thetas=th, axis=(0, 1, 0), device=DEV).cpu()
    ref = torch.stack([O.coord_volume(base[b], 2500.0, 16, th[b], (0, 1, 0)) for b in range(3)])
    check("coord_volumes/rotated", out, ref, 2e-7)
    out = op.build_coord_volumes(base, 2500.0, 8, cmu_transfer=True, device=DEV).cpu()
    ref = torch.stack([O.coord_volume(base[b], 2500.0, 8, cmu_transfer=True) for b in range(3)])
    check("coord_volumes/cmu (bit exact)", out, ref, 0.0)
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    check("coord_volumes/golden theta=0.9", op.build_coord_volumes(g["cv_base"][None], 2500.0, 8, thetas=[0.9], device=DEV)[0].cpu(), g["cv_ref"], 2e-7)
    x = torch.randn(5, 7, 3)
    check("rotate_coord_volume", volumetric.rotate_coord_volume(x.to(DEV), 0.7, [1, 2, 3]).cpu(),
          (torch.from_numpy(O.rotation_matrix([1, 2, 3], 0.7)).float() @ x.reshape(-1, 3).t()).t().reshape(5, 7, 3), 1e-6)


@pytest.mark.parametrize("C", [32, 8, 5])
@pytest.mark.parametrize("method", ["sum", "max", "softmax", "conf"])
def test_unproject_vs_reference_golden(golden_dir, C, method):
    """Reference outputs of op.unproject_heatmaps: non-square maps (h/w swap quirk), a camera inside the cube (z <= 0
    mask), out-of-frame voxels (zero padding).  Gate 1e-5 * max|ref| (pixel coordinates ~50 carry fp32 rounding 4e-6)."""
    from mvn.utils import op
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    hm = torch.from_numpy(g["unproj_hm_C%d" % C]).to(DEV); conf = torch.from_numpy(g["unproj_cin_C%d" % C]).to(DEV)
    P = torch.from_numpy(g["unproj_P"]).to(DEV); cv = torch.from_numpy(g["unproj_cv"]).to(DEV)
    out = op.unproject_heatmaps(hm, P, cv, method, conf)
    assert out.shape == (2, C, 7, 7, 7)
    check("unproject/golden/%s/C%d" % (method, C), out.cpu(), g["unproj_%s_C%d" % (method, C)], 1e-5)
    if C % 8 == 0:
        outb = op.unproject_heatmaps(hm.bfloat16(), P, cv, method, conf)
        refb = O.unproject_heatmaps(bf16_round(hm.cpu()), P.cpu(), cv.cpu(), method, conf.cpu())
        check("unproject/bf16/%s/C%d" % (method, C), outb.float().cpu(), refb, 1e-2)


def test_unproject_bricks_xcd_pin_many_views():
    """Bricked voxel order (V=16), XCD-pinned sample mapping (B=8) and the NV>8 path, against the oracle."""
    from mvn.utils import op
    g = torch.Generator().manual_seed(11)
    for B, NV, V in ((8, 3, 16), (3, 9, 16), (1, 4, 32)):
        K, R, t = __import__("oracle.synth", fromlist=["x"]).ring_cameras(NV, 96, inside=(NV == 3))
        P = torch.from_numpy(O.resized_projection(K, R, t, (96, 96), (24, 24))).float()[None].repeat(B, 1, 1, 1).contiguous()
        hm = torch.randn(B, NV, 32, 24, 24, generator=g)
        base = torch.randn(B, 3, generator=g).numpy() * 100
        cv = torch.stack([O.coord_volume(base[b], 2500.0, V, 0.2 * b) for b in range(B)])
        for method in ("softmax", "max"):
            out = op.unproject_heatmaps(hm.to(DEV), P.to(DEV), cv.to(DEV), method)
            check("unproject/B%d_NV%d_V%d/%s" % (B, NV, V, method), out.cpu(), O.unproject_heatmaps(hm, P, cv, method), 1e-5)
    # raises like the reference for an unknown method (op.py:164)
    with pytest.raises(ValueError, match="Unknown volume_aggregation_method: bogus"):
        op.unproject_heatmaps(hm.to(DEV), P.to(DEV), cv.to(DEV), "bogus")


def test_softargmax_and_dlt_vs_reference_golden(golden_dir):
    from mvn.utils import multiview, op
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    vols = torch.from_numpy(g["int3d_in"]).to(DEV); cvs = torch.from_numpy(g["int3d_cv"]).to(DEV)
    for sm in (True, False):
        for layout in ("joint_major", "channels_last"):
            v = vols if layout == "joint_major" else vols.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
            c, p = op.integrate_tensor_3d_with_coordinates(v, cvs, softmax=sm)
            check("integrate3d/%s/softmax=%d/coords" % (layout, sm), c.cpu(), g["int3d_coords_%d" % sm], 5e-6)
            check("integrate3d/%s/softmax=%d/volumes" % (layout, sm), p.cpu(), g["int3d_vols_%d" % sm], 5e-6)
        c, h = op.integrate_tensor_2d(torch.from_numpy(g["int2d_in"]).to(DEV), softmax=sm)
        check("integrate2d/softmax=%d/coords" % sm, c.cpu(), g["int2d_coords_%d" % sm], 5e-6)
        check("integrate2d/softmax=%d/heatmaps" % sm, h.cpu(), g["int2d_hm_%d" % sm], 5e-6)
    P, pts, conf = (torch.from_numpy(g[k]).to(DEV) for k in ("dlt_P", "dlt_pts", "dlt_conf"))
    # the reference's own fp32 SVD carries ~1e-6*cond error: compare at 1e-3 like the oracle pin, and against the
    # exactly known 3D points for the noise-free sample
    check("dlt/conf", multiview.triangulate_batch_of_points(P, pts, conf).cpu(), g["dlt_out"], 1e-3)
    check("dlt/noconf", multiview.triangulate_batch_of_points(P, pts).cpu(), g["dlt_out_noconf"], 1e-3)
    check("dlt/known_answer", multiview.triangulate_batch_of_points(P, pts, conf)[0].cpu(), g["dlt_X"], 1e-5)


def test_softargmax3d_large_sharp():
    """64^3 volume, 17 joints, logits std 5 (the 'sharpened' regime), both layouts."""
    from mvn.utils import op
    g = torch.Generator().manual_seed(13)
    lg = torch.randn(2, 17, 64, 64, 64, generator=g) * 5
    cv = torch.stack([O.coord_volume(np.array([10.0 * b, -20.0, 30.0]), 2500.0, 64) for b in range(2)])
    rc, rv = O.integrate_tensor_3d_with_coordinates(lg.double(), cv.double())
    for layout in ("joint_major", "channels_last"):
        v = lg.to(DEV)
        if layout == "channels_last":
            v = v.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
        c, p = op.integrate_tensor_3d_with_coordinates(v, cv.to(DEV))
        rel = ((c.cpu().double() - rc).abs() / rc.abs().clamp(min=1.0)).max()
        record("integrate3d/64^3 sharp/%s coords max rel (1mm floor)" % layout, float(rel))
        assert float(rel) < 2e-5
        check("integrate3d/64^3 sharp/%s volumes" % layout, p.cpu(), rv, 1e-5)
        assert float((p.sum(dim=(2, 3, 4)) - 1).abs().max()) < 1e-4
    # the ReLU variant (op.py:90-94, not normalised) on the channels-last fast path
    rc0, rv0 = O.integrate_tensor_3d_with_coordinates((lg * 0.01).double(), cv.double(), softmax=False)
    v = (lg * 0.01).to(DEV).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    c0, p0 = op.integrate_tensor_3d_with_coordinates(v, cv.to(DEV), softmax=False)
    check("integrate3d/64^3 relu/channels_last coords", c0.cpu(), rc0.float(), 2e-4)
    check("integrate3d/64^3 relu/channels_last volumes", p0.cpu(), rv0.float(), 1e-6)
    c1, p1 = op.integrate_tensor_3d_with_coordinates((lg * 0.01).to(DEV), cv.to(DEV), softmax=False)
    check("integrate3d/64^3 relu/joint_major coords", c1.cpu(), rc0.float(), 2e-4)
    check("integrate3d/64^3 relu/joint_major volumes", p1.cpu(), rv0.float(), 1e-6)


@pytest.mark.parametrize("V,J", [(24, 17), (8, 3), (20, 32)])
def test_softargmax3d_planar_ragged(V, J):
    """Joint-major volumes whose voxel count is not a multiple of the 4096-voxel chunk of the vectorised planar kernels (ragged last
    workgroup, lanes without voxels), 1..32 joints, multiplier != 1."""
    from mvn.utils import op
    g = torch.Generator().manual_seed(V * 100 + J)
    lg = torch.randn(3, J, V, V, V, generator=g) * 3
    cv = torch.stack([O.coord_volume(np.array([100.0 * b, 50.0, -30.0]), 2000.0, V) for b in range(3)])
    for sm in (True, False):
        x = lg if sm else lg * 0.01
        rc, rv = O.integrate_tensor_3d_with_coordinates((x * 2.0).double(), cv.double(), softmax=sm)
        c, p = op.integrate_tensor_3d_with_coordinates(x.to(DEV) * 2.0, cv.to(DEV), softmax=sm)
        rel = ((c.cpu().double() - rc).abs() / rc.abs().clamp(min=1.0)).max()
        record("integrate3d/%d^3 J=%d softmax=%d planar coords max rel (1mm floor)" % (V, J, sm), float(rel))
        assert float(rel) < (2e-5 if sm else 2e-4)
        check("integrate3d/%d^3 J=%d softmax=%d planar volumes" % (V, J, sm), p.cpu(), rv.float(), 1e-5)


def test_conv3d_column_walk_fp32_store(monkeypatch):
    """The 3^3 32 -> 32 column-walk kernel with LT_EPI_STORE_F32 (the V2V layers of the mixed-precision training step: bf16 operands, fp32
    accumulation AND fp32 output) against the implicit-GEMM path on the same operands (LT_HALO_NO_COL=1: summation order only) and torch."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(4, 32, 16, 64, 64, generator=g)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * (1.0 / (32 * 27) ** 0.5)
    bias = torch.randn(32, generator=g) * 0.1
    st = torch.cuda.current_stream().cuda_stream

    def run():
        b = E.PlanBuilder(DEV, torch.bfloat16)
        y = b.conv(E.Act(to_cl(x, None, torch.bfloat16)), w, bias, None, stride=1, pad=1, out_f32=True)
        b.finish().run_eager(st); torch.cuda.synchronize()
        assert y.t.dtype == torch.float32
        return from_cl(y.t, 3)
    monkeypatch.delenv("LT_HALO_NO_COL", raising=False)
    col = run()
    monkeypatch.setenv("LT_HALO_NO_COL", "1")
    igemm = run()
    monkeypatch.delenv("LT_HALO_NO_COL", raising=False)
    check("conv3d 32->32 column walk, fp32 store vs implicit GEMM", col, igemm, 2e-5)
    check("conv3d 32->32 column walk, fp32 store vs torch on bf16 operands", col, F.conv3d(bf16_round(x), bf16_round(w), bias, 1, 1), 1e-4)


def test_conv_fp32_residual_on_a_bf16_convolution():
    """LT_EPI_RES_F32 (with LT_EPI_STORE_F32): bf16 operands, fp32 accumulation, an fp32 residual added in the epilogue, fp32 output -- the
    input-gradient accumulation of the mixed-precision training step.  Vector and ragged-channel epilogues, strided / transposed phases."""
    g = torch.Generator().manual_seed(21)
    st = torch.cuda.current_stream().cuda_stream
    cases = [(2, 64, 128, 3, 1, 1, False, (20, 24)), (3, 32, 17, 1, 1, 0, False, (4, 6, 8)), (2, 32, 64, 4, 2, 1, True, (6, 8)), (3, 32, 64, 3, 1, 1, False, (4, 8, 8))]
    for nd, cin, cout, k, s_, p_, tr, sp in cases:
        x = torch.randn(2, cin, *sp, generator=g)
        w = torch.randn(*((cin, cout) if tr else (cout, cin)), *([k] * nd), generator=g) * (1.0 / (cin * k ** nd) ** 0.5)
        conv = {(2, False): F.conv2d, (3, False): F.conv3d, (2, True): F.conv_transpose2d, (3, True): F.conv_transpose3d}[(nd, tr)]
        ref0 = conv(bf16_round(x), bf16_round(w), None, stride=s_, padding=p_)
        res = torch.randn(ref0.shape, generator=g) * 3.0 + 1e-3 * torch.randn(ref0.shape, generator=g)      # not representable in bf16
        for tile in ((0, 4, 14) if not tr else (0,)):
            b = E.PlanBuilder(DEV, torch.bfloat16, tile_override=tile)
            y = b.conv(E.Act(to_cl(x, None, torch.bfloat16)), w, None, None, stride=s_, pad=p_, transposed=tr, residual=E.Act(to_cl(res, None, torch.float32)),
                       out_f32=True, residual_f32=True)
            b.finish().run_eager(st); torch.cuda.synchronize()
            out = from_cl(y.t, nd)
            check("conv bf16 + fp32 residual nd%d %d->%d k%d%s tile%d" % (nd, cin, cout, k, " T" if tr else "", tile), out - res, ref0, 1e-4)


FP8_CASES = [  # nd, N, cin, cout, k, stride, pad, spatial
    (3, 2, 32, 32, 3, 1, 1, (8, 8, 8)),        # V2V 3^3 32 -> 32
    (3, 1, 16, 32, 3, 1, 1, (6, 8, 10)),       # 16 input channels: one 16-byte vector per voxel
    (3, 2, 64, 64, 3, 1, 1, (4, 8, 8)),
    (3, 1, 128, 128, 3, 1, 1, (4, 4, 4)),
    (2, 2, 256, 64, 1, 1, 0, (9, 7)),          # pointwise fast path
    (3, 2, 32, 64, 3, 1, 1, (4, 8, 16)),       # halo kernel shapes (with the two above: 32 -> 32 at 8^3, 64 -> 64 at 4 x 8 x 8)
    (3, 1, 64, 64, 3, 1, 1, (8, 16, 8)),
]


@pytest.mark.parametrize("case", FP8_CASES, ids=lambda c: "nd%d_%dto%d_k%d" % (c[0], c[2], c[3], c[4]))
def test_conv_fp8_operands(case):
    """lt_conv_fwd(dtype = LT_FP8): e4m3 operands (per-tensor amax scaling by lt_amax_f32 / lt_quant_fp8, the scale product in the epilogue's
    ``scale``, the bias in ``shift``), fp32 accumulation and storage, an fp32 residual -- against torch's conv over the SAME quantised operands
    (torch.float8_e4m3fn is the OCP format gfx950 converts to): products of two e4m3 values are exact in fp32, so only the summation
    order differs.  Also checks the quantisation kernels bit for bit against torch's cast."""
    nd, N, cin, cout, k, s_, p_, sp = case
    lib = H.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(cin * 3 + cout + k)
    x = torch.randn(N, cin, *sp, generator=g) * 1.7
    w = torch.randn(cout, cin, *([k] * nd), generator=g) * (1.0 / (cin * k ** nd) ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    # ---- quantisation on the device: amax -> scale -> e4m3 bytes
    xcl = to_cl(x)                                       # N,D,H,W,C fp32
    amax = torch.zeros(2, dtype=torch.float32, device=DEV)
    scales = torch.zeros(2, dtype=torch.float32, device=DEV)
    x8 = torch.empty(xcl.shape, dtype=torch.uint8, device=DEV)
    H.check(lib.lt_amax_f32(xcl.data_ptr(), xcl.numel(), amax.data_ptr(), st), "lt_amax_f32")
    H.check(lib.lt_quant_fp8(xcl.data_ptr(), x8.data_ptr(), xcl.numel(), amax.data_ptr(), scales.data_ptr(), st), "lt_quant_fp8")
    torch.cuda.synchronize()
    sx = float(x.abs().max()) / 448.0
    assert abs(float(amax[0]) - float(x.abs().max())) == 0.0 and abs(float(scales[0]) - np.float32(np.float32(x.abs().max()) / np.float32(448.0))) <= 1e-12
    sx = float(scales[0])
    xq_ref = (xcl.cpu() * np.float32(1.0 / np.float32(sx))).to(torch.float8_e4m3fn)
    mism = (x8.cpu() != xq_ref.view(torch.uint8)).float().mean()
    record("fp8/quantisation bytes differing from torch's e4m3fn cast (nd%d %d->%d)" % (nd, cin, cout), float(mism))
    assert float(mism) == 0.0, float(mism)
    sw = float(w.abs().max()) / 448.0
    wq = (w / sw).to(torch.float8_e4m3fn)
    conv = F.conv2d if nd == 2 else F.conv3d
    xq_nc = xq_ref.float().permute(0, 4, 1, 2, 3)
    xq_nc = xq_nc[:, :, 0] if nd == 2 else xq_nc
    ref0 = conv(xq_nc, wq.float(), None, stride=s_, padding=p_) * (sx * sw) + bias.reshape([1, -1] + [1] * nd)
    res = torch.randn(ref0.shape, generator=g)
    for with_res in (False, True):
        for tile in (0, TILES["v2_256x32"] if E.cout_pad_of(cout) == 32 else TILES["v2_128x64"]):
            b = E.PlanBuilder(DEV, torch.float8_e4m3fn, tile_override=tile)
            # the epilogue is (acc + bias) * scale + shift: scale = sx * sw and shift = the convolution's bias, through the BatchNorm slot
            bn = (torch.full((cout,), sx * sw), bias, torch.zeros(cout), torch.ones(cout) - 1e-5)          # invstd = 1 / sqrt(var + 1e-5) = 1
            xa = E.Act(x8.view(torch.float8_e4m3fn))
            ra = E.Act(to_cl(res)) if with_res else None
            y = b.conv(xa, wq.float(), None, bn, stride=s_, pad=p_, residual=ra, out_f32=True, residual_f32=with_res, relu=with_res)
            b.finish().run_eager(st); torch.cuda.synchronize()
            ref = torch.relu(ref0 + res) if with_res else ref0
            check("conv fp8 nd%d %d->%d k%d tile%d%s" % (nd, cin, cout, k, tile, " +res" if with_res else ""), from_cl(y.t, nd), ref, 1e-4)
    # bf16 store with a bf16 residual (the 16-bit-activation training step: train_precision "fp8v2v")
    b = E.PlanBuilder(DEV, torch.float8_e4m3fn)
    bn = (torch.full((cout,), sx * sw), bias, torch.zeros(cout), torch.ones(cout) - 1e-5)
    y = b.conv(E.Act(x8.view(torch.float8_e4m3fn)), wq.float(), None, bn, stride=s_, pad=p_, residual=E.Act(to_cl(res, None, torch.bfloat16)), relu=True)
    b.finish().run_eager(st); torch.cuda.synchronize()
    assert y.t.dtype == torch.bfloat16
    check("conv fp8 nd%d %d->%d k%d bf16 store + bf16 residual" % (nd, cin, cout, k), from_cl(y.t, nd), torch.relu(ref0 + bf16_round(res)), 1.5e-2)
    # the halo kernel on e4m3 operands (input halo of the 4 x 8 x 8 tile in LDS, bf16 stores): forced, with and without the residual / ReLU epilogue
    if nd == 3 and k == 3 and (cin, cout) in ((32, 32), (32, 64), (64, 64)) and sp[0] % 4 == 0 and sp[1] % 8 == 0 and sp[2] % 8 == 0:
        for with_res in (False, True):
            b = E.PlanBuilder(DEV, torch.float8_e4m3fn, tile_override=H.TILE_HALO)
            y = b.conv(E.Act(x8.view(torch.float8_e4m3fn)), wq.float(), None, bn, stride=s_, pad=p_,
                       residual=E.Act(to_cl(res, None, torch.bfloat16)) if with_res else None, relu=with_res)
            b.finish().run_eager(st); torch.cuda.synchronize()
            refh = torch.relu(ref0 + bf16_round(res)) if with_res else ref0
            check("conv fp8 halo kernel %d->%d @%s%s" % (cin, cout, "x".join(map(str, sp)), " +res" if with_res else ""), from_cl(y.t, nd), refh, 1.5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_epilogue_variants(dtype):
    """fp32 store from bf16 compute (V2V logits), sigmoid head (linear as 1x1 conv over an N-pixel row), ragged Cout."""
    g = torch.Generator().manual_seed(15)
    rd = (lambda t: bf16_round(t)) if dtype == torch.bfloat16 else (lambda t: t)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    st = torch.cuda.current_stream().cuda_stream
    for tile in (0, 4, 14, 3, 13):
        b = E.PlanBuilder(DEV, dtype, tile_override=tile)
        x = torch.randn(2, 32, 6, 7, 5, generator=g); w = torch.randn(17, 32, 1, 1, 1, generator=g) * 0.2; bias = torch.randn(17, generator=g)
        y = b.conv(E.Act(to_cl(x, None, dtype)), w, bias, None, out_f32=True)
        xl = torch.randn(6, 256, generator=g); wl = torch.randn(40, 256, generator=g) * 0.1; bl = torch.randn(40, generator=g)
        yl = b.conv(E.Act(xl.reshape(1, 1, 1, 6, 256).to(DEV, dtype)), wl[:, :, None, None], bl, None, sigmoid=True, out_f32=True)
        b.finish().run_eager(st); torch.cuda.synchronize()
        assert y.t.dtype == torch.float32
        check("conv1x1x1->17 fp32 store/%s/tile%d" % (dtype, tile), from_cl(y.t, 3), F.conv3d(rd(x), rd(w), bias), tol)
        check("linear+sigmoid/%s/tile%d" % (dtype, tile), yl.t.reshape(6, 40).cpu(), torch.sigmoid(F.linear(rd(xl), rd(wl), bl)), tol)


HALO_CASES = {  # name: (N, cin, cout, k, (D,H,W), dtypes)
    "halo_3x3_32_32": (2, 32, 32, 3, (8, 16, 16), ("f32", "bf16")),
    "halo_3x3_16_32": (1, 16, 32, 3, (4, 16, 8), ("f32", "bf16")),
    "halo_3x3_64_64": (1, 64, 64, 3, (8, 8, 16), ("bf16",)),
    "halo_3x3_32_64": (3, 32, 64, 3, (4, 8, 16), ("bf16",)),
    "halo_7x7_32_16": (1, 32, 16, 7, (8, 16, 8), ("bf16",)),
    "halo_3x3_32_32_xcdpin": (8, 32, 32, 3, (4, 8, 8), ("bf16",)),
    # >= 1024 tiles: the persistent kernel (weights resident, double-buffered halo), raster dealing and XCD-pinned dealing
    "halo_3x3_32_32_persist": (2, 32, 32, 3, (32, 64, 64), ("bf16",)),
    "halo_3x3_32_32_persist_xcdpin": (8, 32, 32, 3, (16, 32, 64), ("bf16",)),
    # 7^3 at a size with several tiles per CU (fragment ring across weight chunks)
    "halo_7x7_32_16_big": (1, 32, 16, 7, (32, 32, 32), ("bf16",)),
    # round 6: the 7^3 layers of the exact-fp32 kernel set -- 32 -> 16 in two channel phases of 16 (the halo image of 32 fp32 channels does not fit LDS),
    # 16 -> 32 (the front layer's input gradient in the fp32 training step); one tile per axis and several, every face padded
    "halo_7x7_32_16_f32": (1, 32, 16, 7, (8, 16, 8), ("f32",)),
    "halo_7x7_32_16_f32_big": (2, 32, 16, 7, (16, 24, 32), ("f32",)),
    "halo_7x7_16_32_f32": (1, 16, 32, 7, (8, 16, 16), ("f32",)),
    # 3^3 32 -> 32 in two channel phases (two workgroups per CU), XCD-pinned samples, against the one-phase kernel (LT_HALO_F3=1)
    "halo_3x3_32_32_f32_xcdpin": (8, 32, 32, 3, (8, 16, 16), ("f32",)),
}


@pytest.mark.parametrize("case", list(HALO_CASES))
def test_conv3d_halo_kernel(case):
    """LDS-resident halo conv3d (LT_TILE_HALO) vs torch conv3d, incl. zero padding at every face, residual + ReLU."""
    N, cin, cout, k, sp, dts = HALO_CASES[case]
    g = torch.Generator().manual_seed(len(case))
    x = torch.randn(N, cin, *sp, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) * (1.0 / (cin * k ** 3) ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    bn = _bn(cout, g)
    res = torch.randn(N, cout, *sp, generator=g)
    for dname in dts:
        dtype = torch.float32 if dname == "f32" else torch.bfloat16
        rd = (lambda t: bf16_round(t)) if dtype == torch.bfloat16 else (lambda t: t)
        ref = torch.relu(_bn_ref(F.conv3d(rd(x), rd(w), bias, 1, k // 2), bn) + rd(res))
        out = run_conv(x, w, bias, bn, 1, k // 2, dtype, H.TILE_HALO, relu=True, residual=res)
        check("conv3d_halo/%s/%s" % (case, dname), out, ref, 2e-5 if dtype == torch.float32 else 1.5e-2)
        out2 = run_conv(x, w, bias, bn, 1, k // 2, dtype, 0, relu=True, residual=res)   # AUTO (generic or halo) agrees
        check("conv3d_auto/%s/%s" % (case, dname), out2, ref, 2e-5 if dtype == torch.float32 else 1.5e-2)
        if case.endswith("_f32_big"):          # the generic tile (LT_HALO_NO_F7=1, read per call) computes the same sums in another order
            os.environ["LT_HALO_NO_F7"] = "1"
            try:
                out4 = run_conv(x, w, bias, bn, 1, k // 2, dtype, 0, relu=True, residual=res)
            finally:
                del os.environ["LT_HALO_NO_F7"]
            check("conv3d_halo vs generic/%s" % case, out, out4, 2e-6)
        if case == "halo_3x3_32_32_f32_xcdpin":
            os.environ["LT_HALO_F3"] = "1"
            try:
                out5 = run_conv(x, w, bias, bn, 1, k // 2, dtype, H.TILE_HALO, relu=True, residual=res)
            finally:
                del os.environ["LT_HALO_F3"]
            check("conv3d_halo two phases vs one/%s" % case, out, out5, 2e-6)
        if "persist" in case:   # no residual, no ReLU: the epilogue without prefetched vectors
            ref3 = _bn_ref(F.conv3d(rd(x), rd(w), bias, 1, k // 2), bn)
            out3 = run_conv(x, w, bias, bn, 1, k // 2, dtype, H.TILE_HALO, relu=False, residual=None)
            check("conv3d_halo_nores/%s/%s" % (case, dname), out3, ref3, 1.5e-2)


@pytest.mark.parametrize("N,hw,cin", [(2, (64, 128), 3), (1, (100, 84), 3), (3, (37, 53), 1), (1, (384, 384), 3)])
def test_stem_pool(N, hw, cin):
    """lt_stem_pool_fwd (conv 7x7/2 + BN + ReLU + max pool 3x3/2 in one pass) vs torch, and bit-identical to the two separate
    launches (conv then pool) it replaces; ragged tiles, odd sizes, pool windows and conv taps over every border."""
    g = torch.Generator().manual_seed(N * 1000 + hw[0])
    x = torch.randn(N, cin, *hw, generator=g)
    w = torch.randn(64, cin, 7, 7, generator=g) * (1.0 / (cin * 49) ** 0.5)
    bn = _bn(64, g)
    b = E.PlanBuilder(DEV, torch.bfloat16)
    xa = E.Act(to_cl(x, 8, torch.bfloat16))
    assert b.can_stem_pool(xa, w, 2, 3, (3, 2, 1))
    y = b.stem_pool(xa, w, bn)
    y1 = b.conv(xa, w, None, bn, stride=2, pad=3, relu=True)
    y2 = b.maxpool(y1, 3, 2, 1, nd=2)
    b.finish().run_eager(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = F.max_pool2d(bf16_round(torch.relu(_bn_ref(F.conv2d(bf16_round(x), bf16_round(w), None, 2, 3), bn))), 3, 2, 1)
    out = from_cl(y.t, 2)
    assert tuple(out.shape) == tuple(ref.shape)
    check("stem_pool/%dx%dx%d" % (N, hw[0], hw[1]), out, ref, 1.5e-2)
    two = from_cl(y2.t, 2)
    record("stem_pool/%dx%dx%d vs conv+pool max abs diff" % (N, hw[0], hw[1]), float((out - two).abs().max()))
    # same bf16 inputs, fp32 accumulation in a different order: a rounding boundary may flip the last bf16 bit
    assert float((out - two).abs().max()) <= 2 ** -7 * float(two.abs().max())
    if cin == 3:   # x_layout 1: the same kernel reading the fp32 (N, 3, H, W) images where they are (a pre op of the plan)
        b2 = E.PlanBuilder(DEV, torch.bfloat16)
        cell = {"ptr": None}
        y3 = b2.stem_pool(E.Act(torch.empty(N, 1, hw[0], hw[1], 8, dtype=torch.bfloat16, device=DEV)), w, bn, image_cell=cell)
        plan = b2.finish()
        assert plan.npre == 1
        with pytest.raises(RuntimeError, match="image_cell"):
            plan.run(torch.cuda.current_stream().cuda_stream)
        xd = x.to(DEV).contiguous()
        cell["ptr"] = xd.data_ptr()
        plan.run(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(from_cl(y3.t, 2), out), "fp32-image path must round exactly like lt_nchw_to_nhwc + the bf16 path"


PW_CASES = {  # name: (N, cin, cout, (D,H,W), deconv)
    "pw_16_32": (2, 16, 32, (8, 8, 16), False),
    "pw_32_64": (1, 32, 64, (4, 8, 8), False),
    "pw_64_32": (3, 64, 32, (4, 4, 8), False),
    "pw_128_64": (1, 128, 64, (4, 4, 4), False),
    "deconv2_64_32": (2, 64, 32, (4, 8, 8), True),
    "deconv2_128_64": (1, 128, 32, (2, 4, 8), True),     # 8 phases x 64 x 128 weights would not fit the 64 KB budget at Cout = 64
    "deconv2_32_32_odd": (1, 32, 32, (2, 4, 8), True),
}


@pytest.mark.parametrize("stream", ["1", "0"], ids=["pw_stream", "igemm"])
@pytest.mark.parametrize("case", list(PW_CASES))
def test_conv_pointwise_stream(case, stream, monkeypatch):
    """conv_pw (streaming kernel of the single-tap layers: 1x1x1 convs and 2x2x2 stride-2 deconvs of V2V, bf16) and the implicit
    GEMM it replaces (LT_CONV_NO_PW=1) vs torch: affine only, ReLU + residual."""
    if stream == "0":
        monkeypatch.setenv("LT_CONV_NO_PW", "1")
    else:
        monkeypatch.delenv("LT_CONV_NO_PW", raising=False)
    N, cin, cout, sp, deconv = PW_CASES[case]
    g = torch.Generator().manual_seed(len(case) + 5)
    x = torch.randn(N, cin, *sp, generator=g)
    rd = bf16_round
    bias = torch.randn(cout, generator=g) * 0.1
    bn = _bn(cout, g)
    if deconv:
        w = torch.randn(cin, cout, 2, 2, 2, generator=g) * (1.0 / cin ** 0.5)
        conv = F.conv_transpose3d(rd(x), rd(w), bias, 2, 0)
        osp = tuple(2 * v for v in sp)
    else:
        w = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
        conv = F.conv3d(rd(x), rd(w), bias)
        osp = sp
    res = torch.randn(N, cout, *osp, generator=g)
    out = run_conv(x, w, bias, bn, 2 if deconv else 1, 0, torch.bfloat16, 0, transposed=deconv)
    check("conv_pw=%s/%s/affine" % (stream, case), out, _bn_ref(conv, bn), 1.5e-2)
    out2 = run_conv(x, w, bias, bn, 2 if deconv else 1, 0, torch.bfloat16, 0, transposed=deconv, relu=True, residual=res)
    check("conv_pw=%s/%s/relu_res" % (stream, case), out2, torch.relu(_bn_ref(conv, bn) + rd(res)), 1.5e-2)


COL_CASES = {  # name: (N, (D,H,W)): 3^3 32->32 bf16 with >= 256 columns of >= 2 tiles -> the column-walking kernel
    "pin_1col_4deep": (8, (16, 32, 64)),        # XCD-pinned samples, one column per workgroup
    "pin_2col_2deep": (16, (8, 32, 64)),        # two columns per workgroup, two tiles per column (ring wraps between columns)
    "raster_uneven": (5, (16, 64, 64)),         # 320 columns on 256 workgroups, raster dealing
}


@pytest.mark.parametrize("col", ["1", "0"], ids=["column_walk", "persistent"])
@pytest.mark.parametrize("case", list(COL_CASES))
def test_conv3d_halo_column_walk(case, col, monkeypatch):
    """conv3d_halo_col_kernel (sliding window of halo planes, epilogue deferred under the next tile's MFMAs) and the persistent
    kernel it replaces (LT_HALO_NO_COL=1) vs torch conv3d: residual + ReLU, affine only, ReLU only."""
    if col == "0":
        monkeypatch.setenv("LT_HALO_NO_COL", "1")
    else:
        monkeypatch.delenv("LT_HALO_NO_COL", raising=False)
    N, sp = COL_CASES[case]
    g = torch.Generator().manual_seed(len(case) + 11)
    x = torch.randn(N, 32, *sp, generator=g)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * (1.0 / (32 * 27) ** 0.5)
    bias = torch.randn(32, generator=g) * 0.1
    bn = _bn(32, g)
    res = torch.randn(N, 32, *sp, generator=g)
    rd = bf16_round
    conv = F.conv3d(rd(x), rd(w), bias, 1, 1)
    ref = torch.relu(_bn_ref(conv, bn) + rd(res))
    out = run_conv(x, w, bias, bn, 1, 1, torch.bfloat16, H.TILE_HALO, relu=True, residual=res)
    check("conv3d_halo_col=%s/%s/res" % (col, case), out, ref, 1.5e-2)
    out2 = run_conv(x, w, bias, bn, 1, 1, torch.bfloat16, 0, relu=False, residual=None)
    check("conv3d_halo_col=%s/%s/plain" % (col, case), out2, _bn_ref(conv, bn), 1.5e-2)
    out3 = run_conv(x, w, None, None, 1, 1, torch.bfloat16, 0, relu=True, residual=None)
    check("conv3d_halo_col=%s/%s/relu_only" % (col, case), out3, torch.relu(F.conv3d(rd(x), rd(w), None, 1, 1)), 1.5e-2)


@pytest.mark.parametrize("planar", [False, True], ids=["channels_last", "planar"])
@pytest.mark.parametrize("nlayers,J", [(3, 17), (2, 32), (1, 17), (3, 5)])
def test_pwchain(nlayers, J, planar):
    """lt_pwchain_fwd (pointwise chain in registers) vs torch: every layer's output rounded to bf16 except the fp32 last one;
    channels-last rows or planar (N, J, D, H, W) storage behind the same channels-last Act view."""
    g = torch.Generator().manual_seed(40 + nlayers + J)
    x = torch.randn(2, 8, 8, 16, 32, generator=g)                      # 2048 voxels, channels last
    widths = [32] * (nlayers - 1) + [J]
    layers, cin = [], 32
    for i, co in enumerate(widths):
        w = torch.randn(co, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
        bias = torch.randn(co, generator=g) * 0.1
        last = i + 1 == nlayers
        bn = None if last else _bn(co, g)
        layers.append((w, bias, bn, not last))
        cin = co
    b = E.PlanBuilder(DEV, torch.bfloat16)
    xa = E.Act(x.to(DEV).to(torch.bfloat16))
    assert b.can_chain_pointwise(xa, layers)
    y = b.pwchain(xa, layers, planar=planar)
    assert y.t.is_contiguous() != planar and (not planar or y.t.permute(0, 4, 1, 2, 3).is_contiguous())
    b.finish().run_eager(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cur = bf16_round(x).permute(0, 4, 1, 2, 3)
    for i, (w, bias, bn, relu) in enumerate(layers):
        cur = F.conv3d(cur, bf16_round(w), bias)
        if bn is not None:
            cur = _bn_ref(cur, bn)
        if relu:
            cur = torch.relu(cur)
        if i + 1 < nlayers:
            cur = bf16_round(cur)
    ref = cur.permute(0, 2, 3, 4, 1)
    assert y.t.dtype == torch.float32 and tuple(y.t.shape) == tuple(ref.shape)
    check("pwchain/L%d_J%d%s" % (nlayers, J, "/planar" if planar else ""), y.t.cpu(), ref, 1e-2)


def test_unproject_bf16_bricked_volumes(monkeypatch):
    """bf16 / C = 32 / bricked volumes through the default dispatch (quad kernel for 4 / 8 views + softmax, generic gather otherwise),
    every aggregation, vs the oracle.  (An LDS-staged variant, opt-in with LT_UNPROJ_LDS=1, measured 0.90 vs 0.57 ms and was removed
    in round 2.)"""
    from mvn.utils import op
    staged = "0"
    synth = __import__("oracle.synth", fromlist=["x"])
    g = torch.Generator().manual_seed(21)
    # (B, views, volume, map size, a camera inside the cube): NV == 4 + softmax takes the quad kernel (unproject_q4_kernel)
    for B, NV, V, hw, inside in ((8, 4, 32, 24, False), (2, 3, 16, 96, True), (1, 8, 16, 24, False), (3, 4, 16, 96, True)):
        K, R, t = synth.ring_cameras(NV, 96, inside=inside)
        P = torch.from_numpy(O.resized_projection(K, R, t, (96, 96), (hw, hw))).float()[None].repeat(B, 1, 1, 1).contiguous()
        hm = torch.randn(B, NV, 32, hw, hw, generator=g)
        conf = torch.rand(B, NV, 32, generator=g) + 0.1
        base = torch.randn(B, 3, generator=g).numpy() * 100
        cv = torch.stack([O.coord_volume(base[b], 2500.0, V, 0.3 * b) for b in range(B)])
        for method in ("softmax", "sum", "max", "conf", "conf_norm"):
            out = op.unproject_heatmaps(hm.to(DEV).bfloat16(), P.to(DEV), cv.to(DEV), method, conf.to(DEV))
            ref = O.unproject_heatmaps(bf16_round(hm), P, cv, method, conf)
            check("unproject_bf16/staged=%s/B%d_NV%d_V%d_hw%d/%s" % (staged, B, NV, V, hw, method), out.float().cpu(), ref, 1e-2)
        if NV == 4 and staged == "0":   # the quad kernel against the generic gather kernel it replaces
            out = op.unproject_heatmaps(hm.to(DEV).bfloat16(), P.to(DEV), cv.to(DEV), "softmax")
            monkeypatch.setenv("LT_UNPROJ_NO_Q4", "1")
            gen = op.unproject_heatmaps(hm.to(DEV).bfloat16(), P.to(DEV), cv.to(DEV), "softmax")
            monkeypatch.delenv("LT_UNPROJ_NO_Q4")
            check("unproject_bf16/q4 vs generic/B%d_V%d_hw%d" % (B, V, hw), out.float().cpu(), gen.float().cpu(), 1e-2)


V5_CASES = {  # name: (nd, N, cin, cout, k, stride, pad, spatial, residual)
    "v5_1x1_256_512": (2, 2, 256, 512, 1, 1, 0, (24, 24), True),
    "v5_1x1_1024_256": (2, 1, 1024, 256, 1, 1, 0, (24, 24), True),       # 32 K steps, one N tile
    "v5_3x3_128_256": (2, 2, 128, 256, 3, 1, 1, (24, 24), True),
    "v5_1x1_64_256_ragged": (2, 1, 64, 256, 1, 1, 0, (25, 23), True),     # M = 575: ragged last tile
    "v5_3x3s2_128_256": (2, 2, 128, 256, 3, 2, 1, (24, 24), False),
    "v5_1x1s2_256_512": (2, 2, 256, 512, 1, 2, 0, (24, 24), False),
    "v5_3x3x3_64_256": (3, 1, 64, 256, 3, 1, 1, (6, 8, 10), True),
}


@pytest.mark.parametrize("bsrc", ["registers", "registers_144", "lds", "v7_mfma32"])
@pytest.mark.parametrize("case", list(V5_CASES))
def test_conv_v5_288x256(case, bsrc, monkeypatch):
    """288x256 tile / 32-element K steps (Cout % 256 == 0), forced with LT_CONV_V5=1, vs torch (bf16): conv_igemm6 (weights read
    from global memory in fragment order, activations in a six-stage ring) and conv_igemm5 (both operands staged, LT_CONV_NO_V6=1)."""
    monkeypatch.setenv("LT_CONV_V5", "1")
    if bsrc == "lds":
        monkeypatch.setenv("LT_CONV_NO_V6", "1")
    else:
        monkeypatch.delenv("LT_CONV_NO_V6", raising=False)
    monkeypatch.setenv("LT_CONV_V6_BM144", "1" if bsrc == "registers_144" else "0")   # 144-row tiles, two workgroups per CU
    if bsrc == "v7_mfma32":     # conv_igemm7 (default): the same tile on 32x32x16 MFMAs (weights packed by lt_conv_pack_weights32 when the plan is built)
        monkeypatch.delenv("LT_CONV_NO_V7", raising=False)
    else:
        monkeypatch.setenv("LT_CONV_NO_V7", "1")
    nd, N, cin, cout, k, s, p, sp, with_res = V5_CASES[case]
    g = torch.Generator().manual_seed(len(case) * 5 + cin)
    x = torch.randn(N, cin, *sp, generator=g)
    w = torch.randn(cout, cin, *([k] * nd), generator=g) * (1.0 / (cin * k ** nd) ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    bn = _bn(cout, g)
    conv = F.conv2d if nd == 2 else F.conv3d
    pre = _bn_ref(conv(bf16_round(x), bf16_round(w), bias, s, p), bn)
    res = torch.randn(pre.shape, generator=g) if with_res else None
    ref = torch.relu(pre + bf16_round(res)) if with_res else torch.relu(pre)
    out = run_conv(x, w, bias, bn, s, p, torch.bfloat16, H.TILE3_288, relu=True, residual=res)
    check("conv_v5/%s/%s/forced" % (bsrc, case), out, ref, 1.5e-2)
    out3 = run_conv(x, w, bias, bn, s, p, torch.bfloat16, H.TILE3_288, relu=False, relu_pre=True, residual=res)
    ref3 = torch.relu(pre) + bf16_round(res) if with_res else torch.relu(pre)
    check("conv_v5/%s/relu_pre" % case, out3, ref3, 1.5e-2)


@pytest.mark.parametrize("bsrc", ["registers", "lds", "v7_mfma32"])
def test_deconv4x4_phases_288x256(bsrc, monkeypatch):
    """ConvTranspose2d 4x4 / stride 2 / pad 1 (the backbone's deconv head) through the 288x256 kernels: every output parity is one
    launch of the single-phase kernel (conv3_try loops over the phases); vs torch and vs the implicit GEMM that took all phases."""
    monkeypatch.setenv("LT_CONV_V5", "1")
    if bsrc == "lds":
        monkeypatch.setenv("LT_CONV_NO_V6", "1")
    else:
        monkeypatch.delenv("LT_CONV_NO_V6", raising=False)
    if bsrc == "v7_mfma32":
        monkeypatch.delenv("LT_CONV_NO_V7", raising=False)
    else:
        monkeypatch.setenv("LT_CONV_NO_V7", "1")
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 256, 12, 12, generator=g)
    w = torch.randn(256, 256, 4, 4, generator=g) * (1.0 / (256 * 4) ** 0.5)
    bn = _bn(256, g)
    ref = torch.relu(_bn_ref(F.conv_transpose2d(bf16_round(x), bf16_round(w), None, 2, 1), bn))
    out = run_conv(x, w, None, bn, 2, 1, torch.bfloat16, H.TILE3_288, transposed=True, relu=True)
    check("deconv4x4_288x256/%s" % bsrc, out, ref, 1.5e-2)
    gen = run_conv(x, w, None, bn, 2, 1, torch.bfloat16, TILES["v2_128x128"], transposed=True, relu=True)
    check("deconv4x4_288x256/%s vs igemm2" % bsrc, out, gen, 1.5e-2)


@pytest.mark.parametrize("N,H_", [(64, 24), (128, 24), (32, 48)])
def test_conv2d_3x3_256_256_layer3(N, H_):
    """3x3 / stride 1 / pad 1, 256 -> 256 on 24-wide maps (ResNet-152 layer3 at 384^2 inputs) through the default dispatch vs torch:
    ReLU, residual + ReLU.  (A dedicated row-band kernel for this layer, opt-in with LT_CONV_BAND=1, was 8 % faster per layer on dense
    data and 1 % slower inside the forward: removed in round 2.)"""
    g = torch.Generator().manual_seed(N + H_)
    x = torch.randn(N, 256, H_, 24, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) * (1.0 / (256 * 9) ** 0.5)
    bn = _bn(256, g)
    res = torch.randn(N, 256, H_, 24, generator=g)
    rd = bf16_round
    conv = _bn_ref(F.conv2d(rd(x), rd(w), None, 1, 1), bn)
    out = run_conv(x, w, None, bn, 1, 1, torch.bfloat16, 0, relu=True)
    check("conv2d_3x3_256/N%d_H%d/relu" % (N, H_), out, torch.relu(conv), 1.5e-2)
    out2 = run_conv(x, w, None, bn, 1, 1, torch.bfloat16, 0, relu=True, residual=res)
    check("conv2d_3x3_256/N%d_H%d/res" % (N, H_), out2, torch.relu(conv + rd(res)), 1.5e-2)


@pytest.mark.parametrize("th", ["8", "4"])
@pytest.mark.parametrize("N,Hh", [(1, 8), (3, 24), (9, 16), (16, 24)])
def test_conv2d_halo_3x3_256(N, Hh, th, monkeypatch):
    """conv2d_halo_kernel (round 5): the 3x3 256 -> 256 convolution of ResNet layer3's bottlenecks (pose_resnet.py:75-95, conv2 + bn2 + relu) on 24-wide maps with
    the input halo resident in LDS, 8 x 24 and 4 x 24 pixel tiles, vs torch on bf16-rounded operands and vs the implicit-GEMM kernel it replaces
    (LT_CONV_NO_H2D=1).  One-tile images (every halo row outside), three tiles per image (halo rows shared across tiles), odd image counts (XCD remap)."""
    monkeypatch.setenv("LT_H2D_ANY_SIZE", "1")
    monkeypatch.setenv("LT_H2D_TH", th)
    g = torch.Generator().manual_seed(4200 + N + Hh)
    x = torch.relu(torch.randn(N, 256, Hh, 24, generator=g))
    w = torch.randn(256, 256, 3, 3, generator=g) / (9 * 256) ** 0.5
    bn = _bn(256, g)
    rd = bf16_round
    ref = torch.relu(_bn_ref(F.conv2d(rd(x), rd(w), None, 1, 1), bn))

    def run(halo):
        if halo:
            monkeypatch.delenv("LT_CONV_NO_H2D", raising=False)
        else:
            monkeypatch.setenv("LT_CONV_NO_H2D", "1")
        b = E.PlanBuilder(DEV, torch.bfloat16)
        y = b.conv(E.Act(to_cl(x, None, torch.bfloat16)), w, None, bn, stride=1, pad=1, relu=True)
        assert b.last_info["desc"].phase[0].weight_frag_layout == (2 if halo else 3)     # layout 2 = the halo kernel's fragment order: the dispatcher follows it
        plan = b.finish()
        plan.run_eager(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return from_cl(y.t, 2)
    yh = run(True)
    name = "conv2d_halo/N%d_H%d/th%s" % (N, Hh, th)
    check(name + "/vs_torch", yh, ref, 1.5e-2)
    yi = run(False)
    check(name + "/vs_implicit_gemm", yh, yi, 1.5e-2)
    rms = float((yh.float() - yi.float()).pow(2).mean().sqrt() / yi.float().pow(2).mean().sqrt())
    record(name + "/rms_vs_implicit_gemm", rms)
    assert rms < 2e-3          # same products in another summation order + one output rounding: a wrong tap / halo row / K block is far above


def test_conv2d_halo_ragged_last_round_runs_as_half_height_tiles(monkeypatch):
    """128 images of 24 x 24 (the 32-sample forward, config 4 at 16 samples): 384 tiles of 8 rows = one round of 256 + 128; the tail runs as 256 tiles of 4 rows
    in a second launch (half a round instead of a whole one).  Same result as the single launch (LT_H2D_NO_TAIL4=1) BIT FOR BIT -- the tile height does not
    change a pixel's summation order -- and as torch on a sample of images."""
    monkeypatch.setenv("LT_H2D_ANY_SIZE", "1")
    monkeypatch.delenv("LT_H2D_TH", raising=False)
    g = torch.Generator().manual_seed(4242)
    N = 128
    x = torch.relu(torch.randn(N, 256, 24, 24, generator=g))
    w = torch.randn(256, 256, 3, 3, generator=g) / (9 * 256) ** 0.5
    bn = _bn(256, g)
    x_cl = to_cl(x, None, torch.bfloat16)

    def run(tail4):
        if tail4:
            monkeypatch.delenv("LT_H2D_NO_TAIL4", raising=False)
        else:
            monkeypatch.setenv("LT_H2D_NO_TAIL4", "1")
        b = E.PlanBuilder(DEV, torch.bfloat16)
        y = b.conv(E.Act(x_cl), w, None, bn, stride=1, pad=1, relu=True)
        assert b.last_info["desc"].phase[0].weight_frag_layout == 2
        b.finish().run_eager(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return y.t.clone()
    ya, yb = run(True), run(False)
    assert torch.equal(ya, yb)
    rd = bf16_round
    for i in (0, 85, 86, 127):          # images on both sides of the split (tile 256 = image 85, rows 8-15)
        ref = torch.relu(_bn_ref(F.conv2d(rd(x[i:i + 1]), rd(w), None, 1, 1), bn))
        check("conv2d_halo/tail4/image%d" % i, from_cl(ya[i:i + 1], 2), ref, 1.5e-2)


@pytest.mark.parametrize("N,Hh,W", [(1, 8, 24), (3, 24, 24), (2, 16, 48), (5, 48, 48)])
def test_deconv4x4_256_halo(N, Hh, W, monkeypatch):
    """conv2d_halo_kernel<8, 4, 4> (round 5): the 4x4 / stride-2 / pad-1 transposed convolutions 256 -> 256 of the deconvolution head (pose_resnet.py:208-233,
    + BatchNorm + ReLU) as four output parities of 2 x 2 taps over ONE input halo, vs torch on bf16-rounded operands and vs the four implicit-GEMM launches it
    replaces (LT_DECONV_NO_H2D=1).  One-tile maps, 24- and 48-wide maps (two column tiles per row), odd image counts."""
    monkeypatch.setenv("LT_H2D_ANY_SIZE", "1")
    g = torch.Generator().manual_seed(4300 + N + Hh + W)
    x = torch.relu(torch.randn(N, 256, Hh, W, generator=g))
    w = torch.randn(256, 256, 4, 4, generator=g) / (4 * 256) ** 0.5
    bn = _bn(256, g)
    rd = bf16_round
    ref = torch.relu(_bn_ref(F.conv_transpose2d(rd(x), rd(w), None, 2, 1), bn))

    def run(halo):
        if halo:
            monkeypatch.delenv("LT_DECONV_NO_H2D", raising=False)
        else:
            monkeypatch.setenv("LT_DECONV_NO_H2D", "1")
        b = E.PlanBuilder(DEV, torch.bfloat16)
        y = b.conv(E.Act(to_cl(x, None, torch.bfloat16)), w, None, bn, stride=2, pad=1, transposed=True, relu=True)
        assert all(b.last_info["desc"].phase[i].weight_frag_layout == (2 if halo else 3) for i in range(4))
        plan = b.finish()
        plan.run_eager(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return from_cl(y.t, 2)
    yh = run(True)
    name = "deconv4x4_halo/N%d_%dx%d" % (N, Hh, W)
    check(name + "/vs_torch", yh, ref, 1.5e-2)
    yi = run(False)
    check(name + "/vs_implicit_gemm", yh, yi, 1.5e-2)
    rms = float((yh.float() - yi.float()).pow(2).mean().sqrt() / yi.float().pow(2).mean().sqrt())
    record(name + "/rms_vs_implicit_gemm", rms)
    assert rms < 2e-3


@pytest.mark.parametrize("wsrc", ["registers", "lds"])
@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 64), (128, 128), (16, 32)])
@pytest.mark.parametrize("N,sp", [(1, (8, 8, 16)), (3, (4, 16, 8)), (8, (8, 16, 16))])
def test_conv3d_halo_64_64_weight_source(N, sp, cin, cout, wsrc, monkeypatch):
    """3^3 64 -> 64 / 32 -> 64 / 128 -> 128 / (round 5) 16 -> 32 bf16: conv3d_halo_wreg_kernel (weights as fragments from global memory, halo-only LDS) and
    the kernels it replaces (LT_HALO_NO_WREG=1: loader-wave halo kernel or implicit GEMM) vs torch: residual + ReLU, affine only;
    padding at every face."""
    if wsrc == "lds":
        monkeypatch.setenv("LT_HALO_NO_WREG", "1")
    else:
        monkeypatch.delenv("LT_HALO_NO_WREG", raising=False)
    g = torch.Generator().manual_seed(N * 31 + sp[0] + cin)
    x = torch.randn(N, cin, *sp, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (cin * 27) ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    bn = _bn(cout, g)
    res = torch.randn(N, cout, *sp, generator=g)
    rd = bf16_round
    conv = F.conv3d(rd(x), rd(w), bias, 1, 1)
    tile = H.TILE_HALO if (wsrc == "registers" or cin != 128) else 0     # no LDS-weight halo kernel for 128 -> 128: implicit GEMM
    out = run_conv(x, w, bias, bn, 1, 1, torch.bfloat16, tile, relu=True, residual=res)
    check("conv3d_halo_%d_%d/%s/N%d/res" % (cin, cout, wsrc, N), out, torch.relu(_bn_ref(conv, bn) + rd(res)), 1.5e-2)
    out2 = run_conv(x, w, bias, bn, 1, 1, torch.bfloat16, tile, relu=False, residual=None)
    check("conv3d_halo_%d_%d/%s/N%d/plain" % (cin, cout, wsrc, N), out2, _bn_ref(conv, bn), 1.5e-2)


@pytest.mark.parametrize("case", ["halo_7x7_32_16", "halo_7x7_32_16_big"])
def test_conv3d_halo7_variants(case):
    """The 7^3 loader-wave kernel (kd-register-blocked; the tap-major kernel it superseded was removed in round 2): residual + ReLU,
    affine only."""
    kdb = "1"
    N, cin, cout, k, sp, dts = HALO_CASES[case]
    g = torch.Generator().manual_seed(len(case) + 3)
    x = torch.randn(N, cin, *sp, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) * (1.0 / (cin * k ** 3) ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    bn = _bn(cout, g)
    res = torch.randn(N, cout, *sp, generator=g)
    rd = bf16_round
    ref = torch.relu(_bn_ref(F.conv3d(rd(x), rd(w), bias, 1, k // 2), bn) + rd(res))
    out = run_conv(x, w, bias, bn, 1, k // 2, torch.bfloat16, H.TILE_HALO, relu=True, residual=res)
    check("conv3d_halo7/kdb=%s/%s/res" % (kdb, case), out, ref, 1.5e-2)
    ref2 = _bn_ref(F.conv3d(rd(x), rd(w), bias, 1, k // 2), bn)
    out2 = run_conv(x, w, bias, bn, 1, k // 2, torch.bfloat16, H.TILE_HALO, relu=False, residual=None)
    check("conv3d_halo7/kdb=%s/%s/plain" % (kdb, case), out2, ref2, 1.5e-2)


@pytest.mark.parametrize("NV,V,dtype,method,cmu", [(4, 32, torch.bfloat16, "softmax", False), (8, 32, torch.bfloat16, "softmax", False),
                                                   (8, 16, torch.bfloat16, "softmax", True), (4, 16, torch.float32, "softmax", False),
                                                   (3, 16, torch.bfloat16, "sum", False), (8, 12, torch.bfloat16, "softmax", False)])
def test_unproject_grid_fused_vs_separate(NV, V, dtype, method, cmu):
    """lt_unproject_grid_fwd (voxel centres computed in registers and written to the returned tensor; quad kernel for bf16 / 32 channels /
    4 or 8 views / softmax / V % 16 == 0, lt_coord_volumes + generic gather inside for everything else) against lt_coord_volumes followed
    by lt_unproject_fwd: coordinates BIT-identical, volumes identical where the same kernel arithmetic runs and within bf16 rounding
    against the oracle; rotated cuboids, the CMU axis permutation, a camera inside the cube."""
    from mvn.utils import op, volumetric
    g = torch.Generator().manual_seed(NV * 100 + V)
    B, hw, C = 2, 24, 32
    K, R, t = synth.ring_cameras(NV, 96, inside=True)
    P = torch.from_numpy(O.resized_projection(K, R, t, (96, 96), (hw, hw))).float()[None].repeat(B, 1, 1, 1).contiguous()
    hm = torch.randn(B, NV, C, hw, hw, generator=g)
    base = torch.randn(B, 3, generator=g).double().numpy() * 100
    thetas = [0.0, 0.7]
    side = 2500.0
    cv_ref = op.build_coord_volumes(base, side, V, thetas=thetas, cmu_transfer=cmu, device=DEV)
    feats = hm.to(DEV).to(dtype).permute(0, 1, 3, 4, 2).contiguous()
    pos = torch.from_numpy((base - side / 2).astype(np.float32)).to(DEV)
    cen = torch.from_numpy(base.astype(np.float32)).to(DEV)
    rot = torch.from_numpy(np.stack([volumetric.get_rotation_matrix((0, 0, 1), th) for th in thetas]).astype(np.float32)).to(DEV).contiguous()
    coords = torch.full((B, V, V, V, 3), float("nan"), device=DEV)
    out = torch.empty(B, V, V, V, C, dtype=dtype, device=DEV)
    step = float(np.float32(side / (V - 1)))
    Pg = P.to(DEV)
    H.check(H.lib().lt_unproject_grid_fwd(H.dtype_code(dtype), feats.data_ptr(), Pg.data_ptr(), pos.data_ptr(), cen.data_ptr(), rot.data_ptr(), step,
                                          int(cmu), coords.data_ptr(), None, out.data_ptr(), B, NV, C, hw, hw, V, H.AGG[method], H.cur_stream()),
            "lt_unproject_grid_fwd")
    assert torch.equal(coords, cv_ref), "coordinates written by the fused kernel differ from lt_coord_volumes"
    sep = op.unproject_heatmaps(hm.to(DEV).to(dtype), P.to(DEV), cv_ref, method)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    check("unproject_grid/NV%d_V%d_%s_%s_cmu%d vs separate" % (NV, V, str(dtype)[6:], method, cmu), out.permute(0, 4, 1, 2, 3).float().cpu(), sep.float().cpu(), tol)
    ref = O.unproject_heatmaps(bf16_round(hm) if dtype == torch.bfloat16 else hm, P, cv_ref.cpu(), method)
    check("unproject_grid/NV%d_V%d_%s_%s_cmu%d vs oracle" % (NV, V, str(dtype)[6:], method, cmu), out.permute(0, 4, 1, 2, 3).float().cpu(), ref, tol)


@pytest.mark.parametrize("N,sp", [(1, (2, 2, 2)), (1, (8, 8, 8)), (5, (4, 4, 4)), (32, (2, 2, 2)), (32, (8, 8, 8))])
def test_conv3d_splitk_tiny_levels(N, sp, monkeypatch):
    """V2V's 3^3 128 -> 128 layers on <= 8^3 voxels (bf16 plans): S tap-group phases of one lt_conv_fwd (fp32 partial sums) +
    lt_splitk_reduce (bias, folded BatchNorm, residual, ReLU) vs torch, and vs the single-launch path (LT_CONV_NO_SPLITK=1)."""
    g = torch.Generator().manual_seed(N * 7 + sp[0])
    x = torch.randn(N, 128, *sp, generator=g)
    w = torch.randn(128, 128, 3, 3, 3, generator=g) * (1.0 / (128 * 27) ** 0.5)
    bias = torch.randn(128, generator=g) * 0.1
    bn = _bn(128, g)
    res = torch.randn(N, 128, *sp, generator=g)
    rd = bf16_round
    ref = torch.relu(_bn_ref(F.conv3d(rd(x), rd(w), bias, 1, 1), bn) + rd(res))
    monkeypatch.delenv("LT_CONV_NO_SPLITK", raising=False)
    b = E.PlanBuilder(DEV, torch.bfloat16)
    y = b.conv(E.Act(to_cl(x, None, torch.bfloat16)), w, bias, bn, stride=1, pad=1, relu=True, residual=E.Act(to_cl(res, None, torch.bfloat16)))
    assert len(b.ops) == 2 and "split-K" in b.ops[0][1]["label"], [m["label"] for _, m in b.ops]
    b.finish().run_eager(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = from_cl(y.t, 3)
    check("conv3d split-K N%d %s/res+relu" % (N, sp), out, ref, 1.5e-2)
    monkeypatch.setenv("LT_CONV_NO_SPLITK", "1")
    single = run_conv(x, w, bias, bn, 1, 1, torch.bfloat16, 0, relu=True, residual=res)
    check("conv3d split-K N%d %s vs the single launch" % (N, sp), out, single, 1e-2)
    monkeypatch.delenv("LT_CONV_NO_SPLITK", raising=False)
    out2 = run_conv(x, w, bias, bn, 1, 1, torch.bfloat16, 0, relu=False, residual=None)      # plain affine epilogue through the same pair
    check("conv3d split-K N%d %s/plain" % (N, sp), out2, _bn_ref(F.conv3d(rd(x), rd(w), bias, 1, 1), bn), 1.5e-2)



def _bneck_layers(C, P, g):
    ws = [torch.randn(P, C, 1, 1, generator=g) / C ** 0.5, torch.randn(P, P, 3, 3, generator=g) / (9 * P) ** 0.5, torch.randn(C, P, 1, 1, generator=g) / P ** 0.5]
    return ws, [_bn(P, g), _bn(P, g), _bn(C, g)]


def _bneck_run(x_cl, ws, bns, fused, monkeypatch):
    """The identity Bottleneck block through the plan builder: one lt_bottleneck_fwd (fused) or three lt_conv_fwd launches."""
    if fused:
        monkeypatch.delenv("LT_NO_BNECK", raising=False)
    else:
        monkeypatch.setenv("LT_NO_BNECK", "1")
    b = E.PlanBuilder(DEV, torch.bfloat16)
    xa = E.Act(x_cl)
    if fused:
        assert b.can_bottleneck(xa, ws, (1, 1, 1))
        y = b.bottleneck(xa, ws, bns)
    else:
        t1 = b.conv(xa, ws[0], None, bns[0], relu=True)
        t2 = b.conv(t1, ws[1], None, bns[1], pad=1, relu=True)
        y = b.conv(t2, ws[2], None, bns[2], relu=True, residual=xa)
    plan = b.finish()
    assert len(plan.ops) == (1 if fused else 3)
    plan.run_eager(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return y.t


@pytest.mark.parametrize("C,P,N,Hh,W", [(256, 64, 1, 8, 16), (256, 64, 3, 24, 32), (512, 128, 1, 8, 16), (512, 128, 2, 16, 48), (256, 64, 9, 16, 16)])
def test_bottleneck_fused(C, P, N, Hh, W, monkeypatch):
    """lt_bottleneck_fwd (pose_resnet.py:75-95, identity block) against (a) torch fp32 on bf16-rounded operands with the two inner tensors
    rounded to bf16 where the separate launches store them, (b) the three lt_conv_fwd launches it replaces.  One-tile maps (every halo pixel
    outside the image), several tiles per image (halo exchange across tile borders, both directions) and an odd image count (XCD remap)."""
    g = torch.Generator().manual_seed(C + N + Hh)
    x = torch.randn(N, C, Hh, W, generator=g)
    ws, bns = _bneck_layers(C, P, g)
    x_cl = to_cl(x, None, torch.bfloat16)
    y = _bneck_run(x_cl, ws, bns, True, monkeypatch)
    assert y.data_ptr() != x_cl.data_ptr() and tuple(y.shape) == (N, 1, Hh, W, C)
    rd = bf16_round
    xr = rd(x)
    t1 = rd(torch.relu(_bn_ref(F.conv2d(xr, rd(ws[0])), bns[0])))
    t2 = rd(torch.relu(_bn_ref(F.conv2d(t1, rd(ws[1]), None, 1, 1), bns[1])))
    ref = torch.relu(_bn_ref(F.conv2d(t2, rd(ws[2])), bns[2]) + xr)
    name = "bneck/%d_%d/%dx%dx%d" % (C, P, N, Hh, W)
    check(name + "/vs_torch", from_cl(y, 2), ref, 1.5e-2)
    y3 = _bneck_run(x_cl, ws, bns, False, monkeypatch)
    check(name + "/vs_three_launches", from_cl(y, 2), from_cl(y3, 2), 1.5e-2)
    # the three launches and the fused kernel agree to bf16 rounding noise: a structural mistake (one tap, one halo row) is far above this
    record(name + "/rms_vs_three_launches", float((y.float() - y3.float()).pow(2).mean().sqrt() / y3.float().pow(2).mean().sqrt()))
    assert float((y.float() - y3.float()).pow(2).mean().sqrt() / y3.float().pow(2).mean().sqrt()) < 2e-3


@pytest.mark.parametrize("N,D,Hh,W", [(4, 16, 64, 64), (16, 8, 64, 32), (1, 64, 128, 128)])
def test_conv3d_with_computed_skip_residual(N, D, Hh, W, monkeypatch):
    """lt_conv_skip_fwd (round 5; v2v.py:20-42, the 16 -> 32 Res3DBlock of :76): relu(bn2(conv3x3x3(y)) + bn_s(conv1x1x1_s(x))) with the skip convolution computed
    inside the second convolution's launch (column-walk halo kernel), against (a) torch fp32 on bf16-rounded operands with the skip branch's BatchNorm scale
    folded into its weights before the ONE bf16 rounding -- what the plan builder does -- and (b) the two launches it replaces (skip convolution stored in bf16,
    then read as the residual: one rounding apart).  Shapes: the smallest the column walk takes, a flat one (two tiles per column), one big sample."""
    g = torch.Generator().manual_seed(8100 + N + D)
    y = torch.relu(torch.randn(N, 32, D, Hh, W, generator=g))
    x = torch.relu(torch.randn(N, 16, D, Hh, W, generator=g))
    w = torch.randn(32, 32, 3, 3, 3, generator=g) / (27 * 32) ** 0.5
    ws = torch.randn(32, 16, 1, 1, 1, generator=g) / 4.0
    bias, bs = torch.randn(32, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1
    bn, bns = _bn(32, g), _bn(32, g)
    ya, xa = E.Act(to_cl(y, None, torch.bfloat16)), E.Act(to_cl(x, None, torch.bfloat16))

    def run(fused):
        if fused:
            monkeypatch.delenv("LT_NO_CONV_SKIP", raising=False)
        else:
            monkeypatch.setenv("LT_NO_CONV_SKIP", "1")
        b = E.PlanBuilder(DEV, torch.bfloat16)
        assert b.can_conv_skip(ya.shape, w, xa.shape, ws) == fused
        if fused:
            z = b.conv(ya, w, bias, bn, pad=1, relu=True, skip=(xa, ws, bs, bns))
        else:
            r = b.conv(xa, ws, bs, bns)
            z = b.conv(ya, w, bias, bn, pad=1, relu=True, residual=r)
        plan = b.finish()
        assert len(plan.ops) == (1 if fused else 2)
        plan.run_eager(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return from_cl(z.t, 3)
    zf = run(True)
    rd = bf16_round
    sc_s = bns[0] / torch.sqrt(bns[3] + 1e-5)
    ws_fold = rd(ws * sc_s.view(-1, 1, 1, 1, 1))
    skip_ref = F.conv3d(rd(x), ws_fold) + (bs * sc_s + bns[1] - bns[2] * sc_s).view(1, -1, 1, 1, 1)
    ref = torch.relu(_bn_ref(F.conv3d(rd(y), rd(w), bias, 1, 1), bn) + skip_ref)
    name = "conv_skip/%dx%dx%dx%d" % (N, D, Hh, W)
    check(name + "/vs_torch", zf, ref, 1.5e-2)
    z2 = run(False)
    check(name + "/vs_two_launches", zf, z2, 2e-2)
    rms = float((zf.float() - z2.float()).pow(2).mean().sqrt() / z2.float().pow(2).mean().sqrt())
    record(name + "/rms_vs_two_launches", rms)
    assert rms < 5e-3          # one tap / one K half / one fragment of the skip wrong is far above the branch's bf16 rounding
    b = E.PlanBuilder(DEV, torch.bfloat16)   # shapes the column walk does not take keep the two launches: the builder must say so
    assert not b.can_conv_skip((1, 16, 32, 32, 32), w, (1, 16, 32, 32, 16), ws) and not b.can_conv_skip((4, 4, 64, 64, 32), w, (4, 4, 64, 64, 16), ws)


@pytest.mark.parametrize("N,Ho,Wo,P,Cin2,Cc,st", [(2, 6, 10, 128, 256, 512, 2), (3, 24, 24, 256, 512, 1024, 2), (2, 12, 12, 512, 1024, 2048, 2), (1, 16, 32, 64, 64, 256, 1),
                                                   (5, 13, 7, 128, 256, 512, 2)])
def test_expand_plus_downsample_as_one_pointwise_convolution(N, Ho, Wo, P, Cin2, Cc, st, monkeypatch):
    """lt_conv_cat2_fwd (round 5; pose_resnet.py:75-95 with the stride-2 `downsample` of :196-206 -- the first blocks of layer2 / 3 / 4, and the stride-1 one of
    layer1): relu(bn3(conv1x1(t2)) + bn_d(conv1x1_d(x), stride s)) as one pointwise convolution over [t2 | x at the strided pixels], against (a) torch fp32 on
    bf16-rounded operands with both BatchNorm scales folded into the weights before their ONE bf16 rounding (what the plan builder does) and (b) the two
    launches it replaces.  Ragged tiles (120 / 455 rows), whole tiles, every layer's widths."""
    monkeypatch.setenv("LT_CAT2_ANY_SIZE", "1")      # the builder only takes this path from ~200 tiles on
    g = torch.Generator().manual_seed(9100 + N + Ho + P)
    t2 = torch.relu(torch.randn(N, P, Ho, Wo, generator=g))
    x = torch.relu(torch.randn(N, Cin2, Ho * st, Wo * st, generator=g))
    w3, wd = torch.randn(Cc, P, 1, 1, generator=g) / P ** 0.5, torch.randn(Cc, Cin2, 1, 1, generator=g) / Cin2 ** 0.5
    bn3, bnd = _bn(Cc, g), _bn(Cc, g)
    ta, xa = E.Act(to_cl(t2, None, torch.bfloat16)), E.Act(to_cl(x, None, torch.bfloat16))

    def run(fused):
        b = E.PlanBuilder(DEV, torch.bfloat16)
        if fused:
            assert b.can_conv_cat2(ta.shape, w3, xa.shape, wd, st)
            y = b.conv_cat2(ta, w3, bn3, xa, wd, bnd, st)
        else:
            r = b.conv(xa, wd, None, bnd, stride=st)
            y = b.conv(ta, w3, None, bn3, relu=True, residual=r)
        plan = b.finish()
        assert len(plan.ops) == (1 if fused else 2)
        plan.run_eager(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return from_cl(y.t, 2)
    yf = run(True)
    rd = bf16_round
    s3, sd = bn3[0] / torch.sqrt(bn3[3] + 1e-5), bnd[0] / torch.sqrt(bnd[3] + 1e-5)
    ref = torch.relu(F.conv2d(rd(t2), rd(w3 * s3.view(-1, 1, 1, 1))) + F.conv2d(rd(x), rd(wd * sd.view(-1, 1, 1, 1)), None, st)
                     + ((bn3[1] - bn3[2] * s3) + (bnd[1] - bnd[2] * sd)).view(1, -1, 1, 1))
    name = "conv_cat2/%dx%dx%d/%d+%d_%d/s%d" % (N, Ho, Wo, P, Cin2, Cc, st)
    check(name + "/vs_torch", yf, ref, 1.5e-2)
    y2 = run(False)
    check(name + "/vs_two_launches", yf, y2, 2.5e-2)
    rms = float((yf.float() - y2.float()).pow(2).mean().sqrt() / y2.float().pow(2).mean().sqrt())
    record(name + "/rms_vs_two_launches", rms)
    assert rms < 6e-3          # a wrong pixel map / K split / weight block is far above the rounding of the branch and of the folded weights


def test_conv_skip_refuses_what_it_does_not_cover():
    """lt_conv_skip_fwd has no fallback kernel: a shape outside the column walk (4 columns of tiles) or another skip width must come back as
    LT_ERR_UNSUPPORTED (-2, include/lt_hip.h), not as a silently different path."""
    import ctypes as C
    d = H.ConvDesc()
    d.dtype = H.LT_BF16
    d.N, d.D, d.H, d.W, d.Cin = 1, 8, 16, 16, 32
    d.Do, d.Ho, d.Wo = 8, 16, 16
    d.stride = H.i3((1, 1, 1)); d.pad = H.i3((1, 1, 1)); d.OD, d.OH, d.OW = 8, 16, 16; d.out_stride = H.i3((1, 1, 1))
    d.Cout, d.ldc, d.cout_pad, d.k_pad = 32, 32, 32, 896          # 27 taps x 32 channels, padded to whole 128-byte K steps
    d.nphase, d.flags, d.tile, d.stages = 1, H.EPI_RELU_POST, 0, 0
    wdev = torch.zeros(32, 896, dtype=torch.bfloat16, device=DEV)
    taps = torch.tensor([(a, bb, c, ((a * 16 + bb) * 16 + c) * 32) for a in range(3) for bb in range(3) for c in range(3)], dtype=torch.int32, device=DEV)
    d.phase[0].weight, d.phase[0].taps, d.phase[0].ntaps = wdev.data_ptr(), taps.data_ptr(), 27
    xin = torch.zeros(1, 8, 16, 16, 32, dtype=torch.bfloat16, device=DEV)
    sx = torch.zeros(1, 8, 16, 16, 16, dtype=torch.bfloat16, device=DEV)
    wf = torch.zeros(512, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros(1, 8, 16, 16, 32, dtype=torch.bfloat16, device=DEV)
    zero = torch.zeros(32, device=DEV); one = torch.ones(32, device=DEV)
    for cin in (16, 32):
        sk = H.ConvSkip()
        sk.x, sk.cin, sk.weight_frag = sx.data_ptr(), cin, wf.data_ptr()
        rc = H.lib().lt_conv_skip_fwd(C.byref(d), xin.data_ptr(), zero.data_ptr(), one.data_ptr(), zero.data_ptr(), C.byref(sk), y.data_ptr(), H.cur_stream())
        assert rc == -2, rc
    torch.cuda.synchronize()


@pytest.mark.parametrize("N,Hh,W", [(1, 8, 16), (3, 24, 32), (2, 16, 48), (9, 16, 16), (4, 96, 96)])
def test_bottleneck_with_downsample_fused(N, Hh, W, monkeypatch):
    """lt_bottleneck_ds_fwd (round 5; pose_resnet.py:75-95 with the `downsample` branch of :196-206: the first block of ResNet layer1, 64 -> 64 -> 256) against
    (a) torch fp32 on bf16-rounded operands with the two inner tensors rounded to bf16 where the separate launches store them and the downsample branch
    NOT rounded (the kernel adds it in fp32), (b) the four lt_conv_fwd launches it replaces (which do round the branch: one bf16 rounding apart).
    One-tile maps (every halo pixel outside the image), several tiles per image, an odd image count (XCD remap), the benchmark's 96 x 96 map."""
    Cin, P, C_ = 64, 64, 256
    g = torch.Generator().manual_seed(7000 + N + Hh)
    x = torch.relu(torch.randn(N, Cin, Hh, W, generator=g))       # the stem's output is a ReLU / max-pool of one
    ws = [torch.randn(P, Cin, 1, 1, generator=g) / Cin ** 0.5, torch.randn(P, P, 3, 3, generator=g) / (9 * P) ** 0.5, torch.randn(C_, P, 1, 1, generator=g) / P ** 0.5]
    wd = torch.randn(C_, Cin, 1, 1, generator=g) / Cin ** 0.5
    bns, bnd = [_bn(P, g), _bn(P, g), _bn(C_, g)], _bn(C_, g)
    x_cl = to_cl(x, None, torch.bfloat16)

    def run(fused):
        if fused:
            monkeypatch.delenv("LT_NO_BNECK_DS", raising=False)
        else:
            monkeypatch.setenv("LT_NO_BNECK_DS", "1")
        b = E.PlanBuilder(DEV, torch.bfloat16)
        xa = E.Act(x_cl)
        if fused:
            assert b.can_bottleneck_ds(xa, ws, (1, 1, 1), wd, 1)
            y = b.bottleneck_ds(xa, ws, bns, wd, bnd)
        else:
            assert not b.can_bottleneck_ds(xa, ws, (1, 1, 1), wd, 1)
            r = b.conv(xa, wd, None, bnd)
            t1 = b.conv(xa, ws[0], None, bns[0], relu=True)
            t2 = b.conv(t1, ws[1], None, bns[1], pad=1, relu=True)
            y = b.conv(t2, ws[2], None, bns[2], relu=True, residual=r)
        plan = b.finish()
        assert len(plan.ops) == (1 if fused else 4)
        plan.run_eager(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return y.t
    y = run(True)
    assert y.data_ptr() != x_cl.data_ptr() and tuple(y.shape) == (N, 1, Hh, W, C_)
    rd = bf16_round
    xr = rd(x)
    t1 = rd(torch.relu(_bn_ref(F.conv2d(xr, rd(ws[0])), bns[0])))
    t2 = rd(torch.relu(_bn_ref(F.conv2d(t1, rd(ws[1]), None, 1, 1), bns[1])))
    ref = torch.relu(_bn_ref(F.conv2d(t2, rd(ws[2])), bns[2]) + _bn_ref(F.conv2d(xr, rd(wd)), bnd))
    name = "bneck_ds/%dx%dx%d" % (N, Hh, W)
    check(name + "/vs_torch", from_cl(y, 2), ref, 1.5e-2)
    y4 = run(False)
    check(name + "/vs_four_launches", from_cl(y, 2), from_cl(y4, 2), 1.5e-2)
    rms = float((y.float() - y4.float()).pow(2).mean().sqrt() / y4.float().pow(2).mean().sqrt())
    record(name + "/rms_vs_four_launches", rms)
    assert rms < 4e-3           # a structural mistake (one tap, one halo row, one K block of the branch) is far above the branch's bf16 rounding
    e_f = float((from_cl(y, 2).float() - ref).pow(2).mean().sqrt()); e_4 = float((from_cl(y4, 2).float() - ref).pow(2).mean().sqrt())
    record(name + "/rms_err_vs_torch fused | four launches", [e_f, e_4])
    assert e_f <= e_4 * 1.05 + 1e-6   # adding the branch unrounded cannot be worse than rounding it first


@pytest.mark.parametrize("npb", ["3", "2", "1"])
@pytest.mark.parametrize("N,Hh,W", [(2, 6, 6), (2, 24, 24), (8, 24, 24), (3, 12, 20)])
def test_expand_reduce_seam_fused(N, Hh, W, npb, monkeypatch):
    """lt_expand_reduce_fwd (round 5; pose_resnet.py:75-95, the seam between two identity blocks of layer3: expand + bn3 + residual + ReLU of block i, reduce +
    bn1 + ReLU of block i + 1) against (a) torch fp32 on bf16-rounded operands with y rounded to bf16 where the expand stores it, (b) the two lt_conv_fwd
    launches it replaces.  72 rows (one ragged 128-row tile), 1152 and 4608 rows (whole tiles, several per XCD), 720 rows (5.6 tiles: ragged tail, odd count)."""
    C_, P = 1024, 256
    g = torch.Generator().manual_seed(N * 1000 + Hh)
    t2 = torch.relu(torch.randn(N, P, Hh, W, generator=g))
    res = torch.relu(torch.randn(N, C_, Hh, W, generator=g))
    w3, w1 = torch.randn(C_, P, 1, 1, generator=g) / P ** 0.5, torch.randn(P, C_, 1, 1, generator=g) / C_ ** 0.5
    bn3, bn1 = _bn(C_, g), _bn(P, g)
    t2_cl, res_cl = to_cl(t2, None, torch.bfloat16), to_cl(res, None, torch.bfloat16)

    monkeypatch.setenv("LT_XR_ANY_SIZE", "1")          # the plan builder only picks the seam kernel from 64 tiles on; the kernel itself takes any row count
    monkeypatch.setenv("LT_XR_NPB", npb)               # tiles of 96 / 64 / 32 pixels (the launcher picks by row count: these shapes would all get 32)

    def run(fused):
        if fused:
            monkeypatch.delenv("LT_NO_XR", raising=False)
        else:
            monkeypatch.setenv("LT_NO_XR", "1")
        b = E.PlanBuilder(DEV, torch.bfloat16)
        ta, ra = E.Act(t2_cl), E.Act(res_cl)
        if fused:
            assert b.can_expand_reduce(ta, ra, w3, w1)
            y, t1 = b.expand_reduce(ta, ra, w3, bn3, w1, bn1)
        else:
            assert not b.can_expand_reduce(ta, ra, w3, w1)
            y = b.conv(ta, w3, None, bn3, relu=True, residual=ra)
            t1 = b.conv(y, w1, None, bn1, relu=True)
        plan = b.finish()
        assert len(plan.ops) == (1 if fused else 2)
        plan.run_eager(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return y.t, t1.t
    y, t1 = run(True)
    rd = bf16_round
    y_ref = rd(torch.relu(_bn_ref(F.conv2d(rd(t2), rd(w3)), bn3) + rd(res)))
    t1_ref = torch.relu(_bn_ref(F.conv2d(y_ref, rd(w1)), bn1))
    name = "xr/%dx%dx%d" % (N, Hh, W)
    check(name + "/y vs_torch", from_cl(y, 2), y_ref, 1.5e-2)
    check(name + "/t1 vs_torch", from_cl(t1, 2), t1_ref, 1.5e-2)
    y2, t12 = run(False)
    monkeypatch.delenv("LT_NO_XR", raising=False)
    for nm, a_, b_ in (("y", y, y2), ("t1", t1, t12)):
        rms = float((a_.float() - b_.float()).pow(2).mean().sqrt() / b_.float().pow(2).mean().sqrt())
        record(name + "/%s rms_vs_two_launches" % nm, rms)
        check(name + "/%s vs_two_launches" % nm, from_cl(a_, 2), from_cl(b_, 2), 1.5e-2)
        assert rms < 2e-3, (nm, rms)          # bf16 rounding noise; a structural mistake (one K block, one channel run) is far above this


@pytest.mark.parametrize("kind", ["col_walk_residual", "deconv2_residual", "skip_conv"])
def test_conv_beyond_32_bit_element_offsets(kind):
    """Round 6 ("64-bit element offsets"): lt_conv_fwd / lt_conv_skip_fwd over tensors of >= 2^31 ELEMENTS -- BASELINE config 4 at 32 samples
    per GPU: 32 x 128^3 voxels x 32 channels = 2^31 exactly -- in ONE call.  The kernels index a launch with 32-bit offsets; the entry point walks the batch in
    sample chunks (lt_conv_chunk_samples: here 2 x 16) with every per-sample pointer -- input, output, residual, the skip source -- advanced in 64 bits.
    Checked against the same layer run by hand on the two halves (bit-identical: same kernel, same tiles), and a spot check of the last sample against
    torch on the bf16-rounded operands (a wrong base pointer would read another sample)."""
    N, V = 32, 128
    g = torch.Generator().manual_seed(77)
    bf = torch.bfloat16
    lib = H.lib()
    assert lib.lt_conv_chunk_samples(N, V ** 3 * 32) == 16

    def big(shape):          # random bf16 tensor on the device without a 4-byte staging copy of the whole thing
        t = torch.empty(shape, dtype=bf, device=DEV)
        for n in range(shape[0]):
            t[n] = torch.randn(shape[1:], generator=g).to(bf).to(DEV) if n in (0, 15, 16, 31) else t[n - 1].roll(1, 0)
        return t
    if kind == "deconv2_residual":       # V2V's last upsampling: 64 -> 32 channels, 64^3 -> 128^3, the skip tensor added in the epilogue (2^31 output elements)
        x = big((N, V // 2, V // 2, V // 2, 64))
        w = torch.randn(64, 32, 2, 2, 2, generator=g) * 0.05
        kw = dict(stride=2, pad=0, transposed=True, relu=True)
        oshape = (N, V, V, V, 32)
    else:                                # 3x3x3 32 -> 32 at 128^3 (input AND output 2^31 elements), residual or computed skip branch
        x = big((N, V, V, V, 32))
        w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.03
        kw = dict(stride=1, pad=1, relu=True)
        oshape = (N, V, V, V, 32)
    bn = _bn(32, g)
    res = big(oshape) if kind != "skip_conv" else None
    sx = big((N, V, V, V, 16)) if kind == "skip_conv" else None
    sw, sbn = torch.randn(32, 16, 1, 1, 1, generator=g) * 0.1, _bn(32, g)
    assert x.numel() >= 2 ** 31 or oshape[0] * oshape[1] * oshape[2] * oshape[3] * oshape[4] >= 2 ** 31

    def run(sl):
        b = E.PlanBuilder(DEV, bf)
        xa = E.Act(x[sl])
        if kind == "skip_conv":
            sa = E.Act(sx[sl])
            assert b.can_conv_skip(xa.shape, w, sa.shape, sw)
            y = b.conv(xa, w, None, bn, skip=(sa, sw, None, sbn), **kw)
        else:
            y = b.conv(xa, w, None, bn, residual=E.Act(res[sl]), **kw)
        b.finish().run_eager(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return y.t
    y = run(slice(0, N))
    for lo in (0, 16):
        yh = run(slice(lo, lo + 16))
        assert torch.equal(y[lo:lo + 16].view(torch.int16), yh.view(torch.int16)), "one call over 2^31 elements != the two halves (%s, samples %d..)" % (kind, lo)
        del yh
    # the last sample against torch (fp32 on the bf16-rounded operands), one 32^3 corner of it
    n = N - 1
    xs = x[n].float().cpu().permute(3, 0, 1, 2)[None]
    if kind == "deconv2_residual":
        ref = F.conv_transpose3d(xs[:, :, :16, :16, :16], bf16_round(w), stride=2)
        ref = torch.relu(_bn_ref(ref, bn) + res[n, :32, :32, :32].float().cpu().permute(3, 0, 1, 2)[None])
        got = y[n, :32, :32, :32].float().cpu().permute(3, 0, 1, 2)[None]
        check("conv > 2^31 elements/%s last sample vs torch" % kind, got, ref, 1.5e-2)
    else:
        ref = _bn_ref(F.conv3d(xs[:, :, :34, :34, :34], bf16_round(w), padding=1), bn)[:, :, :32, :32, :32]
        if kind == "skip_conv":
            ref = ref + _bn_ref(F.conv3d(sx[n, :32, :32, :32].float().cpu().permute(3, 0, 1, 2)[None], bf16_round(sw)), sbn)
        else:
            ref = ref + res[n, :32, :32, :32].float().cpu().permute(3, 0, 1, 2)[None]
        # (rows / planes 0..31 of the corner: their 3^3 windows lie inside the 34^3 crop except at the volume's own border, which both sides zero-pad)
        got = y[n, :32, :32, :32].float().cpu().permute(3, 0, 1, 2)[None]
        check("conv > 2^31 elements/%s last sample vs torch" % kind, got[:, :, :32, :32, :32], torch.relu(ref), 1.5e-2)


@pytest.mark.parametrize("N,sp", [(1, (8, 8, 16)), (3, (12, 16, 24))])
def test_conv3d_halo_7x7x7_16_to_32_the_front_layers_input_gradient(N, sp, monkeypatch):
    """Round 6: 7^3 16 -> 32 (the INPUT GRADIENT of V2V's front layer in the 16-bit training step, v2v.py:146: the flipped / transposed filter of the 32 -> 16
    convolution) on the halo kernel instead of the generic 256 x 32 implicit-GEMM tile (four taps per K step: 1.78 ms at 8 samples).  Against torch on the
    bf16-rounded operands, with and without a residual (the input gradient that is already there rides in the epilogue), and against the generic tile."""
    g = torch.Generator().manual_seed(N * 7 + sp[0])
    cin, cout, k = 16, 32, 7
    x = torch.randn(N, cin, *sp, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) * (1.0 / (cin * k ** 3) ** 0.5)
    res = torch.randn(N, cout, *sp, generator=g)
    rd = bf16_round
    conv = F.conv3d(rd(x), rd(w), None, 1, 3)
    out = run_conv(x, w, None, None, 1, 3, torch.bfloat16, H.TILE_HALO, residual=res)          # LT_TILE_HALO: fails loudly if no halo kernel takes the shape
    check("conv3d_halo 7^3 16->32/N%d/res" % N, out, conv + rd(res), 1.5e-2)
    out2 = run_conv(x, w, None, None, 1, 3, torch.bfloat16, 0)
    check("conv3d_halo 7^3 16->32/N%d/plain" % N, out2, conv, 1.5e-2)
    monkeypatch.setenv("LT_HALO_NO_D7", "1")
    out3 = run_conv(x, w, None, None, 1, 3, torch.bfloat16, 0)
    rms = float((out2 - out3).pow(2).mean().sqrt() / out3.pow(2).mean().sqrt())
    record("conv3d_halo 7^3 16->32/N%d rms vs the generic tile" % N, rms)
    assert rms < 2e-3, rms
