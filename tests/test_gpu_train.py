"""GPU (-m gpu): the training step (BASELINE config 5).
  * the training kernels of csrc/train.hip one by one against torch-CPU autograd of the same op (fp64 reference);
  * single layers through lt_train.TrainTape (forward in training mode + input / weight / bias / BatchNorm gradients);
  * ONE WHOLE STEP of VolumetricTriangulationNet -- train-mode forward, MAE + 0.01 CE loss, backward, Adam -- against the step the
    REFERENCE ITSELF takes on CPU (tests/golden/train_step.npz, oracle/make_golden.py ``train``).
Gate for gradients: max|d| <= tol * max|ref| per tensor (fp32 kernels)."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import check, record, to_cl, from_cl
from oracle import spec, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    import lt_hip as H
    return H, H.lib()


def _st():
    return torch.cuda.current_stream(DEV).cuda_stream


@pytest.mark.parametrize("flags_name,with_res", [("none", False), ("relu", False), ("relu", True), ("relu_pre", True)])
@pytest.mark.parametrize("rows,C", [(1000, 64), (37, 16), (70000, 32)])
def test_bn_act_fwd_bwd(rows, C, flags_name, with_res):
    H, lib = _lib()
    g = torch.Generator().manual_seed(rows + C)
    y = torch.randn(rows, C, generator=g) * 2 + 0.5
    res = torch.randn(rows, C, generator=g) if with_res else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    dz = torch.randn(rows, C, generator=g)
    flags = {"none": 0, "relu": H.EPI_RELU_POST, "relu_pre": H.EPI_RELU_PRE}[flags_name]
    # reference (fp64 autograd)
    yd = y.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    rd = res.double().requires_grad_(True) if with_res else None
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    n = (yd - mean) / torch.sqrt(var + 1e-5) * gd + bd
    if flags_name == "relu_pre":
        dz = dz * (n.detach().abs() > 1e-4).float()      # no upstream gradient where fp32 rounding could flip the ReLU mask
        n = F.relu(n)
    if with_res:
        n = n + rd
    if flags_name == "relu":
        dz = dz * (n.detach().abs() > 1e-4).float()
        n = F.relu(n)
    (n * dz.double()).sum().backward()
    # ours
    yg, dzg = y.to(DEV), dz.to(DEV)
    resg = res.to(DEV) if with_res else None
    mg, vg = mean.detach().float().to(DEV), var.detach().float().to(DEV)
    gg, bg = gamma.to(DEV), beta.to(DEV)
    z = torch.empty_like(yg)
    z16 = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV)
    H.check(lib.lt_bn_act_fwd(yg.data_ptr(), mg.data_ptr(), vg.data_ptr(), gg.data_ptr(), bg.data_ptr(), H.ptr(resg), z.data_ptr(), z16.data_ptr(), rows, C, 1e-5, flags, _st()), "fwd")
    assert torch.equal(z16, z.bfloat16()), "the bf16 copy is the rounded fp32 result"
    tag = "train/bn_act %dx%d %s%s" % (rows, C, flags_name, "+res" if with_res else "")
    check(tag + " fwd", z.cpu(), n.detach(), 2e-6)
    dy, dga, dbe = torch.empty_like(yg), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dres = torch.full_like(yg, 3.0) if with_res else None
    ws = torch.empty(max(1, lib.lt_bn_act_bwd_workspace(rows, C)), dtype=torch.uint8, device=DEV)
    dy16 = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV)
    H.check(lib.lt_bn_act_bwd(dzg.data_ptr(), yg.data_ptr(), H.ptr(resg), mg.data_ptr(), vg.data_ptr(), gg.data_ptr(), bg.data_ptr(), dy.data_ptr(), dy16.data_ptr(),
                              dga.data_ptr(), dbe.data_ptr(), H.ptr(dres), 1 if with_res else 0, rows, C, 1e-5, flags, ws.data_ptr(), _st()), "bwd")
    assert torch.equal(dy16, dy.bfloat16())
    check(tag + " dy", dy.cpu(), yd.grad, 2e-5)
    check(tag + " dgamma", dga.cpu(), gd.grad, 2e-5)
    check(tag + " dbeta", dbe.cpu(), bd.grad, 2e-5)
    if with_res:
        check(tag + " dres (accumulated onto 3)", dres.cpu(), rd.grad + 3.0, 2e-6)


@pytest.mark.parametrize("flags_name", ["none", "relu", "relu_pre"])
def test_act_bwd_and_channel_sum(flags_name):
    H, lib = _lib()
    g = torch.Generator().manual_seed(5)
    rows, C = 4099, 17
    pre = torch.randn(rows, C, generator=g); res = torch.randn(rows, C, generator=g); dz = torch.randn(rows, C, generator=g)
    flags = {"none": 0, "relu": H.EPI_RELU_POST, "relu_pre": H.EPI_RELU_PRE}[flags_name]
    pd, rd = pre.double().requires_grad_(True), res.double().requires_grad_(True)
    n = pd
    if flags_name == "relu_pre":
        # z = relu(v) alone: the mask is z > 0.  With a residual BEHIND the ReLU the sign of v cannot be rebuilt from relu(v) + res
        # ((z - res) > 0 loses live gradients below half an ulp of res): the entry point refuses that combination
        n = F.relu(n)
        (n * dz.double()).sum().backward()
        zg, dzg, resg = n.detach().float().to(DEV), dz.to(DEV), res.to(DEV)
        dy, dres = torch.empty_like(zg), torch.empty_like(zg)
        H.check(lib.lt_act_bwd(dzg.data_ptr(), zg.data_ptr(), None, dy.data_ptr(), None, 0, rows * C, flags, _st()), "lt_act_bwd")
        check("train/act_bwd relu_pre dy", dy.cpu(), pd.grad, 1e-6)
        assert lib.lt_act_bwd(dzg.data_ptr(), zg.data_ptr(), resg.data_ptr(), dy.data_ptr(), dres.data_ptr(), 0, rows * C, flags, _st()) == -2
        assert b"residual" in lib.lt_last_error()
    else:
        n = n + rd
        if flags_name == "relu":
            n = F.relu(n)
        (n * dz.double()).sum().backward()
        zg, dzg, resg = n.detach().float().to(DEV), dz.to(DEV), res.to(DEV)
        dy, dres = torch.empty_like(zg), torch.empty_like(zg)
        H.check(lib.lt_act_bwd(dzg.data_ptr(), zg.data_ptr(), resg.data_ptr(), dy.data_ptr(), dres.data_ptr(), 0, rows * C, flags, _st()), "lt_act_bwd")
        check("train/act_bwd %s dy" % flags_name, dy.cpu(), pd.grad, 1e-6)
        check("train/act_bwd %s dres" % flags_name, dres.cpu(), rd.grad, 1e-6)
    out = torch.empty(C, device=DEV)
    ws = torch.empty(max(1, lib.lt_channel_sum_workspace(rows, C)), dtype=torch.uint8, device=DEV)
    H.check(lib.lt_channel_sum(dzg.data_ptr(), rows, C, out.data_ptr(), 0, ws.data_ptr(), _st()), "lt_channel_sum")
    check("train/channel_sum", out.cpu(), dz.double().sum(0), 1e-6)


@pytest.mark.parametrize("nd,k,s,p,shape", [(2, 3, 2, 1, (3, 16, 20, 24)), (3, 2, 2, 0, (2, 32, 8, 8, 8)), (2, 3, 2, 1, (2, 64, 15, 17))])
def test_maxpool_bwd(nd, k, s, p, shape):
    H, lib = _lib()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(shape, generator=g)
    x = F.relu(x)        # ties at zero, as behind the stem's ReLU
    xd = x.double().requires_grad_(True)
    y = (F.max_pool2d if nd == 2 else F.max_pool3d)(xd, k, s, p)
    dy = torch.randn(y.shape, generator=g)
    (y * dy.double()).sum().backward()
    xg, dyg = to_cl(x), to_cl(dy)
    dx = torch.zeros_like(xg)
    N, D, Hh, W, C = xg.shape
    kk, ss, pp = ((1, k, k), (1, s, s), (0, p, p)) if nd == 2 else ((k,) * 3, (s,) * 3, (p,) * 3)
    H.check(lib.lt_maxpool_bwd(xg.data_ptr(), dyg.data_ptr(), dx.data_ptr(), N, D, Hh, W, C, H.i3(kk), H.i3(ss), H.i3(pp), _st()), "lt_maxpool_bwd")
    ref = xd.grad
    ours = from_cl(dx, nd)
    # where the input is exactly zero the gradient dies in the ReLU backward that follows: compare only where it matters, and the total
    m = (x > 0)
    check("train/maxpool_bwd nd=%d k=%d %s (x > 0)" % (nd, k, "x".join(map(str, shape))), ours * m, ref * m, 1e-6)
    assert abs(float(ours.double().sum()) - float(ref.sum())) <= 1e-3 * float(ref.abs().sum())


def _bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("flags_name,with_res", [("none", False), ("relu", True), ("relu_pre", True)])
@pytest.mark.parametrize("rows,C", [(1000, 64), (70000, 32), (37, 16)])
def test_bn_act_fwd_bwd_bf16_activations(rows, C, flags_name, with_res):
    """LT_ACT_BF16 (+ LT_BN_Y_BF16): the 16-bit-activation step's BatchNorm passes -- y, residual, z, dz, dy, dres are bf16 tensors.  Reference: fp64
    autograd over the bf16-ROUNDED inputs; outputs compared after their own rounding to bf16 (2^-8 relative), the fp32 sums at fp32 tolerances."""
    H, lib = _lib()
    g = torch.Generator().manual_seed(rows + C + 1)
    y = _bf(torch.randn(rows, C, generator=g) * 2 + 0.5)
    res = _bf(torch.randn(rows, C, generator=g)) if with_res else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    dz = _bf(torch.randn(rows, C, generator=g))
    flags = {"none": 0, "relu": H.EPI_RELU_POST, "relu_pre": H.EPI_RELU_PRE}[flags_name]
    yd = y.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    rd = res.double().requires_grad_(True) if with_res else None
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    n = (yd - mean) / torch.sqrt(var + 1e-5) * gd + bd
    if flags_name == "relu_pre":
        dz = dz * (n.detach().abs() > 1e-4).float()
        n = F.relu(n)
    if with_res:
        n = n + rd
    if flags_name == "relu":
        dz = dz * (n.detach().abs() > 1e-4).float()
        n = F.relu(n)
    (n * dz.double()).sum().backward()
    bf = torch.bfloat16
    yg, dzg = y.to(DEV, bf), dz.to(DEV, bf)
    resg = res.to(DEV, bf) if with_res else None
    mg, vg = mean.detach().float().to(DEV), var.detach().float().to(DEV)
    gg, bg = gamma.to(DEV), beta.to(DEV)
    fl = flags | H.BN_Y_BF16 | H.ACT_BF16
    # the statistics kernel over the bf16 tensor
    m2, v2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ws0 = torch.empty(max(1, lib.lt_bn_stats_workspace(rows, C)), dtype=torch.uint8, device=DEV)
    H.check(lib.lt_bn_stats_fwd(H.LT_BF16, yg.data_ptr(), rows, C, m2.data_ptr(), v2.data_ptr(), None, None, 0.1, ws0.data_ptr(), _st()), "stats")
    tag = "train/bn_act bf16 activations %dx%d %s%s" % (rows, C, flags_name, "+res" if with_res else "")
    check(tag + " mean", m2.cpu(), mean.detach(), 1e-5)
    check(tag + " var", v2.cpu(), var.detach(), 1e-5)
    z = torch.empty(rows, C, dtype=bf, device=DEV)
    H.check(lib.lt_bn_act_fwd(yg.data_ptr(), mg.data_ptr(), vg.data_ptr(), gg.data_ptr(), bg.data_ptr(), H.ptr(resg), z.data_ptr(), None, rows, C, 1e-5, fl, _st()), "fwd")
    check(tag + " fwd", z.float().cpu(), n.detach(), 8e-3)          # one rounding to bf16: <= 2^-8 of each element, ~1.7e-3 rms
    assert lib.lt_bn_act_fwd(yg.data_ptr(), mg.data_ptr(), vg.data_ptr(), gg.data_ptr(), bg.data_ptr(), H.ptr(resg), z.data_ptr(), z.data_ptr(), rows, C, 1e-5, fl, _st()) == -1
    dy, dga, dbe = torch.empty(rows, C, dtype=bf, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dres = torch.full((rows, C), 3.0, dtype=bf, device=DEV) if with_res else None
    ws = torch.empty(max(1, lib.lt_bn_act_bwd_workspace(rows, C)), dtype=torch.uint8, device=DEV)
    H.check(lib.lt_bn_act_bwd(dzg.data_ptr(), yg.data_ptr(), H.ptr(resg), mg.data_ptr(), vg.data_ptr(), gg.data_ptr(), bg.data_ptr(), dy.data_ptr(), None,
                              dga.data_ptr(), dbe.data_ptr(), H.ptr(dres), 1 if with_res else 0, rows, C, 1e-5, fl, ws.data_ptr(), _st()), "bwd")
    check(tag + " dy", dy.float().cpu(), yd.grad, 8e-3)
    check(tag + " dgamma", dga.cpu(), gd.grad, 2e-5)
    check(tag + " dbeta", dbe.cpu(), bd.grad, 2e-5)
    if with_res:
        check(tag + " dres (accumulated onto 3)", dres.float().cpu(), rd.grad + 3.0, 8e-3)


def test_bf16_activation_helpers():
    """lt_convert_pad (casts and channel padding between fp32 and bf16), lt_channel_sum_dt / lt_act_bwd / lt_maxpool_bwd_dt on bf16 tensors."""
    H, lib = _lib()
    g = torch.Generator().manual_seed(77)
    bf = torch.bfloat16
    rows, C = 4099, 17
    x = torch.randn(rows, C, generator=g)
    xg = x.to(DEV)
    # fp32 -> bf16 with padding 17 -> 32, bf16 -> fp32 plain cast, fp32 -> bf16 plain cast (vector path: rows * C % 4 == 0 needs an even count)
    o16 = torch.empty(rows, 32, dtype=bf, device=DEV)
    H.check(lib.lt_convert_pad(H.LT_F32, xg.data_ptr(), H.LT_BF16, o16.data_ptr(), rows, C, 32, _st()), "pad")
    assert torch.equal(o16[:, :C].cpu(), x.to(bf)) and float(o16[:, C:].float().abs().max()) == 0.0
    x4 = torch.randn(4096, 32, generator=g)
    a16 = torch.empty(4096, 32, dtype=bf, device=DEV)
    x4g = x4.to(DEV)
    H.check(lib.lt_convert_pad(H.LT_F32, x4g.data_ptr(), H.LT_BF16, a16.data_ptr(), 4096, 32, 32, _st()), "cast")
    assert torch.equal(a16.cpu(), x4.to(bf))
    b32 = torch.empty(4096, 32, device=DEV)
    H.check(lib.lt_convert_pad(H.LT_BF16, a16.data_ptr(), H.LT_F32, b32.data_ptr(), 4096, 32, 32, _st()), "cast back")
    assert torch.equal(b32.cpu(), x4.to(bf).float())
    o17 = torch.empty(rows, C, dtype=bf, device=DEV)          # 17 columns, same width: the scalar tail path
    H.check(lib.lt_convert_pad(H.LT_F32, xg.data_ptr(), H.LT_BF16, o17.data_ptr(), rows, C, C, _st()), "cast 17")
    assert torch.equal(o17.cpu(), x.to(bf))
    # channel sums of a bf16 tensor (vector and scalar column paths)
    for cc in (17, 64):
        t = torch.randn(rows, cc, generator=g).to(bf)
        tg = t.to(DEV)
        out = torch.empty(cc, device=DEV)
        ws = torch.empty(max(1, lib.lt_channel_sum_workspace(rows, cc)), dtype=torch.uint8, device=DEV)
        H.check(lib.lt_channel_sum_dt(H.LT_BF16, tg.data_ptr(), rows, cc, out.data_ptr(), 0, ws.data_ptr(), _st()), "lt_channel_sum_dt")
        check("train/channel_sum bf16 C=%d" % cc, out.cpu(), t.double().sum(0), 1e-6)
    # activation backward on bf16 tensors: z = relu(v + res)
    pre, res, dz = (torch.randn(rows, C, generator=g).to(bf) for _ in range(3))
    zz = F.relu(pre.float() + res.float()).to(bf)
    dy, dres = torch.empty(rows, C, dtype=bf, device=DEV), torch.full((rows, C), 2.0, dtype=bf, device=DEV)
    dzg, zzg, resg = dz.to(DEV), zz.to(DEV), res.to(DEV)
    H.check(lib.lt_act_bwd(dzg.data_ptr(), zzg.data_ptr(), resg.data_ptr(), dy.data_ptr(), dres.data_ptr(), 1, rows * C, H.EPI_RELU_POST | H.ACT_BF16, _st()), "lt_act_bwd")
    gref = dz.float() * (zz.float() > 0).float()
    assert torch.equal(dy.cpu(), gref.to(bf)) and torch.equal(dres.cpu(), (gref + 2.0).to(bf))
    # max pool backward on bf16 tensors (V2V's 2^3 / stride 2 pool: one contribution per input, exact)
    xs = F.relu(torch.randn(2, 32, 8, 8, 8, generator=g)).to(bf).float()
    xd = xs.double().requires_grad_(True)
    y = F.max_pool3d(xd, 2, 2, 0)
    dyy = torch.randn(y.shape, generator=g).to(bf).float()
    (y * dyy.double()).sum().backward()
    xcl, dycl = to_cl(xs, None, bf), to_cl(dyy, None, bf)
    dx = torch.zeros_like(xcl)
    N, D, Hh, W, Cc = xcl.shape
    H.check(lib.lt_maxpool_bwd_dt(H.LT_BF16, xcl.data_ptr(), dycl.data_ptr(), dx.data_ptr(), N, D, Hh, W, Cc, H.i3((2, 2, 2)), H.i3((2, 2, 2)), H.i3((0, 0, 0)), _st()), "lt_maxpool_bwd_dt")
    m = xs > 0
    check("train/maxpool_bwd bf16 (x > 0)", from_cl(dx, 3) * m, xd.grad * m, 1e-6)


CONV_CASES = [  # nd, Cin, Cout, k, stride, pad, transposed, spatial
    (2, 16, 32, 3, 1, 1, False, (10, 12)),
    (2, 64, 64, 1, 1, 0, False, (9, 7)),
    (2, 32, 64, 3, 2, 1, False, (12, 16)),
    (2, 32, 128, 1, 2, 0, False, (12, 16)),
    (2, 64, 32, 4, 2, 1, True, (6, 8)),
    (3, 32, 32, 3, 1, 1, False, (6, 6, 6)),
    (3, 16, 32, 1, 1, 0, False, (4, 6, 8)),
    (3, 64, 32, 2, 2, 0, True, (3, 4, 5)),
    (3, 32, 17, 1, 1, 0, False, (4, 4, 4)),
    (3, 32, 16, 7, 1, 3, False, (8, 8, 8)),
    (3, 32, 64, 3, 1, 1, False, (4, 8, 16)),       # whole bricks of 2 x 4 x 16 voxels: the LDS-brick weight-gradient kernel
    (3, 64, 32, 3, 1, 1, False, (2, 4, 32)),
    (3, 32, 16, 7, 1, 3, False, (4, 4, 16)),       # V2V's 7^3 front layer: one kd plane of the filter per workgroup, 16x16x4 MFMA
    (2, 32, 64, 3, 1, 1, False, (16, 8)),          # 8 x 8 pixel bricks: the 2D LDS-brick weight-gradient kernel (one partly filled ci block)
    (2, 256, 32, 3, 1, 1, False, (8, 8)),          # two ci blocks of 128
    (2, 256, 128, 1, 1, 0, False, (5, 7)),         # the LDS-tiled pointwise weight-gradient GEMM (ragged last chunk of rows)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "nd%d_%dto%d_k%ds%dp%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "_T" if c[6] else ""))
@pytest.mark.parametrize("mode", ["bn_relu_res", "bias_only"])
@pytest.mark.parametrize("mixed", [False, True, "act16"], ids=["fp32", "bf16mma", "act16"])
def test_tape_layer_gradients(case, mode, mixed):
    """One layer through TrainTape: z = act(BN_train(conv(x) + b) [+ res]); all gradients vs torch-CPU fp64 autograd.  ``mixed``: the
    convolution and its input gradient on the bf16 MFMA (bf16 copies of the operands, fp32 accumulation and storage), everything else
    fp32 -- compared with the same fp64 reference at bf16-operand tolerances."""
    import lt_engine as E
    import lt_train
    nd, Cin, Cout, k, s, p, tr, sp = case
    if mode == "bn_relu_res" and Cout % 4:
        pytest.skip("BatchNorm layers have Cout % 4 == 0 in these networks (the 17-joint output layer has none)")
    g = torch.Generator().manual_seed(Cin * 7 + Cout + k)
    N = 3
    x = torch.randn(N, Cin, *sp, generator=g)
    wshape = (Cin, Cout) if tr else (Cout, Cin)
    w = torch.randn(*wshape, *([k] * nd), generator=g) * (1.0 / np.sqrt(Cin * k ** nd))
    b = torch.randn(Cout, generator=g) * 0.1
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    conv = {(2, False): F.conv2d, (3, False): F.conv3d, (2, True): F.conv_transpose2d, (3, True): F.conv_transpose3d}[(nd, tr)]
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    gd, btd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = conv(xd, wd, bd, stride=s, padding=p)
    use_bn = mode == "bn_relu_res"
    res = torch.randn(y.shape, generator=g) if use_bn else None
    rd = res.double().requires_grad_(True) if use_bn else None
    if use_bn:
        dims = [0] + list(range(2, 2 + nd))
        mean = y.mean(dims, keepdim=True); var = y.var(dims, unbiased=False, keepdim=True)
        shape = [1, -1] + [1] * nd
        y = (y - mean) / torch.sqrt(var + 1e-5) * gd.reshape(shape) + btd.reshape(shape)
        pre = (y + rd).detach()
        y = F.relu(y + rd)
    dz = torch.randn(y.shape, generator=g)
    if use_bn:
        dz = dz * (pre.abs() > (1e-4 if not mixed else 0.6 if mixed == "fp8v2v" else 5e-2)).float()      # no upstream gradient where rounding (fp32 / bf16 / e4m3 operands) could flip the ReLU mask
    (y * dz.double()).sum().backward()

    wp, bp = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
    gp, btp = torch.nn.Parameter(gamma.to(DEV)), torch.nn.Parameter(beta.to(DEV))
    act16 = mixed in ("act16", "fp8v2v")          # bf16 activations and activation gradients on top of the bf16 MFMA (train_precision "act16")
    fp8 = mixed == "fp8v2v"                       # ... and the 3x3x3 convolutions + their input gradients on the fp8 MFMA (e4m3: 3 mantissa bits)
    adt = torch.bfloat16 if act16 else torch.float32
    tape = lt_train.TrainTape(DEV, params=[wp, bp, gp, btp], mixed=bool(mixed), act16=act16, fp8_3d=fp8)
    T1, T2, T3 = (2e-5, 5e-5, 1e-6) if not mixed else (2e-2, 2e-2, 2e-2)
    if fp8:
        T1, T2 = 0.25, 0.25
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    xa = E.Act(to_cl(x, None, adt))
    ra = E.Act(to_cl(res, None, adt)) if use_bn else None
    z = tape.conv(xa, wp, bp, (gp, btp, rm, rv) if use_bn else None, stride=s, pad=p, transposed=tr, relu=use_bn, residual=ra)
    tag = "train/layer%s nd%d %d->%d k%d s%d p%d%s %s" % (" fp8v2v" if fp8 else " act16" if act16 else " bf16mma" if mixed else "", nd, Cin, Cout, k, s, p, " T" if tr else "", mode)
    if fp8:
        assert any("fp8" in (tape.labels.get(id(f)) or "") for f in tape.fwd_ops), "the fp8 path was not taken"
    assert z.t.dtype == adt
    check(tag + " z", from_cl(z.t, nd), y.detach(), T1)
    dzb = to_cl(dz, None, adt)
    tape.seed(z, dzb)
    pg = tape.run_backward()          # records the backward while running it
    first = {k: v.clone() for k, v in pg.items()}
    # REPLAY with other weights and another upstream gradient, then with the original ones again: the recorded closures read the live values
    with torch.no_grad():
        saved = [t.clone() for t in (wp, bp, gp, btp)]
        for t in (wp, bp, gp, btp):
            t.mul_(1.5).add_(0.01)
        dzb.mul_(-2.0)
    tape.run_forward(); tape.run_backward()
    assert not torch.allclose(pg[wp], first[wp])
    with torch.no_grad():
        for t, s0 in zip((wp, bp, gp, btp), saved):
            t.copy_(s0)
        dzb.copy_(to_cl(dz))
        rm.zero_(); rv.fill_(1.0)
    tape.run_forward(); pg = tape.run_backward()
    for k in first:
        assert torch.allclose(pg[k], first[k], rtol=1e-5, atol=1e-6 * float(first[k].abs().max())), "replay differs from the recording"
    check(tag + " dx", from_cl(tape.grad_of(xa), nd), xd.grad, T2)
    check(tag + " dw", pg[wp].cpu(), wd.grad, T2)
    if use_bn:
        check(tag + " dgamma", pg[gp].cpu(), gd.grad, T2)
        check(tag + " dbeta", pg[btp].cpu(), btd.grad, T2)
        check(tag + " dres", from_cl(tape.grad_of(ra), nd), rd.grad, T3)
        # the bias in front of a training-mode BatchNorm has an exactly zero gradient; ours is rounding noise of the channel sums
        # (bf16 activation gradients: the sum of dy's rounding errors, ~2^-9 sqrt(n) of its magnitude)
        assert float(pg[bp].abs().max()) <= (3e-3 if act16 else 1e-4) * float(dz.abs().sum() / Cout)
        n_el = y.numel() // Cout
        yb = conv(x.double(), w.double(), b.double(), stride=s, padding=p)
        dims = [0] + list(range(2, 2 + nd))
        ts = 1e-5 if not mixed else 0.1 if fp8 else 1e-2          # (the statistics of an e4m3 convolution's output carry its ~4 % noise)
        check(tag + " running_mean", rm.cpu(), 0.1 * yb.mean(dims), ts)
        check(tag + " running_var", rv.cpu(), 0.9 + 0.1 * yb.var(dims, unbiased=True), ts)
    else:
        check(tag + " db", pg[bp].cpu(), bd.grad, T2)


FRAG_CASES = [  # nd, Cin, Cout, k, pad, transposed, N, spatial -- shapes whose bf16 inference kernels read fragment-order weights
    (2, 256, 256, 3, 1, False, 20, (24, 24)),        # conv2d_halo_kernel (60 tiles of 8 x 24 pixels), forward AND input gradient (the flipped filter has the same shape)
    (2, 256, 1024, 1, 0, False, 32, (24, 24)),       # 288 / 144-row fragment-order kernels (conv_igemm6: short K), input gradient 1024 -> 256 (conv_igemm7)
    (2, 256, 256, 4, 1, True, 20, (24, 24)),         # the head's 4x4 / stride-2 transposed convolution: four parities over one halo
    (3, 64, 64, 3, 1, False, 2, (32, 32, 32)),       # conv3d_halo_wreg_kernel
    (3, 16, 32, 3, 1, False, 1, (64, 64, 64)),       # ... its 16 -> 32 instantiation (forward; the 32 -> 16 input gradient has no fragment kernel)
]


@pytest.mark.parametrize("case", FRAG_CASES, ids=lambda c: "nd%d_%dto%d_k%d%s_N%d" % (c[0], c[1], c[2], c[3], "_T" if c[5] else "", c[6]))
def test_tape_live_fragment_weights(case, monkeypatch):
    """Round 6 (VERDICT r5 "next" 2 iv): the 16-bit training tape gathers the LIVE Parameters straight into the fragment-order copies the fast inference
    kernels read (PlanBuilder.live_frag + TrainTape._frag_index_map), so its forward / input gradients leave the generic tiles.  Checked here: (a) the
    gathered fragment copy equals the device packer applied to the gathered plain copy BIT FOR BIT, at record time and after the weights have changed;
    (b) forward and input gradient against torch on the bf16-rounded operands, and against the same tape with the fragments off (LT_TRAIN_NO_FRAG=1)."""
    import lt_engine as E
    import lt_train
    nd, Cin, Cout, k, p, tr, N, sp = case
    g = torch.Generator().manual_seed(Cin + 3 * Cout + k)
    bf = torch.bfloat16
    x = torch.randn(N, Cin, *sp, generator=g).to(bf).float()
    wshape = (Cin, Cout) if tr else (Cout, Cin)
    w = (torch.randn(*wshape, *([k] * nd), generator=g) * (1.0 / np.sqrt(Cin * k ** nd))).to(bf).float()
    conv = {(2, False): F.conv2d, (3, False): F.conv3d, (2, True): F.conv_transpose2d}[(nd, tr)]
    s = 2 if tr else 1
    xr = x.clone().requires_grad_(True)
    y = conv(xr, w, None, stride=s, padding=p)
    dz = torch.randn(y.shape, generator=g).to(bf).float()
    (y * dz).sum().backward()
    outs = {}
    for mode in ("frag", "plain"):
        if mode == "plain":
            monkeypatch.setenv("LT_TRAIN_NO_FRAG", "1")
        wp = torch.nn.Parameter(w.to(DEV))
        tape = lt_train.TrainTape(DEV, params=[wp], mixed=True, act16=True)
        xa = E.Act(to_cl(x, None, bf))
        z = tape.conv(xa, wp, None, None, stride=s, pad=p, transposed=tr)
        if mode == "frag":
            assert getattr(tape, "n_frag_layers", 0) >= 1, "no fragment-order copy was recorded for this shape"
            info = tape.pbh.last_info
            for pi, wfr, pack in info["wfrag"]:
                ref = torch.zeros_like(wfr); pack(info["wdev"][pi].data_ptr(), ref.data_ptr())
                assert torch.equal(ref.view(torch.int16), wfr.view(torch.int16)), "gathered fragments != pack(gathered plain layout)"
        else:
            assert getattr(tape, "n_frag_layers", 0) == 0
        tape.seed(z, to_cl(dz, None, bf))
        tape.run_backward()
        outs[mode] = (from_cl(z.t, nd).clone(), from_cl(tape.grad_of(xa), nd).clone())
        if mode == "frag":          # the weights change in place (an optimiser step): the replay must see them in BOTH layouts
            n0 = tape.n_frag_layers
            with torch.no_grad():
                wp.mul_(-0.5)
            tape.run_forward()
            torch.cuda.synchronize()
            for pi, wfr, pack in info["wfrag"]:
                ref = torch.zeros_like(wfr); pack(info["wdev"][pi].data_ptr(), ref.data_ptr())
                assert torch.equal(ref.view(torch.int16), wfr.view(torch.int16))
            z2 = from_cl(z.t, nd)
            check("train/live fragments nd%d %d->%d k%d%s: forward after an in-place weight update" % (nd, Cin, Cout, k, " T" if tr else ""), z2, -0.5 * y.detach(), 1.5e-2)
            record("train/live fragments nd%d %d->%d k%d%s: fragment-order copies (forward + input gradient)" % (nd, Cin, Cout, k, " T" if tr else ""), n0)
    tag = "train/live fragments nd%d %d->%d k%d%s N%d" % (nd, Cin, Cout, k, " T" if tr else "", N)
    check(tag + " z vs torch", outs["frag"][0], y.detach(), 1.5e-2)
    check(tag + " dx vs torch", outs["frag"][1], xr.grad, 1.5e-2)
    check(tag + " z vs the generic tiles", outs["frag"][0], outs["plain"][0], 1.5e-2)
    check(tag + " dx vs the generic tiles", outs["frag"][1], outs["plain"][1], 1.5e-2)


W16_CASES = [  # N, (D, H, W), Cin, Cout, k (taps per dim), stride, pad
    (11, (1, 9, 7), 64, 96, (1, 1, 1), 1, 0),        # pointwise, two octet groups with a ragged second one, Cout not a multiple of 32
    (8, (1, 10, 12), 32, 64, (1, 3, 3), 1, 1),       # 2D 3x3
    (3, (1, 12, 16), 32, 64, (1, 3, 3), 2, 1),       # strided
    (16, (1, 6, 8), 8, 64, (1, 7, 7), 2, 3),         # the stem's shape family: 49 taps x 8 channels
    (5, (4, 6, 8), 16, 32, (1, 1, 1), 1, 0),         # K <= 64: the 128 x 64 wave tile
    (2, (4, 4, 8), 32, 17, (1, 1, 1), 1, 0),         # 17 joints: scalar pack path, 32 x 256 wave tile
    (8, (4, 4, 8), 32, 32, (3, 3, 3), 1, 1),         # LDS bricks of octets
    (9, (2, 4, 16), 64, 32, (3, 3, 3), 1, 1),        # ... two ci blocks, two octet groups
    (4, (4, 2, 8), 16, 32, (3, 3, 3), 1, 1),         # ... 16 input channels: two taps per column block
    (8, (6, 6, 6), 32, 32, (3, 3, 3), 1, 1),         # 3^3 whose volume is no whole number of bricks: generic kernel
    (8, (3, 4, 8), 32, 16, (7, 7, 7), 1, 3),         # the 7^3 front layer: one kd plane per workgroup on the 16x16x32 MFMA
    (13, (1, 5, 3), 128, 256, (1, 1, 1), 1, 0),      # two 128-channel tiles of the unpacked kernel, rows of 3 pixels (a step of 4 wraps), ragged octet group
    (8, (1, 2, 1), 64, 128, (1, 3, 3), 1, 1),        # ... a 2 x 1 image: every step wraps in all dimensions, most taps in the padding
    (24, (1, 12, 12), 256, 64, (1, 1, 1), 1, 0),     # ... the 64 x 128 wave tile (layer1's reduce convolutions), three octet groups
]


@pytest.mark.parametrize("case", W16_CASES, ids=lambda c: "N%d_%s_%dto%d_k%s_s%d" % (c[0], "x".join(map(str, c[1])), c[2], c[3], "".join(map(str, c[4])), c[5]))
def test_conv_wgrad_bf16_vs_fp32_kernel_on_rounded_operands(case):
    """lt_pack_n8_bf16 + lt_conv_wgrad_bf16 (image-octet operands on the bf16 MFMA) against lt_conv_wgrad (exact-fp32 MFMA) fed with the SAME
    bf16-rounded operands: products are exact in both, only the fp32 summation order differs (gate 2e-5 of max); against the fp32 kernel on
    the unrounded operands the difference is bf16 rounding of the operands (recorded, gate 2e-2).  Twice: bitwise repeatable."""
    H, lib = _lib()
    N, (D, Hh, W), Cin, Cout, ks, s, p = case
    g = torch.Generator().manual_seed(N * 131 + Cin + Cout)
    pd = tuple(p if k > 1 else 0 for k in ks)
    st3 = tuple(s if k > 1 else 1 for k in ks) if any(k > 1 for k in ks) else (1, 1, 1)
    Do, Ho, Wo = [(n + 2 * q - k) // t + 1 for n, q, k, t in zip((D, Hh, W), pd, ks, st3)]
    x = torch.randn(N, D, Hh, W, Cin, generator=g).to(DEV)
    dy = torch.randn(N, Do, Ho, Wo, Cout, generator=g).to(DEV)
    taps = torch.tensor([(a, b, c, 0) for a in range(ks[0]) for b in range(ks[1]) for c in range(ks[2])], dtype=torch.int32, device=DEV)
    ntaps = taps.shape[0]
    import lt_engine as E
    cop, kp = E.cout_pad_of(Cout), ntaps * Cin
    rows = N * Do * Ho * Wo

    def fp32(dyt, xt):
        dw = torch.zeros(cop, kp, device=DEV)
        ws = torch.empty(max(int(lib.lt_conv_wgrad_workspace(rows, cop, kp)), 16), dtype=torch.uint8, device=DEV)
        H.check(lib.lt_conv_wgrad(dyt.data_ptr(), xt.data_ptr(), taps.data_ptr(), dw.data_ptr(), N, D, Hh, W, Cin, Do, Ho, Wo, H.i3(st3), H.i3(pd), Cout, Cout, cop, kp,
                                  ntaps, 0, ws.data_ptr(), _st()), "lt_conv_wgrad")
        return dw

    def bf16():
        pa = torch.empty(int(lib.lt_pack_n8_bf16_bytes(N, Do * Ho * Wo, Cout)), dtype=torch.uint8, device=DEV)
        pb = torch.empty(int(lib.lt_pack_n8_bf16_bytes(N, D * Hh * W, Cin)), dtype=torch.uint8, device=DEV)
        H.check(lib.lt_pack_n8_bf16(dy.data_ptr(), pa.data_ptr(), N, Do * Ho * Wo, Cout, Cout, _st()), "lt_pack_n8_bf16")
        H.check(lib.lt_pack_n8_bf16(x.data_ptr(), pb.data_ptr(), N, D * Hh * W, Cin, Cin, _st()), "lt_pack_n8_bf16")
        dw = torch.full((cop, kp), float("nan"), device=DEV)
        ws = torch.empty(max(int(lib.lt_conv_wgrad_bf16_workspace((N + 7) // 8 * Do * Ho * Wo, cop, kp)), 16), dtype=torch.uint8, device=DEV)
        H.check(lib.lt_conv_wgrad_bf16(pa.data_ptr(), pb.data_ptr(), taps.data_ptr(), dw.data_ptr(), N, D, Hh, W, Cin, Do, Ho, Wo, H.i3(st3), H.i3(pd), Cout, Cout,
                                       cop, kp, ntaps, 0, ws.data_ptr(), _st()), "lt_conv_wgrad_bf16")
        return dw, pa

    ours, pa = bf16()
    again, _ = bf16()
    torch.cuda.synchronize()
    # the pack itself: octet (g, p, c) = the eight images' values, bf16 round-to-nearest-even, zeros past N
    G = (N + 7) // 8
    pk = pa.view(torch.bfloat16).reshape(G, Do * Ho * Wo, Cout, 8).float().cpu()
    want = torch.zeros(G * 8, Do * Ho * Wo, Cout)
    want[:N] = dy.reshape(N, -1, Cout).bfloat16().float().cpu()
    assert torch.equal(pk, want.reshape(G, 8, -1, Cout).permute(0, 2, 3, 1))
    if Cout % 8 == 0:          # the same octets from the bf16 copy of the tensor
        pa2 = torch.empty_like(pa)
        H.check(lib.lt_pack_n8_from_bf16(dy.bfloat16().contiguous().data_ptr(), pa2.data_ptr(), N, Do * Ho * Wo, Cout, Cout, _st()), "lt_pack_n8_from_bf16")
        torch.cuda.synchronize()
        assert torch.equal(pa2, pa)
    assert torch.equal(ours[:Cout], again[:Cout]), "not bitwise repeatable"
    ref_r = fp32(dy.bfloat16().float(), x.bfloat16().float())
    ref = fp32(dy, x)
    tag = "train/wgrad_bf16 N%d %s %d->%d k%s s%d" % (N, "x".join(map(str, (D, Hh, W))), Cin, Cout, "".join(map(str, ks)), s)
    check(tag + " vs fp32 kernel on rounded operands", ours[:Cout].cpu(), ref_r[:Cout].cpu(), 2e-5)
    check(tag + " vs fp32 kernel", ours[:Cout].cpu(), ref[:Cout].cpu(), 2e-2)
    # the same GEMM straight from the channels-last bf16 tensors (lt_conv_wgrad_bf16_nhwc: the transpose in registers instead of the packs)
    covered = lib.lt_conv_wgrad_bf16_nhwc_ok(N, D, Hh, W, Cin, Cin, Do, Ho, Wo, H.i3(st3), H.i3(pd), Cout, Cout, cop, kp, ntaps)
    # (not covered: 17 joints, the 7^3 layer, and the LDS-brick shapes with 16 input channels -- those keep the packed kernels)
    assert covered == (0 if (Cout % 4 or (ks == (3, 3, 3) and Cin % 32 and (D, Hh, W) != (6, 6, 6)) or ks == (7, 7, 7)) else 1)
    if covered:
        x16, dy16 = x.bfloat16().contiguous(), dy.bfloat16().contiguous()
        ws = torch.empty(max(int(lib.lt_conv_wgrad_bf16_workspace((N + 7) // 8 * Do * Ho * Wo, cop, kp)), 16), dtype=torch.uint8, device=DEV)

        def direct():
            dw = torch.full((cop, kp), float("nan"), device=DEV)
            H.check(lib.lt_conv_wgrad_bf16_nhwc(dy16.data_ptr(), x16.data_ptr(), taps.data_ptr(), dw.data_ptr(), N, D, Hh, W, Cin, Cin, Do, Ho, Wo, H.i3(st3), H.i3(pd),
                                                Cout, Cout, cop, kp, ntaps, 0, ws.data_ptr(), _st()), "lt_conv_wgrad_bf16_nhwc")
            return dw

        d1 = direct()
        d2 = direct()
        torch.cuda.synchronize()
        assert torch.equal(d1[:Cout], d2[:Cout]), "not bitwise repeatable"
        assert not torch.isnan(d1).any()
        check(tag + " unpacked vs fp32 kernel on rounded operands", d1[:Cout].cpu(), ref_r[:Cout].cpu(), 2e-5)


@pytest.mark.parametrize("case", [CONV_CASES[0], CONV_CASES[2], CONV_CASES[4], CONV_CASES[5], CONV_CASES[14]],
                         ids=lambda c: "nd%d_%dto%d_k%ds%dp%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "_T" if c[6] else ""))
def test_tape_layer_gradients_with_bf16_conv_outputs(case, monkeypatch):
    """LT_TRAIN_Y16=1 (opt-in): the convolution output in front of a BatchNorm is stored in bf16 (LT_BN_Y_BF16 in lt_bn_act_fwd / _bwd,
    dtype = LT_BF16 in lt_bn_stats_fwd, live weights without fragment-order copies) -- the same gates as the mixed mode."""
    monkeypatch.setenv("LT_TRAIN_Y16", "1")
    test_tape_layer_gradients(case, "bn_relu_res", True)


def test_act16_step_with_a_frozen_backbone_batchnorm(golden_dir):
    """train_precision "act16" with the backbone's BatchNorm modules left in eval() (LT_BN_FROZEN | LT_ACT_BF16: frozen running statistics in the
    normalisation and its backward, bf16 tensors): against the reference's own fp32 step of that setting (tests/golden/train_step_frozen_bn.npz)."""
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    G = np.load(os.path.join(golden_dir, "train_step_frozen_bn.npz"))
    c, cfg, sd, inp = _train_case()
    m = VolumetricTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV).train()
    for mod in m.backbone.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.eval()
    m.train_precision = "act16"
    batch = {"cameras": _cameras(inp, c["B"]), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    np.random.seed(c["seed"] + 100)
    kp, feats, vols, conf, cuboids, cvs, bps = m(inp["images"].to(DEV), None, batch)
    gt, val = torch.from_numpy(G["gt"]).to(DEV), torch.from_numpy(G["val"]).to(DEV)
    mae = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val)
    ce = L.VolumetricCELoss()(cvs, vols, gt, val)
    (mae + 0.01 * ce).backward()
    named = dict(m.named_parameters())
    errs = []
    for n in G["names"]:
        n = str(n)
        if ZERO_GRAD.search(n):
            continue
        f = named[n].grad.detach().double().cpu().reshape(-1)
        sub = f[::max(1, f.numel() // 129)][:129]
        errs.append(float((sub - torch.from_numpy(G["g/" + n]).double()).abs().max()) / float(G["gn/" + n][1]))
    errs.sort()
    d = float(((kp.detach().cpu().double() - torch.from_numpy(G["kp"]).double()).abs() / torch.from_numpy(G["kp"]).double().abs().clamp(min=1.0)).max())
    st = {"joints_max_rel": d, "mae": float(mae.detach()), "mae_reference": float(G["mae"]), "parameter_gradient_err_median": errs[len(errs) // 2],
          "parameter_gradient_err_p90": errs[int(len(errs) * 0.9)]}
    record("train/act16 with frozen backbone BatchNorm, one step vs the reference's fp32 step", st)
    print(st)
    # a running-statistics BatchNorm has none of the batch-statistics sensitivity: tighter than the plain act16 gate
    assert d < 6e-2 and errs[len(errs) // 2] <= 0.07 and errs[int(len(errs) * 0.9)] <= 0.22, st          # achieved: 3.2e-2, 4.3 %, 15.5 %
    # the frozen modules' running statistics are untouched
    bn1 = m.backbone.bn1
    assert torch.equal(bn1.running_mean.cpu(), sd["backbone.bn1.running_mean"]) and int(bn1.num_batches_tracked) == int(sd["backbone.bn1.num_batches_tracked"])


def test_act16_step_with_the_confidence_heads_tracks_the_fp32_storage_mode(golden_dir):
    """train_precision "act16" with ``conf_norm`` aggregation: the vol_confidences head (conv + BN + max pool twice, global average pool, three linears,
    sigmoid -- the last layer stores fp32, the unprojection reads and differentiates fp32 confidences) through the 16-bit-activation tape, against
    round 3's "bf16" mode (bf16 MFMA over fp32 storage, itself gated against the reference's step): same weights, inputs, rotation."""
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    c, cfg, sd, inp = _train_case("conf_norm")
    batch = {"cameras": _cameras(inp, c["B"]), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    gt = (torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float() + 20.0).to(DEV)
    val = torch.ones(c["B"], 17, 1, device=DEV)

    def run(prec):
        m = VolumetricTriangulationNet(cfg, device=DEV)
        m.load_state_dict(sd, strict=True)
        m.to(DEV).train()
        m.train_precision = prec
        np.random.seed(c["seed"] + 100)
        kp, feats, vols, conf, cuboids, cvs, bps = m(inp["images"].to(DEV), None, batch)
        loss = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)
        loss.backward()
        torch.cuda.synchronize()
        g = {n: p.grad.detach().double().cpu().reshape(-1) for n, p in m.named_parameters() if p.grad is not None}
        return kp.detach().cpu().double(), conf.detach().cpu().double(), float(loss.detach()), g

    k0, c0, l0, g0 = run("bf16")
    k1, c1, l1, g1 = run("act16")
    assert set(g0) == set(g1) and all(bool(torch.isfinite(v).all()) for v in g1.values())
    head = [n for n in g0 if "vol_confidences" in n]
    assert head, "the confidence head received no gradient"
    cos = {grp: float((torch.cat([g0[n] for n in names]) @ torch.cat([g1[n] for n in names])) /
                      (torch.cat([g0[n] for n in names]).norm() * torch.cat([g1[n] for n in names]).norm() + 1e-300))
           for grp, names in (("confidence head", head), ("backbone", [n for n in g0 if n.startswith("backbone.") and "vol_confidences" not in n]))}
    d_conf = float((c0 - c1).abs().max())
    d_kp = float(((k0 - k1).abs() / k0.abs().clamp(min=1.0)).max())
    record("train/act16 with conf_norm vs the bf16 (fp32-storage) mode, first step", {"confidences_max_abs": d_conf, "joints_max_rel_1mm_floor": d_kp, "loss_bf16": l0,
                                                                                      "loss_act16": l1, "gradient_cosine": cos})
    print(d_conf, d_kp, l0, l1, cos)
    assert d_conf < 2e-2 and d_kp < 0.12 and abs(l1 - l0) <= 0.02 * abs(l0) and cos["backbone"] > 0.9 and cos["confidence head"] > 0.8, (d_conf, d_kp, l0, l1, cos)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0] == 3 and c[3] == 3 and c[4] == 1 and not c[6]],
                         ids=lambda c: "nd%d_%dto%d_k%ds%dp%d" % (c[0], c[1], c[2], c[3], c[4], c[5]))
def test_tape_layer_gradients_fp8_3d(case):
    """train_precision "fp8v2v" (BASELINE config 5: "fp8 MFMA for V2V 3D convs"): the 3x3x3 layer and its input gradient on the fp8 MFMA -- e4m3
    operands with per-tensor amax scales computed on the device every step (activations, gradients and the live weights) -- inside the
    16-bit-activation tape; the weight gradient stays on the bf16 MFMA.  Gated at e4m3's resolution (3 mantissa bits: ~4 % rms per sum)."""
    test_tape_layer_gradients(case, "bn_relu_res", "fp8v2v")


def test_adam_step_vs_torch():
    import lt_train
    g = torch.Generator().manual_seed(3)
    p0 = [torch.randn(1000, generator=g), torch.randn(33, 7, generator=g)]
    ref = [torch.nn.Parameter(t.clone()) for t in p0]
    ours = [torch.nn.Parameter(t.clone().to(DEV)) for t in p0]
    o_ref = torch.optim.Adam([{"params": ref[:1]}, {"params": ref[1:], "lr": 1e-2}], lr=1e-3)
    o_our = lt_train.Adam([{"params": ours[:1]}, {"params": ours[1:], "lr": 1e-2}], lr=1e-3)
    for step in range(5):
        for r, o in zip(ref, ours):
            gr = torch.randn(r.shape, generator=g) * (10.0 ** (step - 2))
            r.grad = gr.clone(); o.grad = gr.clone().to(DEV)
        o_ref.step(); o_our.step()
    for i, (r, o) in enumerate(zip(ref, ours)):
        check("train/adam 5 steps tensor %d (parameter delta)" % i, o.detach().cpu() - p0[i], r.detach() - p0[i], 1e-4)      # 1e-4 of the largest delta = 1-2 ulp of the parameters themselves


ZERO_GRAD = re.compile(r"^volume_net\.(.*\.(block\.0|res_branch\.0|res_branch\.3|skip_con\.0)|output_layer)\.bias$")


def _train_case(method="softmax", nl=18, cmu=False):
    c = dict(nl=nl, B=2, NV=3, H=128, V=64, seed=12)
    cfg = synth.vol_config(c["nl"], c["V"], method, 1.0, "coco" if cmu else "mpii")
    if cmu:
        cfg.model.transfer_cmu_to_human36m = True
        cfg.model["transfer_cmu_to_human36m"] = True
    sd = synth.make_state_dict(spec.vol_net_spec(c["nl"], 17, method.startswith("conf")), seed=c["seed"], sharpen=60.0, basic_block=nl < 50)
    inp = synth.make_inputs(c["B"], c["NV"], c["H"], seed=c["seed"], inside=False)
    return c, cfg, sd, inp


@pytest.mark.parametrize("method", ["softmax", "conf_norm", "frozen_bn", "sum", "max", "r50", "cmu"])
def test_whole_training_step_vs_reference(golden_dir, method):
    """model.train(); forward; MAE(kp * 0.1) + 0.01 * VolumetricCELoss; backward; Adam (train.py:148-243, :430-437) -- every parameter's
    gradient, the BatchNorm running statistics and the parameters after the step against the reference's own step on CPU.  conf_norm: the
    vol_confidences head (pose_resnet.py:140-174) is part of the graph -- its gradients (through the view aggregation, op.py:150-151, and
    the normalisation over the views, triangulation.py:268-269) and the backbone's second gradient path are gated the same way."""
    import lt_train
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    # frozen_bn: softmax aggregation with the BACKBONE's BatchNorm modules left in eval() (frozen running statistics, everything still trainable --
    # fine-tuning with a frozen backbone); the reference's step of the same setting: tests/golden/train_step_frozen_bn.npz
    # r50: softmax aggregation over a BOTTLENECK backbone (ResNet-50: 1x1 reduce / 3x3 / 1x1 expand blocks, strided downsample convolutions -- the block type
    # of ResNet-152); the reference's step of the same network: tests/golden/train_step_r50.npz
    # cmu: kind "coco" (cuboid centred between the hips) with transfer_cmu_to_human36m (the grid's axis permutation, triangulation.py:336-339) -- the
    # CMU Panoptic setting, whose unprojection backward walks the permuted grid: tests/golden/train_step_cmu.npz
    frozen, r50, cmu = method == "frozen_bn", method == "r50", method == "cmu"
    G = np.load(os.path.join(golden_dir, "train_step.npz" if method == "softmax" else "train_step_%s.npz" % method))
    c, cfg, sd, inp = _train_case("softmax" if (frozen or r50 or cmu) else method, 50 if r50 else 18, cmu)
    TAG = "" if method == "softmax" else "[%s] " % method
    m = VolumetricTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    m.train()
    if frozen:
        for mod in m.backbone.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.eval()
    lr, pf_lr, vn_lr = [float(v) for v in G["lrs"]]
    opt = lt_train.Adam([{"params": list(m.backbone.parameters())}, {"params": list(m.process_features.parameters()), "lr": pf_lr},
                         {"params": list(m.volume_net.parameters()), "lr": vn_lr}], lr=lr)
    batch = {"cameras": _cameras(inp, c["B"]), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    np.random.seed(c["seed"] + 100)
    kp, feats, vols, conf, cuboids, cvs, bps = m(inp["images"].to(DEV), torch.zeros(c["B"], c["NV"], 3, 4, device=DEV), batch)
    # the reference's own deviation between 1 and 8 threads / under a 1e-6 relative change of the images rides on every gate below
    kp_noise, loss_noise = float(G["kp_noise"]), float(G["loss_noise"])
    d = (kp.detach().cpu().double() - torch.from_numpy(G["kp"]).double()).abs() / torch.from_numpy(G["kp"]).double().abs().clamp(min=1.0)
    record(TAG + "train/step forward keypoints (train-mode BN), rel with 1 mm floor", {"err": float(d.max()), "tol": 1e-4 + 2 * kp_noise, "reference_self_noise": kp_noise})
    assert float(d.max()) <= 1e-4 + 2 * kp_noise, float(d.max())
    check(TAG + "train/step forward volumes", vols.detach().cpu()[:, :, ::4, ::4, ::4], G["vol_sub"], 1e-3 + 10 * kp_noise)
    check(TAG + "train/step forward features", feats.detach().cpu().reshape(c["B"] * c["NV"], *feats.shape[2:])[:, :, ::2, ::2], G["feat_sub"], 1e-4)
    if method.startswith("conf"):
        assert conf is not None and tuple(conf.shape) == tuple(G["vconf"].shape)
        e_conf = float((conf.detach().cpu() - torch.from_numpy(G["vconf"])).abs().max())
        record(TAG + "train/step returned vol_confidences (%s)" % method, {"err": e_conf, "tol": 1e-5 + 4 * float(G["vconf_noise"])})
        assert e_conf <= 1e-5 + 4 * float(G["vconf_noise"]), e_conf
    else:
        assert conf is None
    gt, val = torch.from_numpy(G["gt"]).to(DEV), torch.from_numpy(G["val"]).to(DEV)
    mae = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val)
    ce = L.VolumetricCELoss()(cvs, vols, gt, val)
    assert abs(float(mae.detach()) - float(G["mae"])) <= (1e-4 + 2 * loss_noise) * float(G["mae"]), (float(mae.detach()), float(G["mae"]))
    assert abs(float(ce.detach()) - float(G["ce"])) <= 1e-3 * float(G["ce"]), (float(ce.detach()), float(G["ce"]))
    opt.zero_grad()
    (mae + 0.01 * ce).backward()
    named = dict(m.named_parameters())
    gn2, table = 0.0, []
    gnorm_ref = float(G["grad_norm"])
    n_zero = 0
    for n in G["names"]:
        n = str(n)
        p = named[n]
        assert p.grad is not None, "no gradient for " + n
        gr = p.grad.detach().double().cpu()
        ref_norm, ref_max, ref_sum = [float(v) for v in G["gn/" + n]]
        noise = float(G["noise/" + n])
        gn2 += float(gr.pow(2).sum())
        if ZERO_GRAD.search(n) or noise > 0.05:
            # a convolution bias in front of a training-mode BatchNorm (V2V's Conv3d / ConvTranspose3d layers, v2v.py:10-16, :57-61) or
            # the output layer's bias under the softmax: the exact gradient is 0, both sides hold rounding noise of their channel sums
            assert float(gr.abs().max()) <= 10 * ref_max + 1e-6 * gnorm_ref, (n, float(gr.abs().max()), ref_max)
            n_zero += 1
            continue
        f = gr.reshape(-1)
        sub = f[::max(1, f.numel() // 129)][:129]
        e = float((sub - torch.from_numpy(G["g/" + n]).double()).abs().max()) / ref_max
        en = abs(float(gr.norm()) - ref_norm) / ref_norm
        table.append((max(e, en) / (1e-3 + 4 * noise), max(e, en), noise, n))
    table.sort(reverse=True)
    print("worst parameter gradients (err / gate, err, reference self-noise):", *["%.2f %.2e %.2e %s" % t for t in table[:8]], sep="\n  ")
    errs = sorted(t[1] for t in table)
    record(TAG + "train/step parameter gradients vs the reference's step (max|d|/max|ref| on samples, and norm; gate 1e-3 + 4 x reference self-noise)",
           {"worst_err_over_gate": table[0][0], "worst_err": errs[-1], "median_err": errs[len(errs) // 2], "parameters_compared": len(table),
            "zero_gradient_parameters": n_zero, "median_reference_self_noise": sorted(t[2] for t in table)[len(table) // 2]})
    assert table[0][0] <= 1.0, table[:8]
    for n in G["no_grad"]:
        assert named[str(n)].grad is None
    gn = float(np.sqrt(gn2))
    assert abs(gn - float(G["grad_norm"])) <= 2e-3 * float(G["grad_norm"]), (gn, float(G["grad_norm"]))
    record(TAG + "train/step global gradient norm", {"ours": gn, "reference": float(G["grad_norm"])})
    # running statistics (momentum 0.1, unbiased variance)
    bufs = dict(m.named_buffers())
    w_rs = 0.0
    for key in G.files:
        if key.startswith("rs/"):
            b = bufs[key[3:]].detach().double().cpu().reshape(-1)
            sub = b[::max(1, b.numel() // 129)][:129]
            ref = torch.from_numpy(G[key]).double()
            w_rs = max(w_rs, float((sub - ref).abs().max() / ref.abs().max().clamp(min=1e-30)))
    record(TAG + "train/step BatchNorm running statistics", {"err": w_rs, "tol": 1e-4})
    assert w_rs <= 1e-4, w_rs
    # the Adam step: parameter deltas (the first step moves every element by ~lr * sign(g); compare the moved parameters)
    opt.step()
    torch.cuda.synchronize()
    # Adam's first step is lr * g / (|g| + 1e-8), a sign function of the gradient: elements whose reference gradient is below the
    # reference's own noise (exact zeros of dead channels on our side, 1e-12 on the reference's) can differ by 2 lr -- only the elements
    # the reference knows the sign of are compared (lt_adam_step itself: test_adam_step_vs_torch)
    w_p, w_name, n_known = 0.0, None, 0
    for n in G["names"]:
        n = str(n)
        if ZERO_GRAD.search(n):
            continue
        f = named[n].detach().double().cpu().reshape(-1)
        sub = f[::max(1, f.numel() // 129)][:129]
        ref = torch.from_numpy(G["p1/" + n]).double()
        gs = torch.from_numpy(G["g/" + n]).double().abs()
        known = gs > 100 * (float(G["noise/" + n]) + 1e-3) * float(G["gn/" + n][1])
        n_known += int(known.sum())
        lr_n = lr if n.startswith("backbone.") else pf_lr if n.startswith("process_features.") else vn_lr
        e = float(((sub - ref).abs() * known).max()) / lr_n      # in units of one full Adam step
        if e > w_p:
            w_p, w_name = e, n
    record(TAG + "train/step parameters after Adam, worst |d| in units of lr", {"err": w_p, "tol": 2e-2, "name": w_name, "elements_compared": n_known})
    assert w_p <= 2e-2, (w_p, w_name)
    assert n_known > 1000, n_known
    # the REPLAYED training forward reads the updated parameters: same result as a fresh model (fresh recording) with the new state dict
    np.random.seed(c["seed"] + 100)
    kp_replay = m(inp["images"].to(DEV), None, batch)[0].detach().clone()
    m2 = VolumetricTriangulationNet(cfg, device=DEV)
    m2.load_state_dict(m.state_dict(), strict=True)
    m2.to(DEV)
    m2.train()
    if frozen:
        for mod in m2.backbone.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.eval()
    sd_before = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    np.random.seed(c["seed"] + 100)
    kp_fresh = m2(inp["images"].to(DEV), None, batch)[0].detach()
    check(TAG + "train/step replayed forward after Adam vs a fresh recording with the updated weights", kp_replay.cpu(), kp_fresh.cpu(), 1e-6)
    assert float((kp_replay.cpu() - torch.from_numpy(G["kp"])).abs().max()) > 1e-3      # and the step did move the prediction
    # and the next inference forward uses the UPDATED weights (plan cache fingerprint)
    m.eval()
    with torch.no_grad():
        kp2 = m(inp["images"].to(DEV), None, batch)[0]
    assert torch.isfinite(kp2).all()


def test_training_loss_decreases_over_steps():
    """Ten Adam steps on one fixed batch: the loss of the reference's objective goes down (sanity of the whole loop)."""
    import lt_train
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    c, cfg, sd, inp = _train_case()
    m = VolumetricTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    m.train()
    opt = lt_train.Adam(list(m.parameters()), lr=1e-4)
    batch = {"cameras": _cameras(inp, c["B"]), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    gt = torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float().to(DEV)
    val = torch.ones(c["B"], 17, 1, device=DEV)
    losses = []
    for it in range(10):
        np.random.seed(0)
        kp, _, vols, _, _, cvs, _ = m(inp["images"].to(DEV), None, batch)
        loss = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    record("train/loss over 10 Adam steps", losses)
    assert losses[-1] < losses[0], losses


def _dp_grads(rank, reducer, precision="fp32"):
    """One training forward / backward of the small case on shard ``rank`` (its own samples); returns {name: gradient (cpu)}."""
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    cfg = synth.vol_config(18, 32, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(18, 17, False), seed=12, sharpen=60.0, basic_block=True)
    inp = synth.make_inputs(2, 3, 128, seed=30 + rank, inside=False)
    m = VolumetricTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    m.train()
    m.train_precision = precision
    m.grad_reducer = reducer
    batch = {"cameras": _cameras(inp, 2), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    gt = torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float().to(DEV)
    val = torch.ones(2, 17, 1, device=DEV)
    out = {}
    for it in range(2):          # the second pass is the REPLAY of the recorded step (with the all-reduce ops inside the backward)
        np.random.seed(77)
        kp, _, vols, _, _, cvs, _ = m(inp["images"].to(DEV), None, batch)
        loss = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)
        for p in m.parameters():
            p.grad = None
        loss.backward()
        torch.cuda.synchronize()
        out[it] = {n: p.grad.detach().float().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}
    return out


def _dp_worker(rank, world, port, q, backend="gloo", own_device=False):
    import sys
    for p in (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "learnable-triangulation-pytorch_amd"),
              os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import lt_dist
    # one GPU on the box: both ranks share cuda:0, the exchange goes through gloo (RCCL needs a GPU per rank); with a GPU per rank
    # (own_device) the backend is "nccl" = RCCL over xGMI and the model lives on the rank's own device
    global DEV
    if own_device:
        os.environ["LOCAL_RANK"] = str(rank)
        DEV = "cuda:%d" % rank
    lt_dist.init(backend)
    torch.cuda.set_device(rank if own_device else 0)
    red = lt_dist.GradReducer(bucket_bytes=4 << 20)
    g = _dp_grads(rank, red)
    q.put((rank, {it: {n: (float(np.linalg.norm(v.astype(np.float64))), v.reshape(-1)[::max(1, v.size // 64)][:64].copy()) for n, v in gi.items()} for it, gi in g.items()},
           red.buckets_sent, lt_dist.comm_info(DEV)))
    lt_dist.barrier()
    lt_dist.shutdown()


def _two_rank_gradient_mean(backend, own_device, tag):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, backend, own_device)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = [_dp_grads(r, None)[0] for r in range(2)]
    worst = 0.0
    for rank, res, nb, info in got:
        assert info["nranks"] == 2 and info["backend"] == backend, info
        if own_device:
            assert len(set(info["devices"])) == 2, info          # two different GPUs took part in the all-reduce
    # the census the assertions above checked, printed and kept (VERDICT r4 "next" 7): ranks counted by an all-reduce, backend, RCCL version, every rank's device
    census = {"rank %d" % rank: {"communicator": info, "gradient_buckets_sent": nb} for rank, res, nb, info in got}
    print(tag, "census:", census)
    record(tag + " -- communicator census", census)
    return _check_two_rank_gradients(got, single, tag)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_data_parallel_two_ranks_rccl():
    """The N > 1 path on real hardware: two ranks, a GPU each, backend "nccl" (= RCCL over xGMI); construction-time broadcast, bucketed
    ReduceOp.AVG all-reduce from inside the recorded backward.  Auto-skips on the 1-GPU boxes."""
    _two_rank_gradient_mean("nccl", True, "train/data parallel 2 ranks over RCCL: averaged gradients vs mean of the shards' gradients")


def test_data_parallel_two_ranks_gradient_mean():
    """Data-parallel training (train.py:450-453): two ranks, each with its own samples, gradients averaged by lt_dist.GradReducer from
    inside the recorded backward (bucket by bucket) -- equal to the mean of the two shards' single-process gradients, in the recording
    step and in the replayed one."""
    _two_rank_gradient_mean("gloo", False, "train/data parallel 2 ranks: averaged gradients vs mean of the shards' gradients")


def _check_two_rank_gradients(got, single, tag):
    worst = 0.0
    for rank, res, nb, info in got:
        assert nb >= 4, nb          # several buckets per backward, two backwards
        for it in (0, 1):
            for n, (norm, sample) in res[it].items():
                want = 0.5 * (single[0][n].astype(np.float64) + single[1][n].astype(np.float64))
                ws = want.reshape(-1)[::max(1, want.size // 64)][:64]
                scale = max(float(np.abs(want).max()), 1e-6 * float(np.linalg.norm(want)) + 1e-12)
                if ZERO_GRAD.search(n):
                    continue
                worst = max(worst, float(np.abs(sample - ws).max()) / scale, abs(norm - float(np.linalg.norm(want))) / max(float(np.linalg.norm(want)), 1e-12))
    # round 3: the unprojection backward is a gather and the max-pool backward walks disjoint window classes -- no float atomics left in
    # the step, so the only difference to the mean of the shards' own gradients is the fp32 rounding of the sum (round 2: 2e-3)
    record(tag, {"err": worst, "tol": 1e-5})
    assert worst <= 1e-5, worst


@pytest.mark.parametrize("precision", ["fp32", "bf16", "act16"])
def test_training_step_is_bitwise_repeatable(precision):
    """Two independent recordings (fresh models, same weights / inputs / rotations) and their replays give BITWISE identical parameter
    gradients: every reduction of the step has a fixed order (column sums in fp64 slabs, weight gradients by slab partials -- on the fp32
    MFMA and, in the mixed step, on the bf16 MFMA over image octets -- the unprojection backward as a gather, the max-pool backward by
    disjoint window classes)."""
    a, b = _dp_grads(0, None, precision), _dp_grads(0, None, precision)
    assert set(a[0]) == set(b[0]) and len(a[0]) > 50
    for it in (0, 1):
        for n in a[it]:
            assert np.array_equal(a[it][n], b[it][n]), "gradient of %s differs between two runs (step %d)" % (n, it)
    for n in a[0]:
        assert np.array_equal(a[0][n], a[1][n]), "replayed step differs from the recorded one: " + n


def test_training_step_on_one_stream(monkeypatch):
    """LT_TRAIN_NO_OVERLAP=1 (the profiling mode: weight gradients, bias sums and the gradient scatter on the main stream, no side stream):
    the replayed step equals the recorded one bit for bit, and the gradients are the two-stream step's (the streams change where a kernel
    runs, not what it adds in which order: gated at 1e-6 of each tensor's largest entry rather than bitwise)."""
    two = _dp_grads(0, None, "act16")
    monkeypatch.setenv("LT_TRAIN_NO_OVERLAP", "1")
    one = _dp_grads(0, None, "act16")
    assert set(one[0]) == set(two[0]) and len(one[0]) > 50
    for n in one[0]:
        assert np.array_equal(one[0][n], one[1][n]), "replayed step differs from the recorded one: " + n
        scale = max(float(np.abs(two[0][n]).max()), 1e-30)
        assert float(np.abs(one[0][n] - two[0][n]).max()) <= 1e-6 * scale, n


def test_rccl_single_rank_process_group_comes_up():
    """The boxes these tests run on have ONE GPU, and RCCL refuses two ranks on one device -- but a world of one rank still loads
    librccl, builds a communicator and launches its kernels: backend "nccl" all-reduce / barrier on device tensors, and the
    GradReducer path driven through it (what bench.py --gpus N / --train use with N ranks)."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["LT_ROOT"], "learnable-triangulation-pytorch_amd"))
import lt_dist
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["LT_PORT"])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
t = torch.arange(1024, dtype=torch.float32, device="cuda:0")
dist.all_reduce(t)
dist.barrier()
red = lt_dist.GradReducer(bucket_bytes=1 << 12)
assert red.avg and red.backend == "nccl"      # RCCL averages inside the collective (ReduceOp.AVG): no division kernel per bucket
red.world = 2                      # force the collective path (the mean over the one rank that exists)
flat = torch.ones(4096, device="cuda:0")
red.reduce_inplace(flat[:2048]); red.reduce_inplace(flat[2048:]); red.wait_all()
torch.cuda.synchronize()
assert float(t[5]) == 5.0 and float(flat.sum()) == 4096.0, (float(t[5]), float(flat.sum()))
info = lt_dist.comm_info("cuda:0")
assert info["nranks"] == 1 and info["backend"] == "nccl" and len(info["devices"]) == 1, info
print("RCCL_OK", info)
dist.destroy_process_group()
'''
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LT_ROOT=root, LT_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    record("dist/RCCL single-rank process group (backend nccl) on the GPU box", r.stdout.strip().splitlines()[-1])


def test_real_ddp_wrapper_world1_nccl():
    """torch.nn.parallel.DistributedDataParallel (the reference's wrapper, train.py:453) around VolumetricTriangulationNet WORKS: the
    training forward is one autograd node over the parameters, so DDP's gradient hooks see ordinary .grad accumulation.  World of one rank
    on backend "nccl" (the box has one GPU); gradients equal the unwrapped model's."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
root = os.environ["LT_ROOT"]
for p in (os.path.join(root, "learnable-triangulation-pytorch_amd"), root, os.path.join(root, "tests")):
    sys.path.insert(0, p)
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["LT_PORT"])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
from oracle import spec, synth
from mvn.models import loss as L
from mvn.models.triangulation import VolumetricTriangulationNet
from test_gpu_models import _cameras
cfg = synth.vol_config(18, 32, "softmax", 1.0, "mpii")
sd = synth.make_state_dict(spec.vol_net_spec(18, 17, False), seed=12, sharpen=60.0, basic_block=True)
inp = synth.make_inputs(2, 3, 128, seed=30, inside=False)
batch = {"cameras": _cameras(inp, 2), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
gt = torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float().cuda()
val = torch.ones(2, 17, 1, device="cuda:0")
def grads(wrap):
    m = VolumetricTriangulationNet(cfg, device="cuda:0")
    m.load_state_dict(sd, strict=True)
    m.to("cuda:0").train()
    net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0]) if wrap else m
    out = None
    for it in range(2):
        np.random.seed(77)
        kp, _, vols, _, _, cvs, _ = net(inp["images"].cuda(), None, batch)
        loss = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)
        for p in m.parameters():
            p.grad = None
        loss.backward()
        torch.cuda.synchronize()
        out = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    return out, float(loss)
a, la = grads(False)
b, lb = grads(True)
assert set(a) == set(b) and len(a) > 50
worst = max(float((a[n] - b[n]).abs().max() / max(float(a[n].abs().max()), 1e-12)) for n in a if float(a[n].abs().max()) > 0)
print("DDP_OK", worst, la, lb)
assert worst <= 2e-3 and abs(la - lb) <= 1e-4 * abs(la)
dist.destroy_process_group()
'''
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LT_ROOT=root, LT_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DDP_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])
    record("dist/real DistributedDataParallel wrapper (world 1, backend nccl): gradients vs the unwrapped model", r.stdout.strip().splitlines()[-1])


def test_training_api_semantics_accumulation_stale_backward_torch_optimizer():
    """What a training loop may do around the step: gradient accumulation over two forward/backward cycles (autograd owns the returned
    gradients, the plan's arena is overwritten), backpropagating a STALE forward (refused loudly), torch.optim.Adam on the same .grad
    tensors (same update as lt_train.Adam), a no_grad forward in training mode, a second batch size (second plan)."""
    import lt_train
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    cfg = synth.vol_config(18, 32, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(18, 17, False), seed=12, sharpen=60.0, basic_block=True)
    inp = synth.make_inputs(2, 3, 128, seed=31, inside=False)
    batch = {"cameras": _cameras(inp, 2), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    gt = torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float().to(DEV)
    val = torch.ones(2, 17, 1, device=DEV)
    images = inp["images"].to(DEV)

    def make():
        m = VolumetricTriangulationNet(cfg, device=DEV)
        m.load_state_dict(sd, strict=True)
        m.to(DEV)
        m.train()
        return m

    def loss_of(m):
        np.random.seed(3)
        kp, _, vols, _, _, cvs, _ = m(images, None, batch)
        return L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)

    m = make()
    loss_of(m).backward()
    g1 = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    # running statistics moved with the first forward: the second cycle sees the same weights, so the same gradients (up to atomics)
    loss_of(m).backward()                              # no zero_grad: accumulation
    worst = 0.0
    for n, p in m.named_parameters():
        if p.grad is None or ZERO_GRAD.search(n):
            continue
        worst = max(worst, float((p.grad - 2 * g1[n]).abs().max() / g1[n].abs().max().clamp(min=1e-20)))
    record("train/api gradient accumulation (2 cycles vs 2 x 1 cycle)", {"err": worst, "tol": 2e-3})
    assert worst <= 2e-3, worst
    # a stale forward cannot be backpropagated
    l_old = loss_of(m)
    l_new = loss_of(m)
    with pytest.raises(RuntimeError, match="LATEST"):
        l_old.backward()
    l_new.backward()
    # torch.optim.Adam vs lt_train.Adam from the same gradients
    ma, mb = make(), make()
    oa = torch.optim.Adam([{"params": ma.backbone.parameters()}, {"params": ma.process_features.parameters(), "lr": 1e-3},
                           {"params": ma.volume_net.parameters(), "lr": 1e-3}], lr=1e-4)
    ob = lt_train.Adam([{"params": list(mb.backbone.parameters())}, {"params": list(mb.process_features.parameters()), "lr": 1e-3},
                        {"params": list(mb.volume_net.parameters()), "lr": 1e-3}], lr=1e-4)
    for mm, oo in ((ma, oa), (mb, ob)):
        oo.zero_grad()
        loss_of(mm).backward()
    with torch.no_grad():          # identical gradients for both optimisers (the two backwards differ by atomics noise, and Adam's first step is a sign)
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            if pa.grad is not None:
                pb.grad.copy_(pa.grad)
    before = [p.detach().clone() for p in ma.parameters()]
    oa.step(); ob.step()
    torch.cuda.synchronize()
    wd = 0.0
    for p0, pa, pb in zip(before, ma.parameters(), mb.parameters()):
        if pa.grad is not None:
            wd = max(wd, float(((pa - pb).abs().max() / (pa - p0).abs().max().clamp(min=1e-12)).detach()))
    record("train/api torch.optim.Adam vs lt_train.Adam, one step from equal gradients (difference / step size)", {"err": wd, "tol": 1e-3})
    assert wd <= 1e-3, wd
    # no_grad forward in training mode (validation without eval()): runs, returns no graph
    with torch.no_grad():
        kp = m(images, None, batch)[0]
    assert not kp.requires_grad and torch.isfinite(kp).all()
    # another batch size: a second recorded plan beside the first
    inp1 = synth.make_inputs(1, 3, 128, seed=32, inside=False)
    b1 = {"cameras": _cameras(inp1, 1), "pred_keypoints_3d": inp1["pred_keypoints_3d"]}
    kp1 = m(inp1["images"].to(DEV), None, b1)[0]
    kp1.sum().backward()
    assert len(m._train_plans) == 2 and torch.isfinite(kp1).all()


def test_dense_gradient_on_the_returned_volumes_in_training():
    """Any loss on the returned volumes trains (the reference's autograd accepts one, train.py:222-230; round 3 refused everything but
    VolumetricCELoss's sparse gradient): the SAME gradient handed over once in CE's sparse form and once as a dense (B, J, V, V, V) tensor --
    (volumes * D).sum() with D = d CE / d volumes -- must give the same parameter gradients (lt_softargmax3d_bwd_dense)."""
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    cfg = synth.vol_config(18, 32, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(18, 17, False), seed=12, sharpen=60.0, basic_block=True)
    inp = synth.make_inputs(2, 3, 128, seed=30, inside=False)
    batch = {"cameras": _cameras(inp, 2), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    gt = (torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float() + 15.0).to(DEV)
    val = torch.ones(2, 17, 1, device=DEV)
    m = VolumetricTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV).train()
    grads = []
    for dense in (False, True):
        np.random.seed(3)
        kp, _, vols, _, _, cvs, _ = m(inp["images"].to(DEV), None, batch)
        mae = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val)
        if dense:
            v2 = vols.detach().clone().requires_grad_(True)
            L.VolumetricCELoss()(cvs, v2, gt, val).backward()          # volumes that are no output of our node: the dense scatter of CE's gradient
            loss = mae + 0.01 * (vols * v2.grad).sum()
        else:
            loss = mae + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)
        for p in m.parameters():
            p.grad = None
        loss.backward()
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None})
        for c in m.modules():          # the same running statistics in front of both steps
            if isinstance(c, torch.nn.modules.batchnorm._BatchNorm):
                c.reset_running_stats()
    assert set(grads[0]) == set(grads[1]) and len(grads[0]) > 50
    worst = max(float((grads[0][n] - grads[1][n]).abs().max() / max(float(grads[0][n].abs().max()), 1e-30)) for n in grads[0] if not ZERO_GRAD.search(n))
    record("train/dense vs sparse gradient on the returned volumes: worst parameter-gradient difference (of each tensor's max)", worst)
    assert worst < 1e-4, worst


# gates of the whole reduced-precision step against the REFERENCE's fp32 step (tests/golden/train_step.npz): joints (1 mm floor), median / p90 of the
# parameter-gradient error (of each tensor's largest reference gradient) -- set at what each mode measures on this fixture plus ~25 % headroom
MIXED_STEP_GATES = {"bf16": (5e-2, 0.08, 0.20), "act16": (8e-2, 0.10, 0.28), "fp8v2v": (8e-2, 0.12, 0.28)}          # measured (round 5): bf16 3.8e-2 / 6.3 % / 15.8 %, act16 5.5e-2 / 7.7 % / 21.4 %, fp8v2v 4.4e-2 / 9.2 % / 22.1 %


@pytest.mark.parametrize("precision", ["bf16", "act16", "fp8v2v"])
def test_mixed_precision_training_step_deviation_and_descent(golden_dir, precision):
    """train_precision = "bf16": the convolutions and their input gradients on the bf16 MFMA, everything else fp32; "act16": bf16 storage of every
    activation / activation gradient too; "fp8v2v" (BASELINE config 5 as named): act16 with V2V's 3x3x3 convolutions and their input gradients on
    the fp8 MFMA.  Outside the fp32 tolerance by construction (like the bf16 inference mode): the deviation of one WHOLE step from the reference's
    own fp32 step is recorded (joints, losses, parameter gradients) and GATED at MIXED_STEP_GATES, and ten Adam steps must lower the loss."""
    import lt_train
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    G = np.load(os.path.join(golden_dir, "train_step.npz"))
    c, cfg, sd, inp = _train_case()
    m = VolumetricTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    m.train()
    m.train_precision = precision
    batch = {"cameras": _cameras(inp, c["B"]), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    np.random.seed(c["seed"] + 100)
    kp, feats, vols, conf, cuboids, cvs, bps = m(inp["images"].to(DEV), None, batch)
    d = (kp.detach().cpu().double() - torch.from_numpy(G["kp"]).double()).abs() / torch.from_numpy(G["kp"]).double().abs().clamp(min=1.0)
    gt, val = torch.from_numpy(G["gt"]).to(DEV), torch.from_numpy(G["val"]).to(DEV)
    mae = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val)
    ce = L.VolumetricCELoss()(cvs, vols, gt, val)
    (mae + 0.01 * ce).backward()
    named = dict(m.named_parameters())
    errs = []
    for n in G["names"]:
        n = str(n)
        if ZERO_GRAD.search(n):
            continue
        f = named[n].grad.detach().double().cpu().reshape(-1)
        sub = f[::max(1, f.numel() // 129)][:129]
        errs.append(float((sub - torch.from_numpy(G["g/" + n]).double()).abs().max()) / float(G["gn/" + n][1]))
    errs.sort()
    record("train/mixed precision (%s) one step vs the reference's fp32 step -- gated at MIXED_STEP_GATES (joints, median, p90): %s" % (
        "bf16 MFMA convolutions" if precision == "bf16" else precision, MIXED_STEP_GATES[precision]),
           {"joints_max_rel": float(d.max()), "mae": float(mae.detach()), "mae_reference": float(G["mae"]), "ce": float(ce.detach()), "ce_reference": float(G["ce"]),
            "parameter_gradient_err_median": errs[len(errs) // 2], "parameter_gradient_err_p90": errs[int(len(errs) * 0.9)], "parameter_gradient_err_max": errs[-1]})
    # gated at the level the mode achieves on this fixture (VERDICT r3 "next" 8): median 6.3 %, p90 15.8 % of each tensor's largest reference gradient
    # (act16 rounds every activation and activation gradient to bf16 as well: 7.7 % / 21.4 %; fp8v2v adds e4m3 operands in V2V)
    g_kp, g_med, g_p90 = MIXED_STEP_GATES[precision]
    assert abs(float(mae.detach()) - float(G["mae"])) <= 0.02 * abs(float(G["mae"])) and abs(float(ce.detach()) - float(G["ce"])) <= 0.02 * abs(float(G["ce"])), (float(mae.detach()), float(ce.detach()))
    assert float(d.max()) < g_kp and errs[len(errs) // 2] <= g_med and errs[int(len(errs) * 0.9)] <= g_p90, (float(d.max()), errs[len(errs) // 2], errs[int(len(errs) * 0.9)])
    # descent
    opt = lt_train.Adam(list(m.parameters()), lr=1e-4)
    gt2 = torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float().to(DEV)
    val2 = torch.ones(c["B"], 17, 1, device=DEV)
    losses = []
    for it in range(10):
        np.random.seed(0)
        kp, _, vols, _, _, cvs, _ = m(inp["images"].to(DEV), None, batch)
        loss = L.KeypointsMAELoss()(kp * 0.1, gt2 * 0.1, val2) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt2, val2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    record("train/mixed precision (%s) loss over 10 Adam steps" % precision, losses)
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("use_conf", [True, False], ids=["alg", "alg_no_conf"])
def test_whole_algebraic_training_step_vs_reference(golden_dir, use_conf):
    """AlgebraicTriangulationNet in training mode (round 3; the reference's loop for model_type "alg", train.py:189-243): forward (batch-statistics
    BatchNorm), KeypointsMSESmoothLoss(400) on keypoints * 0.1, backward through the DLT (torch.svd in the reference), the 2D soft-argmax, the
    alg_confidences head and the backbone, Adam -- every parameter's gradient, the running statistics and the parameters after the step against the
    reference's own CPU step (tests/golden/train_step_alg.npz, oracle/make_golden.py train_alg), gated within the reference's measured self-noise."""
    import lt_train
    from mvn.models import loss as L
    from mvn.models.triangulation import AlgebraicTriangulationNet
    # alg_no_conf: experiments/human36m/train/human36m_alg_no_conf.yaml -- no alg_confidences head, uniform view weights (train_step_alg_noconf.npz)
    G = np.load(os.path.join(golden_dir, "train_step_alg.npz" if use_conf else "train_step_alg_noconf.npz"))
    TA = "train-alg/" if use_conf else "train-alg-no-conf/"
    c = dict(nl=18, B=2, NV=3, H=128, seed=21)
    cfg = synth.alg_config(c["nl"], use_conf)
    cfg.model.heatmap_multiplier = 1.0
    cfg.model["heatmap_multiplier"] = 1.0
    sd = synth.make_state_dict(spec.alg_net_spec(c["nl"], 17, use_conf), seed=c["seed"], basic_block=True)
    inp = synth.make_inputs(c["B"], c["NV"], c["H"], seed=c["seed"], inside=False)
    m = AlgebraicTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    m.train()
    lr = float(G["lr"])
    opt = lt_train.Adam([p for p in m.parameters() if p.requires_grad], lr=lr)
    P = torch.from_numpy(G["P"]).to(DEV)
    kp3, kp2, hm, conf = m(inp["images"].to(DEV), P, {})
    kp_noise, loss_noise = float(G["kp_noise"]), float(G["loss_noise"])
    d = (kp3.detach().cpu().double() - torch.from_numpy(G["kp3"]).double()).abs() / torch.from_numpy(G["kp3"]).double().abs().clamp(min=1.0)
    record(TA + "step forward keypoints_3d (train-mode BN), rel with 1 mm floor", {"err": float(d.max()), "tol": 1e-4 + 2 * kp_noise, "reference_self_noise": kp_noise})
    assert float(d.max()) <= 1e-4 + 2 * kp_noise, float(d.max())
    check(TA + "step forward keypoints_2d", kp2.detach().cpu(), G["kp2"], 1e-4)
    check(TA + "step forward confidences", conf.detach().cpu(), G["conf"], 1e-4)
    check(TA + "step forward heatmaps", hm.detach().cpu().reshape(c["B"] * c["NV"], *hm.shape[2:])[:, :, ::2, ::2], G["hm_sub"], 1e-4)
    gt, val = torch.from_numpy(G["gt"]).to(DEV), torch.from_numpy(G["val"]).to(DEV)
    loss = L.KeypointsMSESmoothLoss(400)(kp3 * 0.1, gt * 0.1, val)
    assert abs(float(loss.detach()) - float(G["loss"])) <= (1e-4 + 2 * loss_noise) * float(G["loss"]), (float(loss.detach()), float(G["loss"]))
    opt.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    gn2, table, n_zero = 0.0, [], 0
    gnorm_ref = float(G["grad_norm"])
    for n in G["names"]:
        n = str(n)
        p = named[n]
        assert p.grad is not None, "no gradient for " + n
        gr = p.grad.detach().double().cpu()
        ref_norm, ref_max, _ = [float(v) for v in G["gn/" + n]]
        noise = float(G["noise/" + n])
        gn2 += float(gr.pow(2).sum())
        if noise > 0.05:          # a convolution bias in front of a training-mode BatchNorm: the exact gradient is 0, both sides hold rounding noise
            assert float(gr.abs().max()) <= 10 * ref_max + 1e-6 * gnorm_ref, (n, float(gr.abs().max()), ref_max)
            n_zero += 1
            continue
        # The gated reference is the reference's OWN step in fp64 (the same modules, .double()): its fp32 step is 1.4e-3 (median) / 1.2e-2 (max) of
        # each parameter's max gradient away from that -- torch.svd's backward in fp32 alone moves the tail's gradients by 1.2e-3 (stored as
        # svd32_rel) -- while lt_triangulate_dlt_bwd works in fp64 (3e-6 against fp64 autograd, test_softargmax2d_and_dlt_backward_...).  The
        # deviation from the fp32 step is recorded next to it.
        f = gr.reshape(-1)
        sub = f[::max(1, f.numel() // 129)][:129]
        n64, m64 = [float(v) for v in G["gn64/" + n]]
        e = float((sub - torch.from_numpy(G["g64/" + n]).double()).abs().max()) / m64
        en = abs(float(gr.norm()) - n64) / n64
        e32 = max(float((sub - torch.from_numpy(G["g/" + n]).double()).abs().max()) / ref_max, abs(float(gr.norm()) - ref_norm) / ref_norm)
        table.append((max(e, en) / (1e-3 + 4 * noise), max(e, en), noise, n, e32))
    table.sort(reverse=True)
    print("worst parameter gradients (err / gate, err vs the fp64 step, reference self-noise, name, err vs the fp32 step):", *["%.2f %.2e %.2e %s %.2e" % t for t in table[:8]], sep="\n  ")
    errs = sorted(t[1] for t in table)
    e32s = sorted(t[4] for t in table)
    record(TA + "step parameter gradients vs the reference's fp64 step (gate 1e-3 + 4 x reference self-noise)",
           {"worst_err_over_gate": table[0][0], "worst_err": errs[-1], "median_err": errs[len(errs) // 2], "parameters_compared": len(table),
            "zero_gradient_parameters": n_zero, "median_reference_self_noise": sorted(t[2] for t in table)[len(table) // 2],
            "vs_the_fp32_step_median": e32s[len(e32s) // 2], "vs_the_fp32_step_worst": e32s[-1], "reference_fp32_svd_backward_rel_err": float(G["svd32_rel"])})
    assert table[0][0] <= 1.0, table[:8]
    assert e32s[-1] <= 3e-2, e32s[-1]          # and never far from the fp32 step either (its own distance to fp64: up to 1.2e-2)
    assert len(G["no_grad"]) == 0 and len(table) > 60
    gn = float(np.sqrt(gn2))
    assert abs(gn - float(G["grad_norm64"])) <= 2e-3 * gnorm_ref and abs(gn - gnorm_ref) <= 1e-2 * gnorm_ref, (gn, gnorm_ref, float(G["grad_norm64"]))
    bufs = dict(m.named_buffers())
    w_rs = 0.0
    for key in G.files:
        if key.startswith("rs/"):
            b = bufs[key[3:]].detach().double().cpu().reshape(-1)
            sub = b[::max(1, b.numel() // 129)][:129]
            ref = torch.from_numpy(G[key]).double()
            w_rs = max(w_rs, float((sub - ref).abs().max() / ref.abs().max().clamp(min=1e-30)))
    record(TA + "step BatchNorm running statistics", {"err": w_rs, "tol": 1e-4})
    assert w_rs <= 1e-4, w_rs
    opt.step()
    torch.cuda.synchronize()
    w_p, w_name, n_known = 0.0, None, 0
    for n in G["names"]:
        n = str(n)
        if float(G["noise/" + n]) > 0.05:
            continue
        f = named[n].detach().double().cpu().reshape(-1)
        sub = f[::max(1, f.numel() // 129)][:129]
        ref = torch.from_numpy(G["p1/" + n]).double()
        gs = torch.from_numpy(G["g/" + n]).double().abs()
        known = gs > 100 * (float(G["noise/" + n]) + 1e-3) * float(G["gn/" + n][1])
        n_known += int(known.sum())
        e = float(((sub - ref).abs() * known).max()) / lr
        if e > w_p:
            w_p, w_name = e, n
    record(TA + "step parameters after Adam, worst |d| in units of lr", {"err": w_p, "tol": 2e-2, "name": w_name, "elements_compared": n_known})
    assert w_p <= 2e-2 and n_known > (300 if use_conf else 200), (w_p, w_name, n_known)          # (the no-conf fixture is the noisier one: 292 robust signs)
    # second step = the REPLAY of the recorded tape with the updated weights: finite, and the loss moves
    kp3b = m(inp["images"].to(DEV), P, {})[0]
    loss2 = L.KeypointsMSESmoothLoss(400)(kp3b * 0.1, gt * 0.1, val)
    opt.zero_grad(); loss2.backward(); opt.step()
    assert torch.isfinite(kp3b).all() and float(loss2.detach()) != float(loss.detach())


@pytest.mark.parametrize("y16", [False, True, "act16"], ids=["fp32_conv_outputs", "bf16_conv_outputs", "act16"])
def test_mixed_precision_step_tracks_fp32_on_a_bottleneck_backbone(y16, monkeypatch):
    """(``y16``: the opt-in LT_TRAIN_Y16=1 variant -- bf16 convolution outputs in front of BatchNorm, hence the whole set of bf16 forward kernels
    over LIVE weights at real layer shapes -- against the same gates.)
    VERDICT r2 weak 7: the mixed step (bf16 MFMA convolutions) was only compared with fp32 on the basic-block ResNet-18 fixture.  Here a
    BOTTLENECK backbone (ResNet-50: 1x1 reduce / 3x3 / 1x1 expand, strided downsample convolutions -- the code paths ResNet-152 takes), 2 samples x
    3 views of 128^2, 64^3 volume: same weights, inputs and rotations in both precisions.  Gated: the first-step loss within 1 %, three Adam steps
    within 5 % of each other.  RECORDED per parameter group: the cosine between the bf16 and the fp32 gradients.  Training-mode BatchNorm over few
    samples is ill-conditioned towards the bottom of V2V's hourglass (the 2^3 level normalises B x 8 values per channel; the reference's OWN
    gradients move by 1e-3 median under a 1e-6 change of the images at this batch size, DESIGN.md "Training step"), so a 2^-9 operand rounding
    decorrelates the deep V2V gradients in ANY implementation -- which is why the mixed step's loss curve leaves the fp32 one at small batches --
    while the backbone, whose BatchNorm layers see thousands of values, must keep its direction: gated on the backbone's median cosine."""
    import lt_train
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    cfg = synth.vol_config(50, 64, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(50, 17, False), seed=3, sharpen=60.0)
    inp = synth.make_inputs(2, 3, 128, seed=31, inside=False)
    batch = {"cameras": _cameras(inp, 2), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    gt = (torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float() + 25.0).to(DEV)
    val = torch.ones(2, 17, 1, device=DEV)

    def run(prec):
        m = VolumetricTriangulationNet(cfg, device=DEV)
        m.load_state_dict(sd, strict=True)
        m.to(DEV).train()
        m.train_precision = prec
        opt = lt_train.Adam([{"params": list(m.backbone.parameters())}, {"params": list(m.process_features.parameters()), "lr": 1e-3},
                             {"params": list(m.volume_net.parameters()), "lr": 1e-3}], lr=1e-4)
        losses, g0 = [], None
        for it in range(3):
            np.random.seed(500 + it)
            kp, _, vols, _, _, cvs, _ = m(inp["images"].to(DEV), None, batch)
            loss = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)
            opt.zero_grad()
            loss.backward()
            if it == 0:
                g0 = {n: p.grad.detach().double().cpu().reshape(-1) for n, p in m.named_parameters() if p.grad is not None}
            opt.step()
            losses.append(float(loss.detach()))
        return losses, g0

    if y16 is True:
        monkeypatch.setenv("LT_TRAIN_Y16", "1")
    else:
        monkeypatch.delenv("LT_TRAIN_Y16", raising=False)
    l32, g32 = run("fp32")
    l16, g16 = run("act16" if y16 == "act16" else "bf16")          # act16: bf16 activations and activation gradients as well (train_precision "act16")
    gtot = float(torch.cat(list(g32.values())).norm())

    def group(n):
        if n.startswith("backbone."):
            return "backbone"
        if n.startswith("process_features."):
            return "process_features"
        deep = any(k in n for k in ("res3", "res4", "res5", "mid_res", "upsample3", "upsample4", "upsample5"))
        return "v2v 8^3 and below" if deep else "v2v 64^3-16^3"
    groups = {}
    for n, a in g32.items():
        if float(a.norm()) < 1e-6 * max(1.0, gtot) or ZERO_GRAD.search(n):
            continue
        groups.setdefault(group(n), []).append(float((a @ g16[n]) / (a.norm() * g16[n].norm() + 1e-300)))
    stats = {k: {"tensors": len(v), "median_cosine": sorted(v)[len(v) // 2], "worst_cosine": min(v)} for k, v in groups.items()}
    record("train/mixed%s vs fp32 on a bottleneck backbone (ResNet-50, 64^3): losses and gradient cosines per group" % (" (act16: bf16 activations and gradients)" if y16 == "act16" else " (bf16 conv outputs)" if y16 else ""),
           {"losses_fp32": l32, "losses_bf16": l16, "groups": stats})
    print(stats, l32, l16)
    assert abs(l16[0] - l32[0]) <= 0.01 * abs(l32[0]), (l16, l32)
    assert all(abs(a - b) <= 0.05 * abs(a) for a, b in zip(l32, l16)), (l32, l16)
    assert stats["backbone"]["median_cosine"] > 0.9, stats


@pytest.mark.parametrize("B", [2, 8])
def test_act16_step_tracks_fp32_at_the_config2_shape(B):
    """The first step of train_precision "act16" against the fp32 step at the BASELINE config-2 SHAPE: ResNet-152, 4 views of 384 x 384, 64^3 volume, the
    fixtures' conditioned weights (oracle.synth.make_state_dict: variance-preserving filters, small last-BatchNorm gammas, sharpened output layer), same
    weights, inputs, rotations -- at 2 samples per step (8 images per BatchNorm) and at 8 samples per step, the batch bench.py's ``train_mixed`` leg
    times (VERDICT r4 "next" 1c).  What is ASSERTED (the numbers behind it are recorded in the parity report): loss within 1 % of the fp32 step's,
    V2V's gradient cosine > 0.99, and the backbone's gradient cosine > (CONTROL - 0.35), where CONTROL is the cosine of the fp32 step with its input
    images rounded to bf16 ONCE and nothing else changed (0.86 at B = 2: this network amplifies one 2^-9 perturbation that much through ~150
    batch-statistics BatchNorm layers).  A fixed "backbone cosine > 0.9" is NOT met at this shape (act16: 0.57 at B = 2) and is not what this test claims."""
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    cfg = synth.vol_config(152, 64, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(152, 17, False), seed=5, sharpen=60.0)
    inp = synth.make_inputs(B, 4, 384, seed=41, inside=False)
    batch = {"cameras": _cameras(inp, B), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    gt = (torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float() + 25.0).to(DEV)
    val = torch.ones(B, 17, 1, device=DEV)
    images = inp["images"].to(DEV)

    def run(prec, images=images):
        m = VolumetricTriangulationNet(cfg, device=DEV)
        m.load_state_dict(sd, strict=True)
        m.to(DEV).train()
        m.train_precision = prec
        np.random.seed(900)
        kp, _, vols, _, _, cvs, _ = m(images, None, batch)
        loss = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)
        loss.backward()
        torch.cuda.synchronize()
        g = {n: p.grad.detach().double().cpu().reshape(-1) for n, p in m.named_parameters() if p.grad is not None}
        out = (kp.detach().cpu().double(), float(loss.detach()), g)
        del m
        torch.cuda.empty_cache()
        return out

    k32, l32, g32 = run("fp32")
    k16, l16, g16 = run("act16")
    kb, lb, gb = run("bf16") if B == 2 else (k16, l16, g16)          # round 3's mode (bf16 MFMA over fp32 storage): what bf16 OPERANDS alone do to this network (B = 2 only)
    # CONTROL: the fp32 step itself with ONE 2^-9 perturbation -- the input images rounded to bf16, everything else exact fp32
    kc, lc, gc = run("fp32", images.bfloat16().float())
    assert set(g16) == set(g32) and all(bool(torch.isfinite(v).all()) for v in g16.values())

    def cosine(ga, gbb, names):
        a, b = torch.cat([ga[n] for n in names]), torch.cat([gbb[n] for n in names])
        return float((a @ b) / (a.norm() * b.norm() + 1e-300))
    bb = [n for n in sorted(g32) if n.startswith("backbone.")]
    v2v = [n for n in sorted(g32) if n.startswith("volume_net.") and not ZERO_GRAD.search(n)]
    late = [n for n in bb if "layer4" in n or "deconv" in n]          # the backbone layers behind few BatchNorm layers of backward
    stats = {"loss_fp32": l32, "loss_act16": l16, "loss_bf16": lb, "joints_max_rel_1mm_floor": float(((k16 - k32).abs() / k32.abs().clamp(min=1.0)).max()),
             "act16_vs_fp32": {"backbone": cosine(g32, g16, bb), "backbone layer4 + deconvs": cosine(g32, g16, late), "v2v": cosine(g32, g16, v2v)},
             "bf16_vs_fp32": {"backbone": cosine(g32, gb, bb), "backbone layer4 + deconvs": cosine(g32, gb, late), "v2v": cosine(g32, gb, v2v)},
             "act16_vs_bf16": {"backbone": cosine(gb, g16, bb), "v2v": cosine(gb, g16, v2v)},
             "CONTROL fp32 with bf16-rounded input images vs fp32": {"backbone": cosine(g32, gc, bb), "backbone layer4 + deconvs": cosine(g32, gc, late),
                                                                     "v2v": cosine(g32, gc, v2v), "loss": lc}}
    if B != 2:
        stats.pop("bf16_vs_fp32"); stats.pop("act16_vs_bf16"); stats.pop("loss_bf16")
    stats["asserted"] = "loss within 1 %, v2v cosine > 0.99, backbone cosine > CONTROL backbone cosine - 0.35"
    record("train/act16 vs fp32 at the config-2 shape (ResNet-152, 4 x 384^2, 64^3, %d samples, conditioned weights), first step" % B, stats)
    print(stats)
    # Gated: the loss (1 %), V2V's gradient direction (> 0.99), and the backbone's direction RELATIVE TO THE CONTROL: through ~150 batch-statistics BatchNorm
    # layers over 8 images this network amplifies ONE 2^-9 rounding of its input images -- everything else exact fp32 -- into a backbone gradient cosine of
    # 0.86 (measured here, every run); rounding at every layer (bf16 operands: 0.69; bf16 storage too: 0.57) cannot do better than that by much.  The
    # numbers are recorded in the parity report (DESIGN.md "Training step": the reference's own gradients move by 1e-3 median under a 1e-6 change of the images).
    ctl = stats["CONTROL fp32 with bf16-rounded input images vs fp32"]
    assert abs(l16 - l32) <= 0.01 * abs(l32) and stats["act16_vs_fp32"]["v2v"] > 0.99 and stats["act16_vs_fp32"]["backbone"] > ctl["backbone"] - 0.35, stats


# band of the reduced-precision runs' FINAL loss around the fp32 run's after TRAJ_STEPS Adam updates on one fixed batch (relative to the fp32 final loss), and the
# largest gap allowed anywhere along the run -- set from the measured curves (parity report "train/trajectory ...") with headroom
TRAJ_STEPS, TRAJ_FINAL_BAND, TRAJ_RUN_BAND = 20, {"act16": 0.05, "fp8v2v": 0.08}, {"act16": 0.08, "fp8v2v": 0.12}


def test_training_trajectory_same_batch_fp32_act16_fp8v2v():
    """VERDICT r4 "next" 1b: does reduced precision change where training goes?  The SAME run three times at the BASELINE config-2 shape (ResNet-152, 4 views
    of 384 x 384, 64^3 volume): identical conditioned weights, ONE fixed batch of 4 samples, identical cuboid rotations, the reference's loss (train.py:217-230) and
    three-group Adam (train.py:430-437), 20 updates -- in fp32 (the reference's precision), act16 and fp8v2v (BASELINE config 5 as named).  Gated: every run
    descends, and the reduced-precision runs end within TRAJ_FINAL_BAND of the fp32 run's final loss and stay within TRAJ_RUN_BAND of it at every step;
    the three curves go to the parity report (bench.py prints the same comparison on its own weights as ``train_trajectory``)."""
    import bench
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_models import _cameras
    B = 4
    cfg = synth.vol_config(152, 64, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(152, 17, False), seed=5, sharpen=60.0)
    inp = synth.make_inputs(B, 4, 384, seed=43, inside=False)
    batch = {"cameras": _cameras(inp, B), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    gt = (torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float() + 25.0).to(DEV)

    def make():
        m = VolumetricTriangulationNet(cfg, device=DEV)
        m.load_state_dict(sd, strict=True)
        return m
    curves = bench.trajectory(make, inp["images"].to(DEV), batch, gt, ["fp32", "act16", "fp8v2v"], TRAJ_STEPS, DEV)
    ref = curves["fp32"]
    stats = {"curves": {k: [round(v, 4) for v in c] for k, c in curves.items()},
             "final_rel_to_fp32": {k: (c[-1] - ref[-1]) / abs(ref[-1]) for k, c in curves.items()},
             "max_rel_gap_over_the_run": {k: max(abs(a - b) / abs(b) for a, b in zip(c, ref)) for k, c in curves.items()},
             "asserted": "every run: finite, final < first; act16 / fp8v2v: |final - fp32 final| <= %s, gap at every step <= %s (of the fp32 loss)" % (TRAJ_FINAL_BAND, TRAJ_RUN_BAND)}
    record("train/trajectory: %d Adam steps on one fixed batch of %d samples at the config-2 shape, fp32 vs act16 vs fp8v2v" % (TRAJ_STEPS, B), stats)
    print(stats)
    for k, c in curves.items():
        assert all(np.isfinite(c)) and c[-1] < c[0], (k, c)
    for k in ("act16", "fp8v2v"):
        assert abs(stats["final_rel_to_fp32"][k]) <= TRAJ_FINAL_BAND[k] and stats["max_rel_gap_over_the_run"][k] <= TRAJ_RUN_BAND[k], (k, stats)


@pytest.mark.parametrize("precision", ["bf16", "act16"])
def test_algebraic_training_step_in_mixed_precision_tracks_fp32(golden_dir, precision):
    """AlgebraicTriangulationNet through the mixed-precision tape (bf16 MFMA convolutions incl. the heatmap head and the alg_confidences head's
    layers, bf16 weight gradients over image octets -- 6 images: one ragged octet group): the first step's keypoints and loss against the fp32
    tape on the same weights, inputs and projection matrices; every parameter receives a finite gradient and the whole gradient keeps its
    direction (cosine with the fp32 step's gradient)."""
    from mvn.models import loss as L
    from mvn.models.triangulation import AlgebraicTriangulationNet
    G = np.load(os.path.join(golden_dir, "train_step_alg.npz"))
    cfg = synth.alg_config(18, True)
    cfg.model.heatmap_multiplier = 1.0
    cfg.model["heatmap_multiplier"] = 1.0
    sd = synth.make_state_dict(spec.alg_net_spec(18, 17, True), seed=21, basic_block=True)
    inp = synth.make_inputs(2, 3, 128, seed=21, inside=False)
    P = torch.from_numpy(G["P"]).to(DEV)
    gt, val = torch.from_numpy(G["gt"]).to(DEV), torch.from_numpy(G["val"]).to(DEV)

    g2 = torch.Generator().manual_seed(5)
    t2d = (torch.rand(2, 3, 17, 2, generator=g2) * 32).to(DEV)
    wc = torch.randn(2, 3, 17, generator=g2).to(DEV)

    def run(prec, loss2d=False):
        m = AlgebraicTriangulationNet(cfg, device=DEV)
        m.load_state_dict(sd, strict=True)
        m.to(DEV).train()
        m.train_precision = prec
        kp3, kp2, hm, conf = m(inp["images"].to(DEV), P, {})
        if loss2d:          # a loss on what the tape itself computes (2D soft-argmax + confidences), without the ill-conditioned DLT behind it
            loss = ((kp2 - t2d) ** 2).mean() + (conf * wc).sum()
        else:
            loss = L.KeypointsMSESmoothLoss(400)(kp3 * 0.1, gt * 0.1, val)
        loss.backward()
        torch.cuda.synchronize()
        g = {n: p.grad.detach().double().cpu().reshape(-1) for n, p in m.named_parameters() if p.grad is not None}
        return kp3.detach().cpu().double(), float(loss.detach()), g, (kp2.detach().cpu().double(), hm.detach().cpu().double(), conf.detach().cpu().double())

    k32, l32, g32, i32 = run("fp32")
    k16, l16, g16, i16 = run(precision)          # act16: bf16 activations / gradients too; the heatmap layer and the confidence head's sigmoid store fp32
    assert set(g16) == set(g32) and all(bool(torch.isfinite(v).all()) for v in g16.values())
    d_kp2 = float((i32[0] - i16[0]).abs().max())
    d_hm = float((i32[1] - i16[1]).abs().max() / i32[1].abs().max())
    d_conf = float((i32[2] - i16[2]).abs().max())
    d_kp3 = float(((k16 - k32).abs() / k32.abs().clamp(min=1.0)).max())
    a, b = torch.cat([g32[n] for n in sorted(g32)]), torch.cat([g16[n] for n in sorted(g32)])
    cos = float((a @ b) / (a.norm() * b.norm()))
    # act16 moves the 2D keypoints by ~0.09 px, which moves the triangulated points of this fixture (and with them d loss / d keypoints) arbitrarily:
    # the direction of the gradient is therefore ALSO measured under a loss on the tape's own outputs, and gated there
    _, _, ga, _ = run("fp32", loss2d=True)
    _, _, gb, _ = run(precision, loss2d=True)
    a2, b2 = torch.cat([ga[n] for n in sorted(ga)]), torch.cat([gb[n] for n in sorted(ga)])
    cos2d = float((a2 @ b2) / (a2.norm() * b2.norm()))
    # The DLT of a RANDOM-INIT network is ill-conditioned (all 2D estimates sit near the image centre, the three rays are nearly dependent): 0.05 px on
    # the 2D keypoints moves the triangulated points of this fixture by metres -- in any precision.  So the gates sit on what the tape computes
    # (heatmaps, 2D keypoints, confidences) and on the direction of the gradient; the 3D deviation and the losses are recorded.
    record("train-alg/mixed precision (%s) vs fp32, first step (ResNet-18 fixture, 2 samples x 3 views)" % precision,
           {"heatmaps_rel": d_hm, "keypoints_2d_max_abs_px": d_kp2, "confidences_max_abs": d_conf, "gradient_cosine": cos, "gradient_cosine_2d_loss": cos2d,
            "keypoints_3d_max_rel_1mm_floor (ill-conditioned DLT at random init)": d_kp3, "loss_fp32": l32, "loss_bf16": l16})
    print(d_hm, d_kp2, d_conf, cos, cos2d, d_kp3, l32, l16)
    assert d_hm < (3e-2 if precision == "bf16" else 6e-2) and d_kp2 < 0.25 and d_conf < 3e-2 and cos2d > 0.9 and (cos > 0.9 or precision != "bf16"), (d_hm, d_kp2, d_conf, cos, cos2d)
