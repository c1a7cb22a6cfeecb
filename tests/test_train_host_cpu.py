"""CPU: the HOST logic of the training step (lt_train.TrainTape) without a GPU, against torch autograd:

  * live weights -- the index maps that turn a Parameter into a layer's GEMM layouts (forward matrix, the transposed / flipped
    matrix of the input gradient, the parity phases of the stride-2 adjoints) are built by pushing a tensor of indices through
    lt_engine.make_conv_spec: gather(parameter, map) must equal make_conv_spec(parameter);
  * the three input-gradient formulations (stride 1: correlation with the flipped / transposed filter; stride 2: transposed
    convolution with output_padding 1; ConvTranspose: the strided convolution it is the adjoint of), executed by the CPU interpreter of
    the lt_conv_fwd contract (tests/emul.py), against torch.autograd;
  * the weight-gradient block layout dW[co][tap * Cin + ci] (and the transposed role) and the map that unpacks it into the
    Parameter's own layout, against torch.autograd.
What remains for the GPU suite (tests/test_gpu_train.py) is kernel == contract."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import lt_engine as E
from emul import emulate_conv

CASES = [  # nd, Cin, Cout, k, stride, pad, transposed, spatial
    (2, 8, 16, 3, 1, 1, False, (6, 8)),
    (2, 16, 8, 1, 1, 0, False, (5, 7)),
    (2, 8, 16, 3, 2, 1, False, (8, 6)),
    (2, 8, 32, 1, 2, 0, False, (6, 8)),
    (2, 16, 8, 4, 2, 1, True, (3, 4)),
    (3, 8, 8, 3, 1, 1, False, (4, 4, 6)),
    (3, 16, 8, 2, 2, 0, True, (2, 3, 2)),
    (3, 8, 17, 1, 1, 0, False, (2, 2, 3)),
]
IDS = ["nd%d_%dto%d_k%ds%dp%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "_T" if c[6] else "") for c in CASES]


def _cl(x):  # N,C,(D),H,W -> N,D,H,W,C
    if x.dim() == 4:
        x = x.unsqueeze(2)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _from_cl(y, nd):
    y = y.permute(0, 4, 1, 2, 3)
    return y[:, :, 0] if nd == 2 else y


def _index_like(w):
    return torch.arange(1, w.numel() + 1, dtype=torch.float32).reshape(w.shape)


def _gathered_spec(w, wt, in_shape, dtype=torch.float32, **kw):
    """What TrainTape._live_conv leaves in the layer's weight buffers: the index map from make_conv_spec(indices), gathered from the
    flat parameter."""
    idx = _index_like(w)
    if wt is not None:
        idx = wt(idx).contiguous()
    spec_i = E.make_conv_spec(idx, None, None, in_shape, kw.get("stride", 1), kw.get("pad", 0), dtype, kw.get("transposed", False), 0, kw.get("output_padding", 0))
    flat = w.reshape(-1)
    for ph in spec_i.phases:
        imap = ph.weight.round().to(torch.int64) - 1
        ph.weight = torch.where(imap >= 0, flat[imap.clamp(min=0)], torch.zeros(()))
    return spec_i


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16_layout"])
def test_live_weight_maps_equal_host_packing(case, dtype):
    nd, Cin, Cout, k, s, p, tr, sp = case
    g = torch.Generator().manual_seed(Cin + 3 * Cout + k)
    w = torch.randn(*((Cin, Cout) if tr else (Cout, Cin)), *([k] * nd), generator=g)
    N = 2
    cin_buf = max(Cin, E.min_cin_of(dtype))
    in_shape = (N, 1 if nd == 2 else sp[0], *(sp if nd == 2 else sp[1:]), cin_buf)
    want = E.make_conv_spec(w, None, None, in_shape, s, p, dtype, tr, 0)
    got = _gathered_spec(w, None, in_shape, dtype, stride=s, pad=p, transposed=tr)
    assert len(want.phases) == len(got.phases)
    for a, b in zip(want.phases, got.phases):
        assert torch.equal(a.weight, b.weight) and torch.equal(a.taps, b.taps)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_input_gradient_formulations(case):
    """dx of every layer kind through the lt_conv_fwd contract (emulate_conv) with the weights TrainTape would gather."""
    nd, Cin, Cout, k, s, p, tr, sp = case
    if Cout & (Cout - 1):
        pytest.skip("the tape pads dY to a power-of-two channel count first (covered on the GPU)")
    g = torch.Generator().manual_seed(11 * Cin + Cout + k)
    N = 2
    x = torch.randn(N, Cin, *sp, generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(*((Cin, Cout) if tr else (Cout, Cin)), *([k] * nd), generator=g, dtype=torch.float64)
    conv = {(2, False): F.conv2d, (3, False): F.conv3d, (2, True): F.conv_transpose2d, (3, True): F.conv_transpose3d}[(nd, tr)]
    y = conv(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * dy).sum().backward()
    wf = w.float()
    dyc = _cl(dy.float())
    if not tr and s == 1:
        spec = _gathered_spec(wf, lambda t: t.transpose(0, 1).flip(*range(2, 2 + nd)), tuple(dyc.shape), stride=1, pad=k - 1 - p)
    elif not tr:
        spec = _gathered_spec(wf, None, tuple(dyc.shape), stride=2, pad=p, transposed=True, output_padding=1)
    else:
        spec = _gathered_spec(wf, None, tuple(dyc.shape), stride=2, pad=p)
    dx = _from_cl(emulate_conv(spec, dyc), nd)
    assert dx.shape == x.grad.shape, (dx.shape, x.grad.shape)
    assert float((dx.double() - x.grad).abs().max()) <= 2e-5 * float(x.grad.abs().max())


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_weight_gradient_block_layout_and_unpack_map(case):
    """dW[co][tap * Cin + ci] = sum over output pixels of dY[pix][co] * X[pix * stride + tap - pad][ci] (ConvTranspose: the roles of the
    layer's input and of dY swapped), then the unpack map of TrainTape._conv_bwd -> the Parameter's layout; vs torch.autograd."""
    nd, Cin, Cout, k, s, p, tr, sp = case
    g = torch.Generator().manual_seed(5 * Cin + Cout + 2 * k)
    N = 2
    x = torch.randn(N, Cin, *sp, generator=g, dtype=torch.float64)
    w = torch.randn(*((Cin, Cout) if tr else (Cout, Cin)), *([k] * nd), generator=g, dtype=torch.float64).requires_grad_(True)
    conv = {(2, False): F.conv2d, (3, False): F.conv3d, (2, True): F.conv_transpose2d, (3, True): F.conv_transpose3d}[(nd, tr)]
    y = conv(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * dy).sum().backward()
    ks3 = (1,) + (k,) * 2 if nd == 2 else (k,) * 3
    st3 = (1, s, s) if nd == 2 else (s, s, s)
    pd3 = (0, p, p) if nd == 2 else (p, p, p)
    xc, dyc = _cl(x), _cl(dy)                       # N, D, H, W, C
    if not tr:
        rows_t, gath_t, cr, cg = dyc, xc, Cout, Cin       # "dy" role: rows; "x" role: gathered at row * stride + tap - pad
    else:
        rows_t, gath_t, cr, cg = xc, dyc, Cin, Cout
    cop = E.cout_pad_of(cr)
    kp = ks3[0] * ks3[1] * ks3[2] * cg
    dw = torch.zeros(cop, kp, dtype=torch.float64)
    Nn, Dr, Hr, Wr, _ = rows_t.shape
    _, Dg, Hg, Wg, _ = gath_t.shape
    t = 0
    for a in range(ks3[0]):
        for b in range(ks3[1]):
            for c in range(ks3[2]):
                for od in range(Dr):
                    i_d = od * st3[0] - pd3[0] + a
                    if not 0 <= i_d < Dg:
                        continue
                    for oh in range(Hr):
                        i_h = oh * st3[1] - pd3[1] + b
                        if not 0 <= i_h < Hg:
                            continue
                        for ow in range(Wr):
                            i_w = ow * st3[2] - pd3[2] + c
                            if not 0 <= i_w < Wg:
                                continue
                            dw[:cr, t * cg:(t + 1) * cg] += rows_t[:, od, oh, ow, :].t() @ gath_t[:, i_d, i_h, i_w, :]
                t += 1
    ar = torch.arange(cop * kp).reshape(cop, kp)
    if not tr:
        imap = ar[:Cout].reshape(Cout, *ks3, Cin)[..., :Cin].permute(0, 4, 1, 2, 3)
    else:
        imap = ar[:Cin].reshape(Cin, *ks3, Cout).permute(0, 4, 1, 2, 3)
    imap = (imap[:, :, 0] if nd == 2 else imap).contiguous().reshape(-1)
    gw = dw.reshape(-1)[imap].reshape(w.shape)
    assert float((gw - w.grad).abs().max()) <= 1e-10 * float(w.grad.abs().max())


def test_adam_groups_follow_the_reference_config():
    """train.py:430-437: backbone at opt.lr, process_features / volume_net at their own rates when the YAML names them."""
    import lt_train
    from mvn.models.triangulation import VolumetricTriangulationNet
    from oracle import synth
    m = VolumetricTriangulationNet(synth.vol_config(18, 32), device="cpu")

    class Opt:
        lr = 1e-4
    g = lt_train.adam_groups(m, Opt())
    assert [x.get("lr", 1e-4) for x in g] == [1e-4, 1e-4, 1e-4]
    Opt.process_features_lr, Opt.volume_net_lr = 1e-3, 2e-3
    g = lt_train.adam_groups(m, Opt())
    assert "lr" not in g[0] and g[1]["lr"] == 1e-3 and g[2]["lr"] == 2e-3
    n = sum(len(x["params"]) for x in g)
    assert n == len(list(m.parameters()))          # every parameter in exactly one group
    opt = lt_train.Adam(g, lr=Opt.lr)
    assert [pg["lr"] for pg in opt.param_groups] == [1e-4, 1e-3, 2e-3]


def _pack_n8(t):
    """CPU restatement of lt_pack_n8_bf16 (include/lt_hip.h): [N, D, H, W, C] -> [ceil(N / 8), D, H, W, C, 8], element (g, ..., c, e) = the bf16-rounded
    value of image 8 g + e, zero past N."""
    N = t.shape[0]
    G = (N + 7) // 8
    out = torch.zeros(G * 8, *t.shape[1:], dtype=torch.float32)
    out[:N] = t.float().bfloat16().float()
    return out.reshape(G, 8, *t.shape[1:]).permute(0, 2, 3, 4, 5, 1).contiguous()


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("N", [3, 11])
def test_weight_gradient_from_image_octets(case, N):
    """The formulation behind lt_conv_wgrad_bf16 (csrc/wgrad16.hip): with the operands packed as octets of 8 IMAGES per (pixel, channel) a tap
    shifts whole octets and padding is valid or not for a whole octet, so dW[co][tap * Cin + ci] = sum over octet rows (g, od, oh, ow) of
    <dY octet [co], X octet at the tap-shifted pixel [ci]> -- the 8-term dot product IS one lane's share of a 16-bit MFMA's K dimension.  Zero
    images past N contribute nothing.  Against torch.autograd on the bf16-rounded operands, ragged N (one / two octet groups), strides and the
    transposed role."""
    nd, Cin, Cout, k, s, p, tr, sp = case
    g = torch.Generator().manual_seed(7 * Cin + Cout + k + N)
    rd = lambda t: t.float().bfloat16().double()
    x = rd(torch.randn(N, Cin, *sp, generator=g))
    w = torch.randn(*((Cin, Cout) if tr else (Cout, Cin)), *([k] * nd), generator=g, dtype=torch.float64).requires_grad_(True)
    conv = {(2, False): F.conv2d, (3, False): F.conv3d, (2, True): F.conv_transpose2d, (3, True): F.conv_transpose3d}[(nd, tr)]
    y = conv(x, w, None, stride=s, padding=p)
    dy = rd(torch.randn(y.shape, generator=g))
    (y * dy).sum().backward()
    ks3 = (1,) + (k,) * 2 if nd == 2 else (k,) * 3
    st3 = (1, s, s) if nd == 2 else (s, s, s)
    pd3 = (0, p, p) if nd == 2 else (p, p, p)
    xc, dyc = _cl(x), _cl(dy)
    rows_t, gath_t, cr, cg = (dyc, xc, Cout, Cin) if not tr else (xc, dyc, Cin, Cout)
    ro, ga = _pack_n8(rows_t).double(), _pack_n8(gath_t).double()          # [G, D, H, W, C, 8]
    G = ro.shape[0]
    assert G == (N + 7) // 8 and float(ro[-1, ..., (N - 1) % 8 + 1:].abs().sum()) == 0.0          # images past N are zeros
    kp = ks3[0] * ks3[1] * ks3[2] * cg
    dw = torch.zeros(cr, kp, dtype=torch.float64)
    _, Dr, Hr, Wr, _, _ = ro.shape
    _, Dg, Hg, Wg, _, _ = ga.shape
    t = 0
    for a in range(ks3[0]):
        for b in range(ks3[1]):
            for c in range(ks3[2]):
                for od in range(Dr):
                    i_d = od * st3[0] - pd3[0] + a
                    for oh in range(Hr):
                        i_h = oh * st3[1] - pd3[1] + b
                        for ow in range(Wr):
                            i_w = ow * st3[2] - pd3[2] + c
                            if 0 <= i_d < Dg and 0 <= i_h < Hg and 0 <= i_w < Wg:          # one validity test per OCTET
                                dw[:, t * cg:(t + 1) * cg] += torch.einsum("gce,gke->ck", ro[:, od, oh, ow], ga[:, i_d, i_h, i_w])
                t += 1
    ar = torch.arange(cr * kp).reshape(cr, kp)
    if not tr:
        imap = ar.reshape(Cout, *ks3, Cin).permute(0, 4, 1, 2, 3)
    else:
        imap = ar.reshape(Cin, *ks3, Cout).permute(0, 4, 1, 2, 3)
    imap = (imap[:, :, 0] if nd == 2 else imap).contiguous().reshape(-1)
    gw = dw.reshape(-1)[imap].reshape(w.shape)
    assert float((gw - w.grad).abs().max()) <= 1e-10 * float(w.grad.abs().max())


def test_octet_pack_and_workspace_sizes_through_the_c_abi():
    """Host-side entry points of the bf16 weight gradients (no GPU needed): lt_pack_n8_bf16_bytes = ceil(N / 8) * P * C * 16, and the workspace of
    lt_conv_wgrad_bf16 is positive, bounded (64 MiB) and covers at least one fp32 copy of dW."""
    import lt_hip as H
    lib = H.lib()
    assert lib.lt_pack_n8_bf16_bytes(8, 576, 256) == 1 * 576 * 256 * 16
    assert lib.lt_pack_n8_bf16_bytes(11, 100, 17) == 2 * 100 * 17 * 16
    assert lib.lt_pack_n8_bf16_bytes(0, 100, 17) == 0
    for rows, cop, kp in ((2304, 256, 2304), (2304, 1024, 256), (262144, 32, 864), (262144, 16, 10976), (8, 128, 3456), (147456, 64, 392)):
        ws = lib.lt_conv_wgrad_bf16_workspace(rows, cop, kp)
        assert cop * kp * 4 <= ws <= (64 << 20), (rows, cop, kp, ws)


def _mfma_16x16x32(A, B):
    """v_mfma_f32_16x16x32_bf16 as the kernels use it: A[lane] / B[lane] = the lane's 8 K values (lane % 16 = row m of A / column n of B, lane // 16 = K
    group); returns D[lane][r] = D[m = 4 (lane // 16) + r][n = lane % 16] = sum over K groups q and k of A[16 q + m][k] B[16 q + n][k]."""
    A4, B4 = A.reshape(4, 16, 8), B.reshape(4, 16, 8)          # [q][m or n][k]
    D = np.einsum("qmk,qnk->mn", A4, B4)
    out = np.zeros((64, 4))
    for lane in range(64):
        for r in range(4):
            out[lane, r] = D[4 * (lane // 16) + r, lane % 16]
    return out


@pytest.mark.parametrize("shape", [(11, 5, 3, 8, 24, 1, 0), (9, 4, 6, 8, 40, 3, 1), (3, 6, 5, 16, 136, 3, 1)], ids=["1x1_ragged", "3x3_pad", "3x3_two_co_tiles"])
def test_weight_gradient_with_the_transpose_in_registers(shape):
    """The formulation behind lt_conv_wgrad_bf16_nhwc (csrc/wgrad16.hip, conv_wgrad16u_kernel<8, 4>), lane by lane: a lane of the 16x16x32 MFMA
    (i = lane % 16, q = lane // 16) reads 8 consecutive output channels of dY and 4 consecutive (tap, ci) columns of X for the EIGHT IMAGES of its
    pixel m + q; the 8 x 8 (8 x 4) transpose gives it operand a = channel 8 i + a of the eight images.  So accumulator (a, b), register r of lane l is
    dW[co0 + 8 (4 (l // 16) + r) + a][k0 + 4 (l % 16) + b]: a row permutation of the tile that only the epilogue knows.  K of one MFMA = 4 pixels x 8
    images; images past N and padded taps are zero operands.  Against torch.autograd, 2D, stride 1."""
    N, Hh, W, Cin, Cout, k, p = shape
    g = torch.Generator().manual_seed(N + Cout + k)
    x = torch.randn(N, Cin, Hh, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=1, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * dy).sum().backward()
    Ho, Wo = y.shape[2:]
    xc, dyc = x.permute(0, 2, 3, 1).numpy(), dy.permute(0, 2, 3, 1).numpy()          # channels-last, as the step holds them
    ACH, BCH = 8, 4
    G, ntaps = (N + 7) // 8, k * k
    kp = ntaps * Cin
    cop = (Cout + 127) // 128 * 128
    dw = np.zeros((cop, kp))
    M = G * Ho * Wo          # octet rows
    for co0 in range(0, cop, 16 * ACH):
        for k0 in range(0, kp, 16 * BCH):
            acc = np.zeros((ACH, BCH, 64, 4))
            for m0 in range(0, M, 4):          # one step = one MFMA per (a, b): 4 octet rows x 8 images
                A, B = np.zeros((ACH, 64, 8)), np.zeros((BCH, 64, 8))
                for lane in range(64):
                    i, q = lane % 16, lane // 16
                    m = m0 + q
                    if m >= M:
                        continue
                    gi, r = divmod(m, Ho * Wo)
                    oh, ow = divmod(r, Wo)
                    co, kc = co0 + ACH * i, k0 + BCH * i
                    tap, ci = divmod(kc, Cin)
                    ih, iw = oh - p + tap // k, ow - p + tap % k
                    for e in range(8):
                        n = 8 * gi + e
                        if n >= N:
                            continue          # zero operand: the image does not exist
                        if co < Cout:
                            A[:, lane, e] = dyc[n, oh, ow, co:co + ACH] if co + ACH <= Cout else 0.0
                        if tap < ntaps and 0 <= ih < Hh and 0 <= iw < W:
                            B[:, lane, e] = xc[n, ih, iw, ci:ci + BCH]
                for a in range(ACH):
                    for b in range(BCH):
                        acc[a, b] += _mfma_16x16x32(A[a], B[b])
            for a in range(ACH):
                for b in range(BCH):
                    for lane in range(64):
                        for r in range(4):
                            row, col = co0 + ACH * (4 * (lane // 16) + r) + a, k0 + BCH * (lane % 16) + b
                            if row < cop and col < kp:
                                dw[row, col] = acc[a, b, lane, r]
    want = w.grad.permute(0, 2, 3, 1).reshape(Cout, kp).numpy()          # [co][tap * Cin + ci]
    assert Cout % ACH == 0 and float(np.abs(dw[:Cout] - want).max()) <= 1e-10 * float(np.abs(want).max())
    assert float(np.abs(dw[Cout:]).max()) == 0.0 if cop > Cout else True


def test_unpacked_weight_gradient_coverage_through_the_c_abi(monkeypatch):
    """lt_conv_wgrad_bf16_nhwc_ok (host-only): the 2D layers of the backbone and the V2V 3^3 bricks with whole 32-channel blocks are covered; the 7^3
    layer, the 16-channel brick layer, 17 joints and misaligned rows are not (the tape packs octets for those); LT_WGRAD16_PACKED=1 turns it off."""
    import ctypes as C
    import lt_hip as H
    lib = H.lib()
    i3 = lambda *v: (C.c_int32 * 3)(*v)
    ok = lambda N, D, Hh, W, Cin, ldx, Do, Ho, Wo, st, pd, Cout, ldy, cop, kp, nt: lib.lt_conv_wgrad_bf16_nhwc_ok(N, D, Hh, W, Cin, ldx, Do, Ho, Wo, i3(*st), i3(*pd), Cout, ldy, cop, kp, nt)
    u, z = (1, 1, 1), (0, 0, 0)
    assert ok(32, 1, 24, 24, 256, 256, 1, 24, 24, u, z, 1024, 1024, 1024, 256, 1) == 1                  # layer3 expand
    assert ok(32, 1, 24, 24, 256, 256, 1, 24, 24, u, (0, 1, 1), 256, 256, 256, 2304, 9) == 1            # layer3 3x3
    assert ok(32, 1, 96, 96, 256, 256, 1, 96, 96, u, z, 64, 64, 64, 256, 1) == 1                        # layer1 reduce: the 64 x 128 wave tile
    assert ok(32, 1, 384, 384, 8, 8, 1, 192, 192, (1, 2, 2), (0, 3, 3), 64, 64, 64, 392, 49) == 1       # the stem on its 8-channel input
    assert ok(8, 64, 64, 64, 32, 32, 64, 64, 64, u, (1, 1, 1), 32, 32, 32, 864, 27) == 1                # V2V 3^3 bricks, staged from the tensors
    assert ok(8, 64, 64, 64, 16, 16, 64, 64, 64, u, (1, 1, 1), 32, 32, 32, 432, 27) == 0                # ... 16 input channels: packed
    assert ok(8, 64, 64, 64, 32, 32, 64, 64, 64, u, (3, 3, 3), 16, 16, 16, 10976, 343) == 0             # the 7^3 front layer: packed
    assert ok(8, 64, 64, 64, 32, 32, 64, 64, 64, u, z, 17, 17, 32, 32, 1) == 0                          # 17 joints
    assert ok(8, 64, 64, 64, 32, 32, 64, 64, 64, u, z, 32, 32, 32, 32, 1) == 1                          # ... widened to 32 by the tape
    assert ok(32, 1, 24, 24, 256, 256, 1, 24, 24, u, z, 1024, 1028, 1024, 256, 1) == 0                  # rows of dY not 16-byte aligned
    assert ok(32, 1, 24, 24, 24, 24, 1, 24, 24, u, z, 128, 128, 128, 24, 1) == 0                        # Cin not a power of two
    assert ok(64, 128, 128, 128, 32, 32, 128, 128, 128, u, z, 64, 64, 64, 32, 1) == 0                   # 2^31 elements: 32-bit offsets
    monkeypatch.setenv("LT_WGRAD16_PACKED", "1")
    assert ok(32, 1, 24, 24, 256, 256, 1, 24, 24, u, z, 1024, 1024, 1024, 256, 1) == 0

