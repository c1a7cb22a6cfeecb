"""CPU interpreter of lt_engine.ConvSpec = the semantics lt_conv_fwd implements (include/lt_hip.h), written
with plain torch indexing.  It lets the CPU suite verify every piece of HOST logic (weight re-packing, tap
tables, transposed-conv phase decomposition, BN folding, epilogue flags) against torch's own conv ops
without a GPU; the GPU suite then only has to prove kernel == this contract."""
import torch

EPI_RELU_PRE, EPI_RELU_POST, EPI_STORE_F32, EPI_SIGMOID = 1, 2, 4, 8


def emulate_conv(spec, x, residual=None):
    """x: [N,D,H,W,Cin] fp32 channels-last.  Returns y [N,OD,OH,OW,Cout] fp32."""
    N, D, Hh, W, Cin = x.shape
    assert (N, D, Hh, W, Cin) == (spec.N, spec.D, spec.H, spec.W, spec.Cin)
    y = torch.full((N, spec.OD, spec.OH, spec.OW, spec.Cout), float("nan"))
    xf = x.reshape(-1)
    od = torch.arange(spec.Do); oh = torch.arange(spec.Ho); ow = torch.arange(spec.Wo); nn_ = torch.arange(N)
    n_g, d_g, h_g, w_g = torch.meshgrid(nn_, od, oh, ow, indexing="ij")
    for ph in spec.phases:
        acc = torch.zeros(N, spec.Do, spec.Ho, spec.Wo, spec.cout_pad)
        assert ph.weight.shape == (spec.cout_pad, spec.k_pad)
        for t in range(ph.taps.shape[0]):
            dd, dh, dw, off = (int(v) for v in ph.taps[t])
            assert off == ((dd * Hh + dh) * W + dw) * Cin, "tap element offset"
            i_d = d_g * spec.stride[0] - spec.pad[0] + dd
            i_h = h_g * spec.stride[1] - spec.pad[1] + dh
            i_w = w_g * spec.stride[2] - spec.pad[2] + dw
            ok = (i_d >= 0) & (i_d < D) & (i_h >= 0) & (i_h < Hh) & (i_w >= 0) & (i_w < W)
            # the kernel addresses x + base + tap_offset + c with base from (od*s - p): reproduce that arithmetic
            base = (((n_g * D + (d_g * spec.stride[0] - spec.pad[0])) * Hh + (h_g * spec.stride[1] - spec.pad[1])) * W
                    + (w_g * spec.stride[2] - spec.pad[2])) * Cin
            idx = (base + off).clamp(0, xf.numel() - Cin)
            g = xf[idx.unsqueeze(-1) + torch.arange(Cin)]
            g = torch.where(ok.unsqueeze(-1), g, torch.zeros(()))
            wt = ph.weight[:, t * Cin:(t + 1) * Cin]
            acc += g @ wt.t()
        # beyond the real K the packed weights must be zero
        assert float(ph.weight[:, ph.taps.shape[0] * Cin:].abs().sum()) == 0.0
        v = acc * spec.scale + spec.shift
        v = v[..., :spec.Cout]
        sl = (slice(None), slice(ph.out_off[0], None, spec.out_stride[0]), slice(ph.out_off[1], None, spec.out_stride[1]),
              slice(ph.out_off[2], None, spec.out_stride[2]))
        if spec.flags & EPI_RELU_PRE:
            v = torch.relu(v)
        if residual is not None:
            v = v + residual[sl]
        if spec.flags & EPI_RELU_POST:
            v = torch.relu(v)
        if spec.flags & EPI_SIGMOID:
            v = torch.sigmoid(v)
        y[sl] = v
    assert not torch.isnan(y).any(), "phases do not cover the output"
    return y
