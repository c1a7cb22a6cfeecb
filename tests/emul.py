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
        v = (acc + spec.bias) * spec.scale + spec.shift
        v = v[..., :spec.Cout]
        # the phase's outputs: (Do, Ho, Wo) positions starting at out_off with stride out_stride (parity phases of a transposed convolution;
        # depth slices of a split-K partial tensor)
        sl = (slice(None),) + tuple(slice(ph.out_off[i], ph.out_off[i] + spec.out_stride[i] * (n - 1) + 1, spec.out_stride[i])
                                    for i, n in enumerate((spec.Do, spec.Ho, spec.Wo)))
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


def run_plan_on_cpu(plan):
    """Interpret a DRY-RUN plan (lt_engine.PlanBuilder(dry_run=True)) on the CPU: convolutions through emulate_conv,
    the other ops through plain torch / the oracle.  Executes ops in recorded order over the recorded (aliasing!) buffers,
    so wiring and buffer-reuse mistakes show up as wrong numbers."""
    import torch.nn.functional as F
    from oracle import vol_oracle as O
    assert plan.dry_run
    for _, meta in plan.ops:
        kind, info = meta["kind"], meta["info"]
        if kind == "conv" and info.get("splitk_reduce"):      # lt_splitk_reduce: the S depth slices of the partial tensor, then the real epilogue
            sp, S, part = info["spec"], info["S"], info["part"].t.float()
            N, Do = sp.N, sp.Do
            acc = part.reshape(N, S, Do, sp.Ho, sp.Wo, sp.Cout).sum(dim=1)
            v = (acc + sp.bias[:sp.Cout]) * sp.scale[:sp.Cout] + sp.shift[:sp.Cout]
            if sp.flags & EPI_RELU_PRE:
                v = torch.relu(v)
            if info["res"] is not None:
                v = v + info["res"].t.float()
            if sp.flags & EPI_RELU_POST:
                v = torch.relu(v)
            info["y"].t.copy_(v)
        elif kind == "conv" and info.get("bneck"):           # lt_bottleneck_fwd: the three layers, the two inner tensors rounded to the plan dtype
            x = info["x"].t.float().clone()
            s1, s2, s3 = info["specs"]
            t1 = emulate_conv(s1, x).to(info["x"].t.dtype).float()
            t2 = emulate_conv(s2, t1).to(info["x"].t.dtype).float()
            info["y"].t.copy_(emulate_conv(s3, t2, x))
        elif kind == "conv" and info.get("bneck_ds"):        # lt_bottleneck_ds_fwd: as above, the residual = the downsample branch of x, added in fp32 (not rounded)
            x = info["x"].t.float().clone()
            s1, s2, s3, sd = info["specs"]
            t1 = emulate_conv(s1, x).to(info["x"].t.dtype).float()
            t2 = emulate_conv(s2, t1).to(info["x"].t.dtype).float()
            info["y"].t.copy_(emulate_conv(s3, t2, emulate_conv(sd, x)))
        elif kind == "conv" and info.get("xr"):              # lt_expand_reduce_fwd: expand (+ residual, ReLU) rounded to the plan dtype, then the next block's reduce
            s3, s1 = info["specs"]
            dt = info["x"].t.dtype
            y = emulate_conv(s3, info["x"].t.float().clone(), info["res"].t.float().clone()).to(dt)
            info["y"].t.copy_(y)
            info["t1"].t.copy_(emulate_conv(s1, y.float()))
        elif kind == "conv" and info.get("cat2"):            # lt_conv_cat2_fwd: a pointwise convolution over [x | x2 at the strided pixels] (weights: the scale-folded concatenation)
            st = info["stride2"]
            cat = torch.cat([info["x"].t.float(), info["x2"].t.float()[:, :, ::st, ::st, :]], dim=-1)
            info["y"].t.copy_(emulate_conv(info["spec"], cat))
        elif kind == "conv" and info.get("skip"):            # lt_conv_skip_fwd: the residual is the (scale-folded, bf16-rounded) skip convolution of a second tensor, in fp32
            sk = info["skip"]
            res = sk["x"].t.float() @ sk["w"].t()
            info["y"].t.copy_(emulate_conv(info["spec"], info["x"].t.float().clone(), res))
        elif kind == "conv":
            res = None if info["res"] is None else info["res"].t.float().clone()
            out = emulate_conv(info["spec"], info["x"].t.float().clone(), res)
            info["y"].t.copy_(out)
        elif kind == "pwchain":   # lt_pwchain_fwd: the layers one after the other, intermediates rounded to the plan dtype
            cur = info["x"].t.float().clone()
            for i, spec in enumerate(info["specs"]):
                cur = emulate_conv(spec, cur)
                if i + 1 < len(info["specs"]):
                    cur = cur.to(info["x"].t.dtype).float()
            info["y"].t.copy_(cur)
        elif kind == "stem":      # lt_stem_pool_fwd: conv + affine + ReLU rounded to the plan dtype, then the 3x3/2 max pool
            cur = emulate_conv(info["spec"], info["x"].t.float().clone()).to(info["x"].t.dtype).float()
            pooled = F.max_pool3d(cur.permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2), (0, 1, 1))
            info["y"].t.copy_(pooled.permute(0, 2, 3, 4, 1))
        elif kind == "maxpool":
            x = info["x"].t.float().permute(0, 4, 1, 2, 3)
            y = F.max_pool3d(x, tuple(info["k"]), tuple(info["s"]), tuple(info["p"]))
            info["y"].t.copy_(y.permute(0, 2, 3, 4, 1))
        elif kind == "avgpool":
            x = info["x"].t.float()
            info["y"].t.copy_(x.mean(dim=(1, 2, 3)).reshape(info["y"].t.shape))
        elif kind == "unproject":     # lt_unproject_grid_fwd: the cuboid grid (written to the returned coordinate tensor), then the gather
            o_pos, o_cen, o_rot = info["offs"]
            geo = info["geo"]
            B, V = info["coords"].shape[0], info["coords"].shape[1]
            pos = geo[o_pos:o_cen].reshape(B, 3); cen = geo[o_cen:o_rot].reshape(B, 3); rot = geo[o_rot:].reshape(B, 3, 3)
            idx = torch.arange(V, dtype=torch.float32)
            for b in range(B):
                ax = [pos[b, i] + torch.tensor(info["step"]) * idx for i in range(3)]
                grid = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1)
                g = (rot[b] @ (grid - cen[b]).reshape(-1, 3).t()).t().reshape(V, V, V, 3) + cen[b]
                if info["cmu"]:
                    g = g.permute(0, 2, 1, 3).flip(1)
                info["coords"][b] = g
            f = info["feats"].t.float()
            NV = info["NV"]
            hm = f.reshape(B, NV, f.shape[2], f.shape[3], f.shape[4]).permute(0, 1, 4, 2, 3)
            P = info["geo"][:B * NV * 12].reshape(B, NV, 3, 4)
            conf = info["conf"]
            agg = info["agg"]
            if agg == "conf_norm":
                conf = conf / conf.sum(dim=1, keepdim=True)
            out = O.unproject_heatmaps(hm, P, info["coords"], "conf" if agg.startswith("conf") else agg, conf)
            info["vol"].t.copy_(out.permute(0, 2, 3, 4, 1))
        elif kind == "features_out":  # layout launch into the caller's tensor: nothing to interpret (the tests read info["feats"] of the unprojection)
            pass
        elif kind == "softargmax3d":
            lg = info["logits"].t.float().permute(0, 4, 1, 2, 3) * info["mult"]
            kp, pr = O.integrate_tensor_3d_with_coordinates(lg, info["coords"], bool(info["softmax"]))
            info["kp"].copy_(kp); info["probs"].copy_(pr)
        else:
            raise KeyError(kind)
