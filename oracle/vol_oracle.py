"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- NOT a product path.

Functional CPU restatement of the reference's volumetric / algebraic triangulation forward, driven
directly by a flat ``state_dict`` (no nn.Module tree).  torch-CPU fp32 tensor math, numpy fp64 host
geometry -- the same arithmetic types the reference uses.  Every function cites what it follows in
/root/reference (paths relative to that root).

Pinned by tests/golden/*.npz = outputs of the reference itself (oracle/make_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .spec import RESNET_SPEC, V2V_ENCDEC

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------------
# host geometry (numpy fp64, like the reference)
# --------------------------------------------------------------------------------------------
def resized_projection(K, R, t, image_shape, heatmap_shape):
    """Camera.update_after_resize + Camera.projection, mvn/utils/multiview.py:33-52.

    K (...,3,3), R (...,3,3), t (...,3,1) fp64 -> P (...,3,4) fp64 = K' [R|t] with
    fx,cx scaled by new_w/w and fy,cy by new_h/h.
    """
    K = np.array(K, dtype=np.float64, copy=True)
    h, w = image_shape
    nh, nw = heatmap_shape
    K[..., 0, 0] = K[..., 0, 0] * (nw / w)
    K[..., 1, 1] = K[..., 1, 1] * (nh / h)
    K[..., 0, 2] = K[..., 0, 2] * (nw / w)
    K[..., 1, 2] = K[..., 1, 2] * (nh / h)
    Rt = np.concatenate([np.asarray(R, np.float64), np.asarray(t, np.float64)], axis=-1)
    return K @ Rt


def rotation_matrix(axis, theta):
    """Euler-Rodrigues matrix, mvn/utils/volumetric.py:87-99 (fp64)."""
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / math.sqrt(float(np.dot(axis, axis)))
    a = math.cos(theta / 2.0)
    b, c, d = -axis * math.sin(theta / 2.0)
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * (b * c + a * d), 2 * (b * d - a * c)],
        [2 * (b * c - a * d), a * a + c * c - b * b - d * d, 2 * (c * d + a * b)],
        [2 * (b * d + a * c), 2 * (c * d - a * b), a * a + d * d - b * b - c * c]])


def coord_volume(base_point, cuboid_side, V, theta=0.0, axis=(0, 0, 1), cmu_transfer=False):
    """One sample's (V,V,V,3) fp32 voxel-centre grid, mvn/models/triangulation.py:298-339.

    fp32 arithmetic in the reference's order: f32(position) + f32(side/(V-1)) * idx, minus the
    f32 centre, times R (fp32), plus the centre.  'ij' meshgrid: axis 0 <-> world x.
    """
    base_point = np.asarray(base_point, dtype=np.float64)
    position = base_point - cuboid_side / 2.0
    step = torch.tensor(cuboid_side / (V - 1), dtype=torch.float32)
    idx = torch.arange(V, dtype=torch.float32)
    ax = [torch.tensor(position[i], dtype=torch.float32) + step * idx for i in range(3)]
    grid = torch.stack(torch.meshgrid(ax[0], ax[1], ax[2], indexing="ij"), dim=-1)
    center = torch.from_numpy(base_point).float()
    rot = torch.from_numpy(rotation_matrix(axis, theta)).float()
    g = (grid - center).reshape(-1, 3)
    g = rot.mm(g.t()).t().reshape(V, V, V, 3) + center
    if cmu_transfer:  # triangulation.py:336-339
        g = g.permute(0, 2, 1, 3).flip(1)
    return g.contiguous()


# --------------------------------------------------------------------------------------------
# functional ops
# --------------------------------------------------------------------------------------------
def unproject_heatmaps(heatmaps, proj_matricies, coord_volumes, method="sum", vol_confidences=None):
    """mvn/utils/op.py:99-166 + mvn/utils/multiview.py:55-110, vectorised over views.

    heatmaps (B,NV,C,h,w), proj (B,NV,3,4), coord_volumes (B,V,V,V,3) -> (B,C,V,V,V).
    Quirks kept: x normalised by h and y by w (op.py:128-129); align_corners=True; zero padding;
    z<=0 samples zeroed AFTER sampling and still entering the view softmax (op.py:141,157-162).
    """
    B, NV, C, h, w = heatmaps.shape
    vshape = tuple(coord_volumes.shape[1:4])
    out = torch.zeros(B, C, *vshape)
    for b in range(B):
        X = coord_volumes[b].reshape(-1, 3)
        Xh = torch.cat([X, torch.ones(X.shape[0], 1)], dim=1)
        per_view = []
        for v in range(NV):
            p = Xh @ proj_matricies[b, v].t()
            z = p[:, 2].clone()
            invalid = z <= 0.0
            z[z == 0.0] = 1.0
            u = p[:, 0] / z
            vv = p[:, 1] / z
            grid = torch.stack([2 * (u / h - 0.5), 2 * (vv / w - 0.5)], dim=1)
            s = F.grid_sample(heatmaps[b, v][None], grid[None, :, None, :], mode="bilinear",
                              padding_mode="zeros", align_corners=True)
            s = s.reshape(C, -1).clone()
            s[:, invalid] = 0.0
            per_view.append(s)
        vals = torch.stack(per_view, dim=0)  # (NV,C,N)
        if method.startswith("conf"):
            agg = (vals * vol_confidences[b].reshape(NV, C, 1)).sum(0)
        elif method == "sum":
            agg = vals.sum(0)
        elif method == "max":
            agg = vals.max(0)[0]
        elif method == "softmax":
            agg = (vals * torch.softmax(vals, dim=0)).sum(0)
        else:
            raise ValueError("Unknown volume_aggregation_method: {}".format(method))
        out[b] = agg.reshape(C, *vshape)
    return out


def integrate_tensor_3d_with_coordinates(volumes, coord_volumes, softmax=True):
    """mvn/utils/op.py:84-96: softmax (or ReLU) over all voxels, then expectation of the coords."""
    B, J = volumes.shape[:2]
    flat = volumes.reshape(B, J, -1)
    flat = torch.softmax(flat, dim=2) if softmax else torch.relu(flat)
    coords = torch.einsum("bjn,bnc->bjc", flat, coord_volumes.reshape(B, -1, 3))
    return coords, flat.reshape(volumes.shape)


def integrate_tensor_2d(heatmaps, softmax=True):
    """mvn/utils/op.py:11-47: 2D soft-argmax; x first, then y."""
    N, J, h, w = heatmaps.shape
    flat = heatmaps.reshape(N, J, -1)
    flat = torch.softmax(flat, dim=2) if softmax else torch.relu(flat)
    hm = flat.reshape(N, J, h, w)
    mass_x = hm.sum(dim=2)
    mass_y = hm.sum(dim=3)
    x = (mass_x * torch.arange(w, dtype=torch.float32)).sum(dim=2, keepdim=True)
    y = (mass_y * torch.arange(h, dtype=torch.float32)).sum(dim=2, keepdim=True)
    if not softmax:
        x = x / mass_x.sum(dim=2, keepdim=True)
        y = y / mass_y.sum(dim=2, keepdim=True)
    return torch.cat([x, y], dim=2), hm


def triangulate_batch_of_points(proj_matricies, points, confidences=None):
    """mvn/utils/multiview.py:141-183: per (sample, joint) confidence-weighted DLT via SVD."""
    B, NV, J = points.shape[:3]
    out = torch.zeros(B, J, 3)
    for b in range(B):
        P = proj_matricies[b]
        for j in range(J):
            pt = points[b, :, j, :]
            c = confidences[b, :, j] if confidences is not None else torch.ones(NV)
            A = P[:, 2:3].expand(NV, 2, 4) * pt.reshape(NV, 2, 1) - P[:, :2]
            A = A * c.reshape(-1, 1, 1)
            _, _, vh = torch.svd(A.reshape(-1, 4))
            hom = -vh[:, 3]
            out[b, j] = hom[:3] / hom[3]
    return out


# --------------------------------------------------------------------------------------------
# networks, interpreted straight from the state dict
# --------------------------------------------------------------------------------------------
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.1, BN_EPS)


def pose_resnet(sd, x, num_layers, prefix="backbone.", caffe=False):
    """PoseResNet.forward, mvn/models/pose_resnet.py:293-318 (eval mode).  caffe=True: config.style == 'caffe'
    (pose_resnet.py:322-324, Bottleneck_CAFFE :98-137: bottleneck blocks at every depth, the block stride on the first 1x1).

    Returns (heatmaps, features, alg_confidences|None, vol_confidences|None).
    """
    kind, blocks = RESNET_SPEC[num_layers]
    if caffe:
        kind = "bottleneck"
    g = lambda k: sd[prefix + k]
    x = F.conv2d(x, g("conv1.weight"), None, 2, 3)
    x = F.relu(_bn(sd, prefix + "bn1", x))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nb in enumerate(blocks):
        for bi in range(nb):
            p = "%slayer%d.%d" % (prefix, li + 1, bi)
            st = 2 if (li > 0 and bi == 0) else 1
            res = x
            if kind == "bottleneck":  # pose_resnet.py:75-95 ('simple' style: stride on the 3x3)
                s1, s2 = (st, 1) if caffe else (1, st)
                o = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, s1)))
                o = F.relu(_bn(sd, p + ".bn2", F.conv2d(o, sd[p + ".conv2.weight"], None, s2, 1)))
                o = _bn(sd, p + ".bn3", F.conv2d(o, sd[p + ".conv3.weight"]))
            else:                      # pose_resnet.py:37-54
                o = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, st, 1)))
                o = _bn(sd, p + ".bn2", F.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1))
            if (p + ".downsample.0.weight") in sd:
                res = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, st))
            x = F.relu(o + res)

    def gap_head(hp):  # pose_resnet.py:140-174
        y = F.conv2d(x, sd[hp + ".features.0.weight"], sd[hp + ".features.0.bias"], 1, 1)
        y = F.relu(F.max_pool2d(_bn(sd, hp + ".features.1", y), 2))
        y = F.conv2d(y, sd[hp + ".features.4.weight"], sd[hp + ".features.4.bias"], 1, 1)
        y = F.relu(F.max_pool2d(_bn(sd, hp + ".features.5", y), 2))
        y = y.reshape(y.shape[0], y.shape[1], -1).mean(dim=-1)
        y = F.relu(F.linear(y, sd[hp + ".head.0.weight"], sd[hp + ".head.0.bias"]))
        y = F.relu(F.linear(y, sd[hp + ".head.2.weight"], sd[hp + ".head.2.bias"]))
        return torch.sigmoid(F.linear(y, sd[hp + ".head.4.weight"], sd[hp + ".head.4.bias"]))

    alg_c = gap_head(prefix + "alg_confidences") if (prefix + "alg_confidences.head.0.weight") in sd else None
    vol_c = gap_head(prefix + "vol_confidences") if (prefix + "vol_confidences.head.0.weight") in sd else None
    for i in range(3):
        x = F.conv_transpose2d(x, g("deconv_layers.%d.weight" % (3 * i)), None, 2, 1)
        x = F.relu(_bn(sd, "%sdeconv_layers.%d" % (prefix, 3 * i + 1), x))
    feats = x
    hm = F.conv2d(x, g("final_layer.weight"), g("final_layer.bias"))
    return hm, feats, alg_c, vol_c


def _res3d(sd, p, x):  # v2v.py:20-42
    o = F.relu(_bn(sd, p + ".res_branch.1", F.conv3d(x, sd[p + ".res_branch.0.weight"], sd[p + ".res_branch.0.bias"], 1, 1)))
    o = _bn(sd, p + ".res_branch.4", F.conv3d(o, sd[p + ".res_branch.3.weight"], sd[p + ".res_branch.3.bias"], 1, 1))
    if (p + ".skip_con.0.weight") in sd:
        x = _bn(sd, p + ".skip_con.1", F.conv3d(x, sd[p + ".skip_con.0.weight"], sd[p + ".skip_con.0.bias"]))
    return F.relu(o + x)


def _basic3d(sd, p, x, k):  # v2v.py:7-17
    return F.relu(_bn(sd, p + ".block.1", F.conv3d(x, sd[p + ".block.0.weight"], sd[p + ".block.0.bias"], 1, (k - 1) // 2)))


def _up3d(sd, p, x):  # v2v.py:54-66
    return F.relu(_bn(sd, p + ".block.1", F.conv_transpose3d(x, sd[p + ".block.0.weight"], sd[p + ".block.0.bias"], 2)))


def v2v(sd, x, prefix="volume_net."):
    """V2VModel.forward, mvn/models/v2v.py:164-169 with EncoderDecorder.forward :103-138."""
    x = _basic3d(sd, prefix + "front_layers.0", x, 7)
    for i in (1, 2, 3):
        x = _res3d(sd, prefix + "front_layers.%d" % i, x)
    e = prefix + "encoder_decoder."
    skips = []
    for lvl in range(1, 6):
        skips.append(_res3d(sd, e + "skip_res%d" % lvl, x))
        x = F.max_pool3d(x, 2, 2)
        x = _res3d(sd, e + "encoder_res%d" % lvl, x)
    x = _res3d(sd, e + "mid_res", x)
    for lvl in range(5, 0, -1):
        x = _res3d(sd, e + "decoder_res%d" % lvl, x)
        x = _up3d(sd, e + "decoder_upsample%d" % lvl, x)
        x = x + skips[lvl - 1]
    x = _res3d(sd, prefix + "back_layers.0", x)
    x = _basic3d(sd, prefix + "back_layers.1", x, 1)
    x = _basic3d(sd, prefix + "back_layers.2", x, 1)
    return F.conv3d(x, sd[prefix + "output_layer.weight"], sd[prefix + "output_layer.bias"])


# --------------------------------------------------------------------------------------------
# whole-model forwards
# --------------------------------------------------------------------------------------------
def base_points_from_batch(pred_keypoints_3d, kind):
    """mvn/models/triangulation.py:291-294: pelvis (mpii joint 6) or mid-hip (coco 11,12)."""
    kp = np.asarray(pred_keypoints_3d, dtype=np.float64)
    if kind == "coco":
        return (kp[:, 11, :3] + kp[:, 12, :3]) / 2
    return kp[:, 6, :3].copy()


@torch.no_grad()
def volumetric_forward(sd, config, images, K, R, t, pred_keypoints_3d, thetas=None, stages=False):
    """VolumetricTriangulationNet.forward, mvn/models/triangulation.py:245-355 (eval; theta given).

    images (B,NV,3,H,W); K,R,t per view at image resolution, shared by all samples
    (shape (NV,...)) or per sample ((B,NV,...)).
    Returns dict with keypoints_3d, features, volumes (softmaxed), vol_confidences, coord_volumes,
    base_points and -- with stages=True -- the unprojected volume and the V2V logits.
    """
    m = config.model
    B, NV = images.shape[:2]
    H, W = images.shape[3:]
    hm, feats, _, vol_c = pose_resnet(sd, images.reshape(-1, 3, H, W), m.backbone.num_layers)
    hshape = tuple(hm.shape[2:])
    feats = feats.reshape(B, NV, *feats.shape[1:])
    if vol_c is not None:
        vol_c = vol_c.reshape(B, NV, -1)
        if m.volume_aggregation_method == "conf_norm":
            vol_c = vol_c / vol_c.sum(dim=1, keepdim=True)
    K, R, t = (np.broadcast_to(a, (B,) + a.shape[-3:]) if a.ndim == 3 else a for a in (K, R, t))
    P = torch.from_numpy(resized_projection(K, R, t, (H, W), hshape)).float()
    base = base_points_from_batch(pred_keypoints_3d, m.kind)
    axis = (0, 1, 0) if m.kind == "coco" else (0, 0, 1)
    cmu = bool(m.get("transfer_cmu_to_human36m", False))
    cv = torch.stack([coord_volume(base[b], m.cuboid_side, m.volume_size,
                                   0.0 if thetas is None else float(thetas[b]), axis, cmu)
                      for b in range(B)])
    f = F.conv2d(feats.reshape(-1, *feats.shape[2:]), sd["process_features.0.weight"], sd["process_features.0.bias"])
    f = f.reshape(B, NV, *f.shape[1:])
    unproj = unproject_heatmaps(f, P, cv, m.volume_aggregation_method, vol_c)
    logits = v2v(sd, unproj)
    kp, vols = integrate_tensor_3d_with_coordinates(logits * m.volume_multiplier, cv, softmax=m.volume_softmax)
    out = {"keypoints_3d": kp, "features": f, "volumes": vols, "vol_confidences": vol_c,
           "coord_volumes": cv, "base_points": torch.from_numpy(base).float(), "proj": P}
    if stages:
        out["unprojected"] = unproj
        out["logits"] = logits
    return out


@torch.no_grad()
def algebraic_forward(sd, config, images, K, R, t):
    """AlgebraicTriangulationNet.forward, mvn/models/triangulation.py:149-200 (eval).

    The projection matrices are the IMAGE-resolution ones the caller passes to forward
    (datasets/utils.py:62-63), here rebuilt from K,R,t.
    """
    m = config.model
    B, NV = images.shape[:2]
    H, W = images.shape[3:]
    hm, _, alg_c, _ = pose_resnet(sd, images.reshape(-1, 3, H, W), m.backbone.num_layers)
    J = hm.shape[1]
    if alg_c is None:
        alg_c = torch.ones(B * NV, J)
    kp2d, hm_sm = integrate_tensor_2d(hm * m.heatmap_multiplier, m.heatmap_softmax)
    h, w = hm.shape[2:]
    kp2d = kp2d.reshape(B, NV, J, 2)
    alg_c = alg_c.reshape(B, NV, J)
    alg_c = alg_c / alg_c.sum(dim=1, keepdim=True) + 1e-5
    kp2d = torch.stack([kp2d[..., 0] * (W / w), kp2d[..., 1] * (H / h)], dim=-1)
    K, R, t = (np.broadcast_to(a, (B,) + a.shape[-3:]) if a.ndim == 3 else a for a in (K, R, t))
    P = torch.from_numpy(np.asarray(K, np.float64) @ np.concatenate([R, t], axis=-1)).float()
    kp3d = triangulate_batch_of_points(P, kp2d, alg_c)
    return {"keypoints_3d": kp3d, "keypoints_2d": kp2d, "heatmaps": hm_sm.reshape(B, NV, J, h, w),
            "alg_confidences": alg_c, "proj": P}
