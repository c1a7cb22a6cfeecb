"""Functional ops of the volumetric path with the reference's names and signatures
(mvn/utils/op.py of the reference), executed by liblt_hip.so.

Public tensors keep the reference's layouts (heatmaps B,NV,C,h,w; volumes B,C,V,V,V); internally the
kernels are channels-last, so a channels-last-strided input (what our own networks produce) is consumed
without a copy, and outputs are returned as permuted views of channels-last storage."""
import numpy as np
import torch

import lt_hip as H

_METHODS = ("sum", "max", "softmax", "conf", "conf_norm")


def _as_channels_last(x, n_lead):
    """x: (*lead, C, *spatial) -> contiguous (*lead, *spatial, C) tensor (zero-copy when x already is a
    permuted view of such storage)."""
    nd = x.dim()
    perm = list(range(n_lead)) + list(range(n_lead + 1, nd)) + [n_lead]
    return x.permute(*perm).contiguous()


def build_coord_volumes(base_points, cuboid_side, volume_size, thetas=None, axis=(0, 0, 1), cmu_transfer=False, device="cuda:0"):
    """Per-sample voxel-centre grids (B,V,V,V,3) fp32 -- the loop body of
    VolumetricTriangulationNet.forward (reference mvn/models/triangulation.py:298-339) as one launch.

    base_points: (B,3) array-like (mm), fp64 on the host like the reference; thetas: per-sample rotation
    about ``axis`` through the base point (None = 0).
    """
    from mvn.utils import volumetric
    base = np.asarray(base_points, dtype=np.float64).reshape(-1, 3)
    B = base.shape[0]
    pos = (base - cuboid_side / 2.0).astype(np.float32)
    rot = np.stack([volumetric.get_rotation_matrix(axis, 0.0 if thetas is None else float(thetas[b])) for b in range(B)]).astype(np.float32)
    host = torch.from_numpy(np.concatenate([pos.reshape(-1), base.astype(np.float32).reshape(-1), rot.reshape(-1)]))
    dev = host.to(device)
    out = torch.empty(B, volume_size, volume_size, volume_size, 3, dtype=torch.float32, device=device)
    step = float(np.float32(cuboid_side / (volume_size - 1)))
    H.check(H.lib().lt_coord_volumes(dev.data_ptr(), dev.data_ptr() + 12 * B, dev.data_ptr() + 24 * B, step, B, volume_size,
                                     int(bool(cmu_transfer)), out.data_ptr(), H.cur_stream()), "lt_coord_volumes")
    return out


def unproject_heatmaps(heatmaps, proj_matricies, coord_volumes, volume_aggregation_method="sum", vol_confidences=None):
    """heatmaps (B,NV,C,h,w), proj_matricies (B,NV,3,4), coord_volumes (B,V0,V1,V2,3),
    vol_confidences (B,NV,C) for 'conf*' -> volumes (B,C,V0,V1,V2).  Reference: op.py:99-166."""
    if volume_aggregation_method not in _METHODS:
        raise ValueError("Unknown volume_aggregation_method: {}".format(volume_aggregation_method))
    H.require_gpu(heatmaps, "heatmaps")
    code = H.dtype_code(heatmaps.dtype)
    B, NV, Cc, h, w = heatmaps.shape
    v0, v1, v2 = coord_volumes.shape[1:4]
    feats = _as_channels_last(heatmaps, 2)
    P = proj_matricies.to(heatmaps.device, torch.float32).contiguous()
    cv = coord_volumes.to(heatmaps.device, torch.float32).contiguous()
    conf = None
    if volume_aggregation_method.startswith("conf"):
        conf = vol_confidences.to(heatmaps.device, torch.float32).contiguous()
    out = torch.empty(B, v0, v1, v2, Cc, dtype=heatmaps.dtype, device=heatmaps.device)
    H.check(H.lib().lt_unproject_fwd(code, feats.data_ptr(), P.data_ptr(), cv.data_ptr(), H.ptr(conf), out.data_ptr(), B, NV, Cc, h, w,
                                     v0, v1, v2, H.AGG['conf'] if conf is not None else H.AGG[volume_aggregation_method], H.cur_stream()), "lt_unproject_fwd")
    return out.permute(0, 4, 1, 2, 3)


def integrate_tensor_3d_with_coordinates(volumes, coord_volumes, softmax=True):
    """volumes (B,J,V0,V1,V2) logits, coord_volumes (B,V0,V1,V2,3) -> (coordinates (B,J,3), volumes after
    softmax / ReLU).  Reference: op.py:84-96."""
    H.require_gpu(volumes, "volumes")
    B, J = volumes.shape[:2]
    nvox = int(np.prod(volumes.shape[2:]))
    cl = volumes.permute(0, 2, 3, 4, 1).is_contiguous() and not volumes.is_contiguous()
    lg = (volumes.permute(0, 2, 3, 4, 1) if cl else volumes).float().contiguous()
    cv = coord_volumes.to(volumes.device, torch.float32).contiguous()
    kp = torch.empty(B, J, 3, dtype=torch.float32, device=volumes.device)
    probs = torch.empty((B, J) + tuple(volumes.shape[2:]), dtype=torch.float32, device=volumes.device)
    lib = H.lib()
    ws = torch.empty(max(1, lib.lt_softargmax3d_workspace(B, J, nvox)), dtype=torch.uint8, device=volumes.device)
    H.check(lib.lt_softargmax3d_fwd(lg.data_ptr(), cv.data_ptr(), 1.0, int(bool(softmax)), int(cl), J, kp.data_ptr(), probs.data_ptr(),
                                    B, J, nvox, ws.data_ptr(), H.cur_stream()), "lt_softargmax3d_fwd")
    return kp, probs


def integrate_tensor_2d(heatmaps, softmax=True):
    """heatmaps (N,J,h,w) -> (coordinates (N,J,2) as (x,y), heatmaps after softmax / ReLU).  Reference: op.py:11-47."""
    H.require_gpu(heatmaps, "heatmaps")
    N, J, h, w = heatmaps.shape
    hm = heatmaps.float().contiguous()
    coords = torch.empty(N, J, 2, dtype=torch.float32, device=hm.device)
    probs = torch.empty_like(hm)
    H.check(H.lib().lt_softargmax2d_fwd(hm.data_ptr(), 1.0, int(bool(softmax)), coords.data_ptr(), probs.data_ptr(), N * J, h, w, H.cur_stream()),
            "lt_softargmax2d_fwd")
    return coords, probs
