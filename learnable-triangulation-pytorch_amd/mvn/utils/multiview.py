"""Camera model and multi-view helpers with the reference's names (mvn/utils/multiview.py).

Host geometry stays numpy fp64 exactly like the reference; the batched DLT runs in liblt_hip
(lt_triangulate_dlt)."""
import numpy as np
import torch

import lt_hip as H


class Camera:
    """Pinhole camera R (3x3), t (3x1), K (3x3) -- reference :5-52."""

    def __init__(self, R, t, K, dist=None, name=""):
        self.R = np.array(R, dtype=np.float64).copy()
        assert self.R.shape == (3, 3)
        self.t = np.array(t, dtype=np.float64).copy()
        assert self.t.size == 3
        self.t = self.t.reshape(3, 1)
        self.K = np.array(K, dtype=np.float64).copy()
        assert self.K.shape == (3, 3)
        self.dist = None if dist is None else np.array(dist).copy().flatten()
        self.name = name

    def update_after_crop(self, bbox):
        left, upper, _, _ = bbox
        self.K[0, 2] -= left
        self.K[1, 2] -= upper

    def update_after_resize(self, image_shape, new_image_shape):
        (h, w), (nh, nw) = image_shape, new_image_shape
        self.K[0, 0] *= nw / w
        self.K[1, 1] *= nh / h
        self.K[0, 2] *= nw / w
        self.K[1, 2] *= nh / h

    @property
    def extrinsics(self):
        return np.hstack([self.R, self.t])

    @property
    def projection(self):
        return self.K.dot(self.extrinsics)


def stack_cameras(cameras):
    """batch['cameras'] (list[NV] of list[B] of Camera-like objects with .R .t .K, datasets/utils.py:26)
    -> K (B,NV,3,3), R (B,NV,3,3), t (B,NV,3,1) fp64."""
    nv, bs = len(cameras), len(cameras[0])
    K = np.empty((bs, nv, 3, 3)); R = np.empty((bs, nv, 3, 3)); t = np.empty((bs, nv, 3, 1))
    for v in range(nv):
        for b in range(bs):
            c = cameras[v][b]
            K[b, v] = c.K; R[b, v] = c.R; t[b, v] = np.asarray(c.t).reshape(3, 1)
    return K, R, t


def resized_projections(K, R, t, image_shape, new_image_shape):
    """Vectorised ``update_after_resize`` + ``projection`` for stacked cameras (fp64)."""
    (h, w), (nh, nw) = image_shape, new_image_shape
    K = K.copy()
    K[..., 0, 0] *= nw / w
    K[..., 1, 1] *= nh / h
    K[..., 0, 2] *= nw / w
    K[..., 1, 2] *= nh / h
    return K @ np.concatenate([R, t], axis=-1)


def euclidean_to_homogeneous(points):
    if isinstance(points, np.ndarray):
        return np.hstack([points, np.ones((len(points), 1))])
    if torch.is_tensor(points):
        return torch.cat([points, points.new_ones((points.shape[0], 1))], dim=1)
    raise TypeError("Works only with numpy arrays and PyTorch tensors.")


def homogeneous_to_euclidean(points):
    if isinstance(points, np.ndarray):
        return (points.T[:-1] / points.T[-1]).T
    if torch.is_tensor(points):
        return points[:, :-1] / points[:, -1:]
    raise TypeError("Works only with numpy arrays and PyTorch tensors.")


def project_3d_points_to_image_plane_without_distortion(proj_matrix, points_3d, convert_back_to_euclidean=True):
    """Host helper (reference :89-110).  Inside the unprojection this projection is fused into
    lt_unproject_fwd; this stand-alone form is for small point sets (numpy, or torch on any device)."""
    both_np = isinstance(proj_matrix, np.ndarray) and isinstance(points_3d, np.ndarray)
    both_t = torch.is_tensor(proj_matrix) and torch.is_tensor(points_3d)
    if not (both_np or both_t):
        raise TypeError("Works only with numpy arrays and PyTorch tensors.")
    res = euclidean_to_homogeneous(points_3d) @ (proj_matrix.T if both_np else proj_matrix.t())
    return homogeneous_to_euclidean(res) if convert_back_to_euclidean else res


class _TriangulateFn(torch.autograd.Function):
    """lt_triangulate_dlt / lt_triangulate_dlt_bwd as one node: what autograd derives in the reference through torch.svd (:163) for the 2D
    points and the confidences (the projection matrices get no gradient, as there)."""

    @staticmethod
    def forward(ctx, P, pts, conf):
        B, NV, J = pts.shape[:3]
        out = torch.empty(B, J, 3, dtype=torch.float32, device=pts.device)
        H.check(H.lib().lt_triangulate_dlt(P.data_ptr(), pts.data_ptr(), H.ptr(conf), out.data_ptr(), B, NV, J, H.cur_stream()), "lt_triangulate_dlt")
        ctx.save_for_backward(P, pts, conf if conf is not None else torch.empty(0, device=pts.device))
        ctx.has_conf = conf is not None
        return out

    @staticmethod
    def backward(ctx, g):
        P, pts, conf = ctx.saved_tensors
        conf = conf if ctx.has_conf else None
        B, NV, J = pts.shape[:3]
        g = g.float().contiguous()
        gpts = torch.empty_like(pts)
        gconf = torch.empty_like(conf) if conf is not None else None
        H.check(H.lib().lt_triangulate_dlt_bwd(P.data_ptr(), pts.data_ptr(), H.ptr(conf), g.data_ptr(), gpts.data_ptr(), H.ptr(gconf), B, NV, J,
                                               H.cur_stream()), "lt_triangulate_dlt_bwd")
        return None, gpts, gconf


def triangulate_batch_of_points(proj_matricies_batch, points_batch, confidences_batch=None):
    """Confidence-weighted DLT for every (sample, joint) in one launch (reference :171-183 loops B x J
    torch.svd calls).  proj (B,NV,3,4), points (B,NV,J,2), confidences (B,NV,J) -> (B,J,3) fp32.  Differentiable with respect to the
    points and the confidences when they require grad."""
    H.require_gpu(points_batch, "points_batch")
    P = proj_matricies_batch.to(points_batch.device, torch.float32).contiguous()
    pts = points_batch.float().contiguous()
    conf = None if confidences_batch is None else confidences_batch.float().contiguous()
    if torch.is_grad_enabled() and (pts.requires_grad or (conf is not None and conf.requires_grad)):
        return _TriangulateFn.apply(P, pts, conf)
    B, NV, J = pts.shape[:3]
    out = torch.empty(B, J, 3, dtype=torch.float32, device=pts.device)
    H.check(H.lib().lt_triangulate_dlt(P.data_ptr(), pts.data_ptr(), H.ptr(conf), out.data_ptr(), B, NV, J, H.cur_stream()),
            "lt_triangulate_dlt")
    return out


def triangulate_point_from_multiple_views_linear_torch(proj_matricies, points, confidences=None):
    """Single-point form of the above (reference :141-168)."""
    conf = None if confidences is None else confidences[None, :, None]
    return triangulate_batch_of_points(proj_matricies[None], points[None, :, None, :], conf)[0, 0]
