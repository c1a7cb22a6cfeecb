"""Volume geometry helpers with the reference's names (mvn/utils/volumetric.py of the reference).

``Cuboid3D`` is only the (position, sides) record that ``VolumetricTriangulationNet.forward`` returns
(reference :44-47); the cv2 drawing code is visualisation and out of scope."""
import math

import numpy as np
import torch


class Cuboid3D:
    def __init__(self, position, sides):
        self.position = position
        self.sides = sides


def get_rotation_matrix(axis, theta):
    """Counter-clockwise rotation by theta about axis, Euler-Rodrigues form (reference :87-99), fp64."""
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / math.sqrt(float(ax @ ax))
    a = math.cos(theta / 2.0)
    b, c, d = (-ax * math.sin(theta / 2.0)).tolist()
    return np.array([
        [a * a + b * b - c * c - d * d, 2.0 * (b * c + a * d), 2.0 * (b * d - a * c)],
        [2.0 * (b * c - a * d), a * a + c * c - b * b - d * d, 2.0 * (c * d + a * b)],
        [2.0 * (b * d + a * c), 2.0 * (c * d - a * b), a * a + d * d - b * b - c * c]], dtype=np.float64)


def rotate_coord_volume(coord_volume, theta, axis):
    """Rotate a (..., 3) fp32 coordinate grid about the origin (reference :102-114) with lt_rotate_points.
    Inside VolumetricTriangulationNet the rotation is fused into the grid construction (lt_coord_volumes)."""
    import lt_hip as H
    H.require_gpu(coord_volume, "coord_volume")
    x = coord_volume.float().contiguous()
    rot = torch.from_numpy(get_rotation_matrix(axis, theta)).float().to(x.device).contiguous()
    y = torch.empty_like(x)
    H.check(H.lib().lt_rotate_points(x.data_ptr(), rot.data_ptr(), y.data_ptr(), x.numel() // 3, H.cur_stream()), "lt_rotate_points")
    return y
