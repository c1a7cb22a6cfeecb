"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Deterministic synthetic weights and inputs for the volumetric / algebraic triangulation path
(SURVEY.md section 8d).  Everything is derived from integer seeds with torch's CPU generator, so the
GPU box (same image, no /root/reference, no dataset) regenerates bit-identical tensors; a few
checksums are stored with the golden fixtures to detect generator drift.

Weights are NOT the reference ctor defaults: BatchNorm affine parameters and running statistics,
and every conv bias, are randomised so that BN folding and bias handling are actually exercised
(the ctor defaults gamma=1, beta=0, mean=0, var=1, bias=0 would hide a broken fold).  The same
state_dict is loaded into the real reference (strict=True) when the golden outputs are generated.
"""
import math

import numpy as np
import torch

# "default" weights: V2V output layer scaled so that the logit std is ~0.02 at BASELINE config 2
# (what the reference's own xavier init gives: near-uniform 3D softmax, joints ~ cube centroid).
# "sharpened" weights: output layer x250 on top -> logit std ~5, max prob ~1e-2..1e-1: the 3D
# soft-argmax becomes input-sensitive but stays well conditioned (hard parts).
# Calibrated with oracle/make_golden.py, which prints the measured logit std.
OUT_DEFAULT_GAIN = 1.0 / 450.0
SHARPEN_GAIN = 250.0


def make_state_dict(spec, seed=0, sharpen=False, basic_block=False):
    """spec: OrderedDict name -> (shape, role) from oracle/spec.py.  Returns name -> fp32 tensor."""
    g = torch.Generator().manual_seed(1000003 * seed + 17)
    sd = {}
    for name, (shape, role) in spec.items():
        if role == "conv_w":
            fan_in = int(np.prod(shape[1:]))
            w = torch.randn(shape, generator=g) * math.sqrt(1.5 / fan_in)
        elif role == "deconv_w":
            # transposed conv: each output tap sees Cin * prod(k)/prod(stride=2) inputs
            fan_in = shape[0] * int(np.prod(shape[2:])) / (2 ** (len(shape) - 2))
            w = torch.randn(shape, generator=g) * math.sqrt(1.5 / fan_in)
        elif role in ("conv_b", "lin_b"):
            w = torch.randn(shape, generator=g) * 0.05
        elif role == "lin_w":
            w = torch.randn(shape, generator=g) * math.sqrt(2.0 / shape[1])
        elif role == "bn_gamma":
            last = name.endswith("bn3.weight") or name.endswith("res_branch.4.weight") or \
                (basic_block and "layer" in name and name.endswith("bn2.weight"))
            lo, hi = (0.1, 0.2) if last else (0.8, 1.2)
            w = lo + (hi - lo) * torch.rand(shape, generator=g)
        elif role == "bn_beta":
            w = torch.randn(shape, generator=g) * 0.1
        elif role == "bn_mean":
            w = torch.randn(shape, generator=g) * 0.1
        elif role == "bn_var":
            w = 0.5 + torch.rand(shape, generator=g)
        elif role == "bn_count":
            w = torch.zeros((), dtype=torch.int64)
        else:
            raise KeyError(role)
        sd[name] = w
    for k in list(sd):
        if k.endswith("output_layer.weight") or k.endswith("output_layer.bias"):
            # sharpen: False, True (= SHARPEN_GAIN) or an explicit gain (fixtures whose shape changes the un-scaled logit std)
            sd[k] = sd[k] * (OUT_DEFAULT_GAIN * (1.0 if not sharpen else SHARPEN_GAIN if sharpen is True else float(sharpen)))
    return sd


def state_dict_checksum(sd):
    """Order-independent fp64 digest of a state dict (sum and sum of squares of every tensor)."""
    s = 0.0
    q = 0.0
    for k in sorted(sd):
        t = sd[k].double()
        s += float(t.sum())
        q += float((t * t).sum())
    return [s, q]


def ring_cameras(n_views, image_size, radius=4000.0, height=1000.0, inside=False):
    """NV pinhole cameras on a ring looking at the origin, z up.

    Returns K (NV,3,3), R (NV,3,3), t (NV,3,1) float64 at IMAGE resolution.  ``inside=True`` moves
    camera 0 into the voxel cube so that some voxels have depth <= 0 (exercises the z<=0 mask of
    /root/reference/mvn/utils/op.py:123,141).
    """
    Ks, Rs, ts = [], [], []
    for v in range(n_views):
        phi = 2.0 * math.pi * v / n_views
        C = np.array([radius * math.cos(phi), radius * math.sin(phi), height])
        if inside and v == 0:
            C = np.array([300.0, 100.0, 200.0])
        fwd = -C / np.linalg.norm(C)               # camera z axis: towards the origin
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], axis=0)   # world -> camera
        t = (-R @ C).reshape(3, 1)
        f = 1.2 * image_size
        K = np.array([[f, 0.0, image_size / 2.0], [0.0, f, image_size / 2.0], [0.0, 0.0, 1.0]])
        Ks.append(K); Rs.append(R); ts.append(t)
    return np.stack(Ks), np.stack(Rs), np.stack(ts)


def make_inputs(batch_size, n_views, image_size, seed=0, n_joints=17, inside=False):
    """Synthetic batch: images (B,NV,3,H,H) fp32 ~N(0,1); cameras; pelvis predictions (B,J,3) mm."""
    g = torch.Generator().manual_seed(7919 * seed + 3)
    images = torch.randn(batch_size, n_views, 3, image_size, image_size, generator=g)
    K, R, t = ring_cameras(n_views, image_size, inside=inside)
    rs = np.random.RandomState(seed + 11)
    pred_kp = rs.randn(batch_size, n_joints, 3) * 100.0
    return {"images": images, "K": K, "R": R, "t": t, "pred_keypoints_3d": pred_kp}


class AttrDict(dict):
    """Minimal recursive attribute dict (stand-in for easydict in tests and the oracle)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def vol_config(num_layers=152, volume_size=64, aggregation="softmax", multiplier=1.0, kind="mpii",
               cuboid_side=2500.0, num_joints=17, volume_softmax=True):
    """Same keys as experiments/human36m/train/human36m_vol_softmax.yaml:26-53 of the reference."""
    return AttrDict({
        "model": {
            "name": "vol", "kind": kind, "volume_aggregation_method": aggregation,
            "init_weights": False, "checkpoint": "", "use_gt_pelvis": False,
            "cuboid_side": cuboid_side, "volume_size": volume_size, "volume_multiplier": multiplier,
            "volume_softmax": volume_softmax, "heatmap_softmax": True, "heatmap_multiplier": 100.0,
            "backbone": {"name": "resnet%d" % num_layers, "style": "simple", "init_weights": False,
                         "checkpoint": "", "num_joints": num_joints, "num_layers": num_layers},
        }
    })


def alg_config(num_layers=50, use_confidences=True, num_joints=17):
    """Same keys as experiments/human36m/train/human36m_alg.yaml (model section)."""
    return AttrDict({
        "model": {
            "name": "alg", "init_weights": False, "checkpoint": "",
            "use_confidences": use_confidences, "heatmap_multiplier": 100.0, "heatmap_softmax": True,
            "backbone": {"name": "resnet%d" % num_layers, "style": "simple", "init_weights": False,
                         "checkpoint": "", "num_joints": num_joints, "num_layers": num_layers},
        }
    })
