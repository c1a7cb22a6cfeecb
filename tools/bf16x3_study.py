"""Numerical study for a bf16x3 ("split hi/lo") MFMA mode, on the CPU oracle: every convolution of the volumetric
forward is evaluated as conv(hi_x, hi_w) + conv(hi_x, lo_w) + conv(lo_x, hi_w) with x = hi + lo, hi = bf16(x), lo = bf16(x - hi) (fp32
accumulation, like the MFMA), everything else stays fp32; joints are compared with the exact fp64 soft-argmax of the reference's logits
and with the reference's own fp32 joints (tests/golden).  Usage: python tools/bf16x3_study.py [case ...]   (runs on the CPU, ~1-2 min per
C2-shape case)."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import vol_oracle as O
import test_oracle_golden as T

def split(t):
    hi = t.bfloat16().float()
    lo = (t - hi).bfloat16().float()
    return hi, lo

def make(fn, mode):
    def wrapped(x, w, b=None, *a, **k):
        if mode == "fp32":
            return fn(x, w, b, *a, **k)
        xh, xl = split(x); wh, wl = split(w)
        if mode == "bf16":
            return fn(xh, wh, b, *a, **k)
        y = fn(xh, wh, b, *a, **k) + fn(xh, wl, None, *a, **k) + fn(xl, wh, None, *a, **k)
        if mode == "bf16x4":
            y = y + fn(xl, wl, None, *a, **k)
        return y
    return wrapped

orig = {n: getattr(F, n) for n in ("conv2d", "conv3d", "conv_transpose2d", "conv_transpose3d")}
cases = sys.argv[1:] or ["c2_sharp", "c2_default"]
torch.set_num_threads(32)
for tag in cases:
    g = np.load(os.path.join(R, "tests", "golden", "vol_%s.npz" % tag))
    cfg, sd, inp, c = T.build_vol_case(tag)
    kp_ref, kp64 = torch.from_numpy(g["kp"]).double(), torch.from_numpy(g["kp_fp64"]).double()
    for mode in ("fp32", "bf16x3", "bf16x4", "bf16"):
        for n, fn in orig.items():
            setattr(F, n, make(fn, mode))
        o = O.volumetric_forward(sd, cfg, inp["images"], inp["K"], inp["R"], inp["t"], inp["pred_keypoints_3d"], thetas=g["thetas"] if c["rotate"] else None)
        kp = o["keypoints_3d"].double()
        e64 = float(((kp - kp64).abs() / kp64.abs().clamp(min=1.0)).max())
        er = float(((kp - kp_ref).abs() / kp_ref.abs().clamp(min=1.0)).max())
        print("%-12s %-7s joints max rel (1 mm floor): vs exact soft-argmax of the reference logits %.3e, vs the reference %.3e (reference's own %.1e)" % (
            tag, mode, e64, er, float(g["ref_self_rel"])), flush=True)
    for n, fn in orig.items():
        setattr(F, n, fn)
