"""Numerical study for an fp16 hi/lo split MFMA mode (VERDICT r2 "next" item 2), on the CPU oracle: every convolution of the volumetric
forward is evaluated as conv(hi_x, hi_w) + conv(hi_x, lo_w) + conv(lo_x, hi_w) with x = hi + lo, hi = fp16(s x), lo = fp16(s x - hi)
(2 x 11 significant bits; fp32 accumulation like the MFMA, three products at the 16-bit MFMA rate), everything else fp32.  s is a
power of two (exact): "raw" = 1, "pt" = per tensor so that max |s x| lands in [2^13, 2^14) (lo stays a NORMAL fp16 for every element
above 2^-11 of the tensor's maximum), "st" = static: activations x 2^4, weights per tensor (what a plan can bake without looking at the
data).  Joints are compared with the exact fp64 soft-argmax of the reference's logits and with the reference's own fp32 joints
(tests/golden).  Also prints the largest |activation| any convolution saw (fp16 overflows at 65504).
Usage: python tools/fp16x3_study.py [case ...] [--modes a,b]   (CPU, ~1-2 min per C2-shape case and mode)."""
import math, os, sys
import numpy as np, torch, torch.nn.functional as F
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import vol_oracle as O
import test_oracle_golden as T

STAT = {"amax": 0.0, "amin_scale": 99, "amax_scale": -99}

def pow2_scale(t, top=14):
    m = float(t.abs().max())
    if m == 0.0 or not math.isfinite(m):
        return 1.0
    return 2.0 ** (top - 1 - math.floor(math.log2(m)))      # max |s t| in [2^(top-1), 2^top)

def split16(t, s):
    ts = t * s
    hi = ts.half().float()
    lo = (ts - hi).half().float()
    return hi, lo

def make(fn, mode):
    def wrapped(x, w, b=None, *a, **k):
        if mode == "fp32":
            return fn(x, w, b, *a, **k)
        STAT["amax"] = max(STAT["amax"], float(x.abs().max()))
        if mode == "fp16":
            return fn(x.half().float(), w.half().float(), b, *a, **k)
        sx = {"raw": 1.0, "pt": None, "st": 16.0}[mode.split("_")[1]]
        sw = 1.0 if mode.endswith("raw") else pow2_scale(w)
        if sx is None:
            sx = pow2_scale(x)
        xh, xl = split16(x, sx); wh, wl = split16(w, sw)
        assert torch.isfinite(xh).all() and torch.isfinite(wh).all(), "fp16 overflow"
        if mode.startswith("fp16x2w"):        # diagnostic: weights exact (fp32), activations split -> isolates the activations' representation error
            y = fn(xh, w * sw, None, *a, **k) + fn(xl, w * sw, None, *a, **k)
        elif mode.startswith("fp16x2x"):      # diagnostic: activations exact, weights split
            y = fn(x * sx, wh, None, *a, **k) + fn(x * sx, wl, None, *a, **k)
        else:
            y = fn(xh, wh, None, *a, **k) + (fn(xh, wl, None, *a, **k) + fn(xl, wh, None, *a, **k))
            if mode.startswith("fp16x4"):     # + lo * lo: the dropped term
                y = y + fn(xl, wl, None, *a, **k)
        y = y * (1.0 / (sx * sw))
        if b is not None:
            y = y + b.reshape(1, -1, *([1] * (y.dim() - 2)))
        return y
    return wrapped

orig = {n: getattr(F, n) for n in ("conv2d", "conv3d", "conv_transpose2d", "conv_transpose3d")}
args = [a for a in sys.argv[1:] if not a.startswith("--")]
modes = ["fp32", "fp16x3_raw", "fp16x3_pt", "fp16x3_st", "fp16"]
for a in sys.argv[1:]:
    if a.startswith("--modes"):
        modes = a.split("=", 1)[1].split(",")
cases = args or ["c2_sharp", "c2_default"]
torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "8")))      # the goldens were generated with 8 threads (tests/conftest.py)
for tag in cases:
    g = np.load(os.path.join(R, "tests", "golden", "vol_%s.npz" % tag))
    cfg, sd, inp, c = T.build_vol_case(tag)
    kp_ref, kp64 = torch.from_numpy(g["kp"]).double(), torch.from_numpy(g["kp_fp64"]).double()
    lg_ref = torch.from_numpy(g["logits"]).double() if "logits" in g.files else None
    for mode in modes:
        STAT["amax"] = 0.0
        for n, fn in orig.items():
            setattr(F, n, make(fn, mode))
        try:
            o = O.volumetric_forward(sd, cfg, inp["images"], inp["K"], inp["R"], inp["t"], inp["pred_keypoints_3d"], thetas=g["thetas"] if c["rotate"] else None, stages=True)
        finally:
            for n, fn in orig.items():
                setattr(F, n, fn)
        kp = o["keypoints_3d"].double()
        e64 = float(((kp - kp64).abs() / kp64.abs().clamp(min=1.0)).max())
        er = float(((kp - kp_ref).abs() / kp_ref.abs().clamp(min=1.0)).max())
        el = ""
        if lg_ref is not None and "logits" in o:
            lg = o["logits"].double()
            el = ", logits max abs / max|ref| %.2e" % float((lg - lg_ref.reshape(lg.shape)).abs().max() / lg_ref.abs().max())
        print("%-12s %-11s joints max rel (1 mm floor): vs exact soft-argmax of the reference logits %.3e, vs the reference %.3e (reference's own %.1e)%s; max |conv input| %.3g" % (
            tag, mode, e64, er, float(g["ref_self_rel"]), el, STAT["amax"]), flush=True)
