"""Helpers shared by the -m gpu tests: run liblt_hip ops on cuda:0, compare with torch-CPU fp32 / the oracle,
and collect the achieved error of every check into gpurun_out/parity_report.json."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


def rel_err(a, ref):
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).detach().double().cpu()
    ref = torch.as_tensor(np.asarray(ref) if not torch.is_tensor(ref) else ref).detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


RMS_FRAC = 0.3          # loose (bf16) checks: rms(d) / rms(ref) <= RMS_FRAC * tol as well


def rms_err(a, ref):
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).detach().double().cpu()
    ref = torch.as_tensor(np.asarray(ref) if not torch.is_tensor(ref) else ref).detach().double().cpu()
    return float(torch.sqrt(((a - ref) ** 2).mean()) / max(float(torch.sqrt((ref ** 2).mean())), 1e-30))


def check(name, a, ref, tol):
    """max|d| <= tol * max|ref|.  A bf16-sized tolerance (>= 5e-3) hides structured mistakes -- one wrong tap of a 27-tap kernel with
    small weights stays under 1.5 % of max|ref| -- so those checks ALSO bound the rms error against rms(ref): output rounding to bf16 is
    ~1.1e-3 of each element (2^-9 / sqrt 3), a wrong / missing tap of k taps is ~1 / sqrt(k) (19 % at 27 taps)."""
    e = rel_err(a, ref)
    REPORT[name] = {"err": e, "tol": tol}
    assert e <= tol, "%s: max|d|/max|ref| = %.3e > tol %.1e" % (name, e, tol)
    if tol >= 5e-3:
        r = rms_err(a, ref)
        REPORT[name]["rms_err"] = r
        REPORT[name]["rms_tol"] = RMS_FRAC * tol
        assert r <= RMS_FRAC * tol, "%s: rms(d)/rms(ref) = %.3e > %.1e" % (name, r, RMS_FRAC * tol)
    return e


def record(name, value):
    REPORT[name] = value


def flush_report():
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_report.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(REPORT)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)


def to_cl(x, c_pad=None, dtype=torch.float32):
    """N,C,(D),H,W cpu -> Act-shaped N,D,H,W,C tensor on the GPU."""
    if x.dim() == 4:
        x = x.unsqueeze(2)
    x = x.permute(0, 2, 3, 4, 1).contiguous()
    if c_pad is not None and c_pad > x.shape[-1]:
        x = torch.cat([x, torch.zeros(*x.shape[:-1], c_pad - x.shape[-1])], dim=-1)
    return x.to("cuda:0", dtype).contiguous()


def from_cl(y, nd):
    y = y.float().cpu().permute(0, 4, 1, 2, 3)
    return y[:, :, 0] if nd == 2 else y


def bf16_round(t):
    return t.to(torch.bfloat16).float()
