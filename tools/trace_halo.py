#!/usr/bin/env python
"""Shader-clock accounting of the persistent 3^3 halo kernel's per-tile phases (GPU, profiling build liblt_hip_trace.so)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "learnable-triangulation-pytorch_amd")
# LT_TRACE_DEFS="LT_ABL_NO_STORE": extra defines for an ablation of the traced kernel (results wrong by design, timing only)
EXTRA = os.environ.get("LT_TRACE_DEFS", "").split()
VARIANT = "trace" + "".join("_" + d.lower() for d in EXTRA)
os.environ["LT_HIP_LIB"] = os.path.join(PKG, "lib", "liblt_hip_%s.so" % VARIANT)
sys.path.insert(0, PKG)
if not os.path.exists(os.environ["LT_HIP_LIB"]):   # profiling build of the same sources (hipcc is on the GPU box too)
    import lt_build
    lt_build.build_variant(VARIANT, ["LT_TRACE"] + EXTRA)
import numpy as np
import torch

import lt_engine as E
import lt_hip as H


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    with_res = "--no-residual" not in sys.argv
    lib = H.lib()
    lib.lt_trace_read_halo.restype = C.c_int
    lib.lt_trace_read_halo.argtypes = [C.c_void_p, C.c_int]
    dev, dt = "cuda:0", torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(B, 64, 64, 64, 32, device=dev).to(dt)
    w = torch.randn(32, 32, 3, 3, 3) * 0.05
    res = E.Act(torch.randn(B, 64, 64, 64, 32, device=dev).to(dt))
    b = E.PlanBuilder(dev, dt)
    b.conv(E.Act(x), w, None, None, stride=1, pad=1, relu=True, residual=res if with_res else None)
    plan = b.finish()
    for _ in range(3):
        plan.run_eager(st)
    e0, e1 = H.Event(), H.Event()
    e0.record(st)
    plan.run_eager(st)
    e1.record(st)
    us = e0.elapsed_ms(e1) * 1e3
    buf = np.zeros(8 * 64, dtype=np.int64)
    lib.lt_trace_read_halo(buf.ctypes.data, buf.size)
    r = buf.reshape(-1, 8)
    r = r[r[:, 7] > 0]
    nt = r[:, 7].mean()
    names = ["total", "wait+barrier", "halo issue / tile setup", "res issue", "tap loop (+ deferred epilogue)", "barrier", "epilogue / bookkeeping"]
    kern = "persistent" if os.environ.get("LT_HALO_NO_COL") else "column-walk"
    print(kern + (" [" + " ".join(EXTRA) + "]" if EXTRA else "") + " 3^3 32->32 @64^3 B=%d: %.0f us, %d workgroups sampled, %.1f tiles each" % (B, us, len(r), nt))
    for k, nm in enumerate(names):
        print("  %-32s %9.0f cycles/tile" % (nm, r[:, k].mean() / nt))


if __name__ == "__main__":
    main()
