#!/usr/bin/env python
"""Shader-clock accounting of the conv_igemm2 K loop (GPU, profiling build): where do the cycles of a K step go?
Needs lib/liblt_hip_trace.so (python learnable-triangulation-pytorch_amd/lt_build.py --variant trace LT_TRACE).
Prints, per layer, the mean over the sampled workgroups (wave 0 of every 32nd workgroup) of: cycles in the K loop, split
into wait-for-DMA (s_waitcnt vmcnt), barrier, DMA issue and fragment-read+MFMA, and the shader clock they were counted at."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "learnable-triangulation-pytorch_amd")
os.environ["LT_HIP_LIB"] = os.path.join(PKG, "lib", "liblt_hip_trace.so")
sys.path.insert(0, PKG)
if not os.path.exists(os.environ["LT_HIP_LIB"]):   # profiling build of the same sources (hipcc is on the GPU box too)
    import lt_build
    lt_build.build_variant("trace", ["LT_TRACE"])
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import lt_engine as E
import lt_hip as H
from conv_bench import shapes, TILES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--only", default="rn l3,rn l2 3x3,rn l1 3x3")
    ap.add_argument("--tiles", default="auto", help="comma list: auto,128x128,128x64,64x64,...")
    ap.add_argument("--stages", default="0", help="comma list of ring depths (0 = auto)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="fp32: the exact-fp32 (parity) kernels")
    args = ap.parse_args()
    lib = H.lib()
    lib.lt_trace_read.restype = C.c_int
    lib.lt_trace_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.lt_trace_read3.restype = C.c_int
    lib.lt_trace_read3.argtypes = [C.c_void_p, C.c_int, C.c_int]
    dev, dt = "cuda:0", {"bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    st = torch.cuda.current_stream().cuda_stream
    buf = np.zeros(8 * 1024, dtype=np.int64)
    print("%-26s %-10s %3s | %6s %7s | %6s %6s %6s %6s | %5s %5s" % ("layer", "tile", "nst", "blocks", "cyc/ks", "vmwait", "barr", "issue", "mfma", "MHz", "us"))
    for (name, nd, N, cin, cout, k, s, p, sp, tr) in shapes(args.batch):
        if not any(o in name for o in args.only.split(",")):
            continue
        x = torch.randn(N, *((1,) if nd == 2 else ()), *sp, cin, device=dev).to(dt)
        w = torch.randn(*((cin, cout) if tr else (cout, cin)), *([k] * nd)) * 0.05
        osp = [(d + 2 * p - k) // s + 1 for d in sp]
        res = None if tr else E.Act(torch.randn(N, *((1,) if nd == 2 else ()), *osp, cout, device=dev).to(dt))
        for tname in args.tiles.split(","):
            for nst in [int(v) for v in args.stages.split(",")]:
                tile = 0 if tname == "auto" else (30 if tname == "v3" else TILES[tname])
                rd = lib.lt_trace_read3 if tname == "v3" else lib.lt_trace_read
                if tname == "v3" and nst != int(args.stages.split(",")[0]):
                    continue
                b = E.PlanBuilder(dev, dt, tile_override=tile, stages=nst)
                try:
                    b.conv(E.Act(x), w, None, None, stride=s, pad=p, transposed=tr, relu=True, residual=res)
                    plan = b.finish()
                    for _ in range(3):
                        plan.run_eager(st)
                    lib.lt_trace_read(buf.ctypes.data, buf.size, 1)
                    lib.lt_trace_read3(buf.ctypes.data, buf.size, 1)
                    e0, e1 = H.Event(), H.Event()
                    e0.record(st)
                    plan.run_eager(st)
                    e1.record(st)
                    us = e0.elapsed_ms(e1) * 1e3
                    rd(buf.ctypes.data, buf.size, 1)
                except Exception as e:
                    print("%-26s %-10s %3d | %s" % (name, tname, nst, str(e)[:80]))
                    continue
                r = buf.reshape(-1, 8)
                r = r[r[:, 5] > 0]
                if not len(r):
                    print("%-26s %-10s %3d | no trace rows (kernel is not conv_igemm2?)" % (name, tname, nst))
                    continue
                nk = r[:, 5].mean()
                tot, vm, bar, iss, cmp_ = [r[:, i].mean() for i in range(5)]
                mhz = (r[:, 0] / np.maximum(r[:, 6], 1)).mean() * 100.0
                print("%-26s %-10s %3d | %6d %7.0f | %6.0f %6.0f %6.0f %6.0f | %5.0f %5.0f" %
                      (name, tname, nst, len(r) * (8 if tname == "v3" else 32), tot / nk, vm / nk, bar / nk, iss / nk, cmp_ / nk, mhz, us), flush=True)


if __name__ == "__main__":
    main()
