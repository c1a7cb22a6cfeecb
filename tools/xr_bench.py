#!/usr/bin/env python
"""Timing of lt_expand_reduce_fwd (the seam between two identity blocks of ResNet layer3: expand + reduce in one launch) against the two lt_conv_fwd
launches it replaces, at the benchmark's image count, on post-ReLU-like data; with a library built with -DLT_XR_TRACE (LT_HIP_LIB=...) also where the
waves of the fused kernel spend their time.  Usage: python tools/xr_bench.py [--images 256] [--rounds 5] [--trace]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "learnable-triangulation-pytorch_amd")):
    sys.path.insert(0, p)
import torch

import lt_engine as E
import lt_hip as H


def bn(c, g):
    return (0.2 + 0.4 * torch.rand(c, generator=g), torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, 0.5 + torch.rand(c, generator=g))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--trace", action="store_true")
    args = ap.parse_args()
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    C_, P, S = 1024, 256, 24
    g = torch.Generator().manual_seed(3)
    w3, w1 = torch.randn(C_, P, 1, 1, generator=g) / P ** 0.5, torch.randn(P, C_, 1, 1, generator=g) / C_ ** 0.5
    bn3, bn1 = bn(C_, g), bn(P, g)
    t2 = torch.relu(torch.randn(args.images, 1, S, S, P, generator=g)).to(dev, torch.bfloat16)
    res = torch.relu(torch.randn(args.images, 1, S, S, C_, generator=g)).to(dev, torch.bfloat16)
    out = {}
    for fused in (True, False):
        if not fused:
            os.environ["LT_NO_XR"] = "1"
        b = E.PlanBuilder(dev, torch.bfloat16)
        ta, ra = E.Act(t2), E.Act(res)
        if fused:
            y, t1 = b.expand_reduce(ta, ra, w3, bn3, w1, bn1)
        else:
            y = b.conv(ta, w3, None, bn3, relu=True, residual=ra)
            t1 = b.conv(y, w1, None, bn1, relu=True)
        os.environ.pop("LT_NO_XR", None)
        plan = b.finish()
        if fused:
            plan_f = plan
        for _ in range(3):
            plan.run_eager(st)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(args.rounds):
            e0, e1 = H.Event(), H.Event()
            e0.record(st)
            for _ in range(10):
                plan.run_eager(st)
            e1.record(st)
            best = min(best, e0.elapsed_ms(e1) / 10)
        out[fused] = (best, y.t.float(), t1.t.float())
    M = args.images * S * S
    traffic = (M * P + 2 * M * C_ + M * P) * 2
    d = max(float((out[True][i] - out[False][i]).abs().max() / out[False][i].abs().max()) for i in (1, 2))
    print("xr @%d images (%d rows): fused %.1f us (%.2f TB/s of its %.0f MB, %.0f TFLOP/s) | two launches %.1f us | max rel diff %.2e" % (
        args.images, M, 1e3 * out[True][0], traffic / out[True][0] / 1e9, traffic / 1e6, plan_f.flops / out[True][0] / 1e9, 1e3 * out[False][0], d), flush=True)
    if args.trace:
        import ctypes
        import numpy as np
        lib = H.lib()
        if not hasattr(lib, "lt_xr_trace_read"):
            print("   (no lt_xr_trace_read in this library: build the variant with -DLT_XR_TRACE and point LT_HIP_LIB at it)")
            return
        plan_f.run_eager(st); torch.cuda.synchronize()
        lib.lt_xr_trace_read.restype = ctypes.c_int
        lib.lt_xr_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
        nwg = min(2048, (M + 95) // 96)
        buf = np.zeros(2048 * 8 * 32, dtype=np.uint64)
        rc = lib.lt_xr_trace_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
        assert rc == 0, rc
        tr = buf.reshape(2048, 8, 32)[:nwg].astype(np.int64)
        ex, rd = tr[:, :4], tr[:, 4:]
        print("   trace over %d workgroups (clock ticks of s_memtime, 100 MHz domain x ... see ratio to the kernel span): kernel span %d ticks" % (nwg, int(tr[:, :, 1:27].max() - tr[:, :, 0].min())))
        print("   prologue (entry -> t2 tile + constants in LDS): %.0f" % (tr[:, :, 1] - tr[:, :, 0]).mean())
        for c in range(8):
            e_loop = (ex[:, :, 2 + 3 * c] - (ex[:, :, 1] if c == 0 else ex[:, :, 4 + 3 * (c - 1)])).mean()
            e_epi = (ex[:, :, 3 + 3 * c] - ex[:, :, 2 + 3 * c]).mean()
            e_bar = (ex[:, :, 4 + 3 * c] - ex[:, :, 3 + 3 * c]).mean()
            r_bar = (rd[:, :, 2 + 3 * c] - (rd[:, :, 1] if c == 0 else rd[:, :, 4 + 3 * (c - 1)])).mean()
            r_copy = (rd[:, :, 3 + 3 * c] - rd[:, :, 2 + 3 * c]).mean()
            r_loop = (rd[:, :, 4 + 3 * c] - rd[:, :, 3 + 3 * c]).mean()
            print("   chunk %d: expander MFMA loop %6.0f  epilogue %5.0f  barrier wait %6.0f | reducer barrier wait %6.0f  copy-out %5.0f  MFMA loop %6.0f" % (
                c, e_loop, e_epi, e_bar, r_bar, r_copy, r_loop))
        print("   whole tile (entry -> last reducer stamp): %.0f" % (rd[:, :, 25] - tr[:, 4:, 0]).mean())


if __name__ == "__main__":
    main()
