#!/usr/bin/env python
"""Per-block timing of lt_bottleneck_fwd against the three lt_conv_fwd launches it replaces (ResNet layer1 / layer2 identity blocks at
the benchmark's 128 images), on post-ReLU-like data.  Usage: python tools/bneck_bench.py [--images 128] [--rounds 5]"""
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
    return (0.5 + torch.rand(c, generator=g), torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, 0.5 + torch.rand(c, generator=g))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--trace", action="store_true", help="library built with -DLT_BNECK_TRACE (LT_HIP_LIB=...): print the phase durations of the last fused launch")
    args = ap.parse_args()
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    for C, P, S in ((256, 64, 96), (512, 128, 48)):
        g = torch.Generator().manual_seed(C)
        ws = [torch.randn(P, C, 1, 1, generator=g) / C ** 0.5, torch.randn(P, P, 3, 3, generator=g) / (9 * P) ** 0.5, torch.randn(C, P, 1, 1, generator=g) / P ** 0.5]
        bns = [bn(P, g), bn(P, g), bn(C, g)]
        x = torch.relu(torch.randn(args.images, 1, S, S, C, generator=g)).to(dev, torch.bfloat16)
        res = {}
        for fused in (True, False):
            b = E.PlanBuilder(dev, torch.bfloat16)
            xa = E.Act(x)
            if fused:
                y = b.bottleneck(xa, ws, bns)
            else:
                os.environ["LT_NO_BNECK"] = "1"
                t1 = b.conv(xa, ws[0], None, bns[0], relu=True)
                t2 = b.conv(t1, ws[1], None, bns[1], pad=1, relu=True)
                y = b.conv(t2, ws[2], None, bns[2], relu=True, residual=xa)
                os.environ.pop("LT_NO_BNECK")
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
            res[fused] = (best, y.t.float())
        fl = plan.flops
        if args.trace:
            import ctypes
            import numpy as np
            lib = H.lib()
            plan_f.run_eager(st); torch.cuda.synchronize()
            nwg = min(4096, args.images * (S // 8) * (S // 16))
            buf = np.zeros(4096 * 4 * 8, dtype=np.uint64)
            rc = lib.lt_bneck_trace_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
            assert rc == 0, rc
            tr = buf.reshape(4096, 4, 8)[:nwg].astype(np.int64)
            d = np.diff(tr[:, :, :6], axis=2).astype(np.float64)          # phase durations per wave
            names = ["p1 loop", "p1 epi+bar", "p2 loop", "p2 epi+bar", "p3"]
            print("   trace (%d workgroups, cycles, mean over waves): " % nwg + ", ".join("%s %.0f" % (n, d[:, :, i].mean()) for i, n in enumerate(names)) +
                  " | total %.0f; kernel span %.0f cycles" % ((tr[:, :, 5] - tr[:, :, 0]).mean(), float(tr[:, :, 5].max() - tr[:, :, 0].min())), flush=True)
        d = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
        print("bneck %d/%d @%dx%dx%d: fused %.1f us (%.0f TFLOP/s, %.2f TB/s of x+y) | three launches %.1f us | max rel diff %.2e" % (
            C, P, args.images, S, S, 1e3 * res[True][0], fl / res[True][0] / 1e9, 2 * x.numel() * 2 / res[True][0] / 1e9, 1e3 * res[False][0], d), flush=True)


if __name__ == "__main__":
    main()
