#!/usr/bin/env python
"""The shader clock and socket power the chip settles at under ONE kernel run back to back for a few seconds (bench.ClockSampler: rocm-smi polled from its own
process), for the four kernels that carry most of the bf16 forward and two of the exact-fp32 ones, at the benchmark's sizes (64 samples = 256 images).
TFLOP/s are 2 x MAC / time at that clock; 'at 2.4 GHz' rescales them to the clock the MFMA peaks assume.   python tools/clock_per_kernel.py [--seconds 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "learnable-triangulation-pytorch_amd")):
    sys.path.insert(0, p)
import torch

import bench
import lt_engine as E
import lt_hip as H


def bn(c, g):
    return (0.2 + 0.4 * torch.rand(c, generator=g), torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, 0.5 + torch.rand(c, generator=g))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    args = ap.parse_args()
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    cases = [   # name, dtype, N, spatial, cin, cout, k, pad, residual
        ("layer3 3x3 256->256 (conv2d_halo)", torch.bfloat16, 256, (1, 24, 24), 256, 256, 3, 1, False),
        ("V2V 3^3 32->32 @64^3 (column walk)", torch.bfloat16, 64, (64, 64, 64), 32, 32, 3, 1, True),
        ("V2V 7^3 32->16 @64^3 (halo7b)", torch.bfloat16, 64, (64, 64, 64), 32, 16, 7, 3, False),
        ("layer3 1x1 256->1024 (igemm6)", torch.bfloat16, 256, (1, 24, 24), 256, 1024, 1, 0, True),
        ("fp32 V2V 3^3 32->32 @64^3 (two phases)", torch.float32, 64, (64, 64, 64), 32, 32, 3, 1, True),
        ("fp32 layer3 3x3 256->256 (128 x 128 tile)", torch.float32, 256, (1, 24, 24), 256, 256, 3, 1, False),
    ]
    for name, dt, N, sp, cin, cout, k, pad, has_res in cases:
        b = E.PlanBuilder(dev, dt)
        nd = 2 if sp[0] == 1 else 3
        x = E.Act(torch.relu(torch.randn(N, *sp, cin, generator=g)).to(dev, dt))
        res = E.Act(torch.relu(torch.randn(N, *sp, cout, generator=g)).to(dev, dt)) if has_res else None
        w = torch.randn(cout, cin, *([k] * nd), generator=g) / (cin * k ** nd) ** 0.5
        b.conv(x, w, None, bn(cout, g), stride=1, pad=pad, relu=True, residual=res)
        plan = b.finish()
        for _ in range(3):
            plan.run_eager(st)
        torch.cuda.synchronize()
        smp = bench.ClockSampler(0)
        t0 = time.time()
        n = 0
        while time.time() - t0 < args.seconds:
            for _ in range(20):
                plan.run_eager(st)
            torch.cuda.synchronize()
            n += 20
        t1 = time.time()
        # the last 60 % of the loop: the clock has settled
        tw = t0 + 0.4 * (t1 - t0)
        e0, e1 = H.Event(), H.Event()
        e0.record(st)
        for _ in range(20):
            plan.run_eager(st)
        e1.record(st)
        ms = e0.elapsed_ms(e1) / 20
        clk = smp.window(tw, time.time()) or {}
        tf = plan.flops / ms / 1e9
        peak = 2500.0 if dt == torch.bfloat16 else 157.3
        s = clk.get("sclk_mhz")
        print("%-44s %8.1f us  %7.1f TFLOP/s = %.3f of the peak | sclk %s MHz, %s W (%s reads)%s" % (
            name, 1e3 * ms, tf, tf / peak, s, clk.get("power_w"), clk.get("samples"), ("  -> %.3f at the measured clock" % (tf / peak * 2400.0 / s)) if s else ""), flush=True)
        del plan, b, x, res
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
