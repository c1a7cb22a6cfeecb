#!/usr/bin/env python
"""Per-layer convolution micro-benchmark (GPU): times lt_conv_fwd on the layer shapes of BASELINE config 2 for every valid
tile / ring-depth combination inside ONE process (interleaved rounds, hipEvents on the launch stream) and prints a table.
Usage: python tools/conv_bench.py [--batch 8] [--dtype bf16] [--rounds 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learnable-triangulation-pytorch_amd"))
import torch

import lt_engine as E
import lt_hip as H

TILES = {"v1_128x128": 1, "v1_128x64": 2, "v1_256x32": 3, "v1_256x16": 4, "v1_64x64": 5,
         "128x128": 11, "128x64": 12, "256x32": 13, "256x16": 14, "64x64": 15}


def shapes(B):
    NV = 4 * B
    return [
        # name, nd, N, cin, cout, k, stride, pad, spatial, transposed
        ("rn l3 1x1 1024->256 @24", 2, NV, 1024, 256, 1, 1, 0, (24, 24), False),
        ("rn l3 1x1 256->1024 @24", 2, NV, 256, 1024, 1, 1, 0, (24, 24), False),
        ("rn l3 3x3 256->256 @24", 2, NV, 256, 256, 3, 1, 1, (24, 24), False),
        ("rn l2 1x1 512->128 @48", 2, NV, 512, 128, 1, 1, 0, (48, 48), False),
        ("rn l2 1x1 128->512 @48", 2, NV, 128, 512, 1, 1, 0, (48, 48), False),
        ("rn l2 3x3 128->128 @48", 2, NV, 128, 128, 3, 1, 1, (48, 48), False),
        ("rn l1 1x1 64->256 @96", 2, NV, 64, 256, 1, 1, 0, (96, 96), False),
        ("rn l1 3x3 64->64 @96", 2, NV, 64, 64, 3, 1, 1, (96, 96), False),
        ("rn l4 3x3 512->512 @12", 2, NV, 512, 512, 3, 1, 1, (12, 12), False),
        ("rn deconv 256->256 @48", 2, NV, 256, 256, 4, 2, 1, (48, 48), True),
        ("v2v 3^3 32->32 @64", 3, B, 32, 32, 3, 1, 1, (64, 64, 64), False),
        ("v2v 7^3 32->16 @64", 3, B, 32, 16, 7, 1, 3, (64, 64, 64), False),
        ("v2v 7^3 16->32 @64 (the front layer's input gradient)", 3, B, 16, 32, 7, 1, 3, (64, 64, 64), False),
        ("v2v 3^3 64->64 @32", 3, B, 64, 64, 3, 1, 1, (32, 32, 32), False),
        ("v2v 3^3 128->128 @16", 3, B, 128, 128, 3, 1, 1, (16, 16, 16), False),
        ("v2v 3^3 128->128 @4", 3, B, 128, 128, 3, 1, 1, (4, 4, 4), False),
        ("v2v 1^3 32->32 @64", 3, B, 32, 32, 1, 1, 0, (64, 64, 64), False),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--json", default="")
    ap.add_argument("--only", default="", help="comma list of substrings: keep layers whose name contains one of them")
    ap.add_argument("--residual", action="store_true", help="add a residual input (the epilogue of the res-block layers)")
    ap.add_argument("--variants", default="", help="comma list of variant names to keep (default all)")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    results = []
    for (name, nd, N, cin, cout, k, s, p, sp, tr) in shapes(args.batch):
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        cp = E.cout_pad_of(cout)
        tiles = [t for t, bn in (("128x128", 128), ("128x64", 64), ("64x64", 64), ("256x32", 32), ("256x16", 16)) if cp % bn == 0 and bn <= cp]
        variants = [("auto", 0, 0), ("halo", 20, 0), ("v3", 30, 0)] + [("%s/s%d" % (t, n), TILES[t], n) for t in tiles for n in (2, 3)] + [("v1_" + t, TILES["v1_" + t], 0) for t in tiles[:2]]
        if args.variants:
            keep = set(args.variants.split(","))
            variants = [v for v in variants if v[0] in keep]
        x = torch.randn(N, *( (1,) if nd == 2 else ()), *sp, cin, device=dev).to(dt)
        w = torch.randn(*((cin, cout) if tr else (cout, cin)), *([k] * nd)) * 0.05
        res = None
        if args.residual and not tr:
            osp = [(d + 2 * p - k) // s + 1 for d in sp]
            res = E.Act(torch.randn(N, *((1,) if nd == 2 else ()), *osp, cout, device=dev).to(dt))
        plans = []
        for vname, tile, nst in variants:
            b = E.PlanBuilder(dev, dt, tile_override=tile, stages=nst)
            xa = E.Act(x)
            try:
                y = b.conv(xa, w, None, None, stride=s, pad=p, transposed=tr, relu=True, residual=res)
                plans.append((vname, b.finish(), b.flops))
            except Exception as e:  # LDS too large etc.
                plans.append((vname, None, str(e)))
        times = {v: [] for v, _, _ in plans}
        for r in range(args.rounds + 1):
            for vname, plan, fl in plans:
                if plan is None:
                    continue
                try:
                    e0, e1 = H.Event(), H.Event()
                    e0.record(st)
                    for _ in range(5):
                        plan.run_eager(st)
                    e1.record(st)
                    ms = e0.elapsed_ms(e1) / 5
                except RuntimeError as e:
                    ms = float("nan")
                if r > 0:
                    times[vname].append(ms)
        flops = [f for _, pl, f in plans if pl is not None][0]
        row = {"layer": name, "gflop": flops / 1e9}
        line = "%-26s %7.1f GF |" % (name, flops / 1e9)
        for vname, plan, _ in plans:
            if plan is None or not times[vname]:
                continue
            ms = sorted(times[vname])[len(times[vname]) // 2]
            row[vname] = {"us": ms * 1e3, "tflops": flops / (ms * 1e-3) / 1e12}
            line += " %s %.0fus %.0fTF |" % (vname, ms * 1e3, flops / (ms * 1e-3) / 1e12)
        print(line, flush=True)
        results.append(row)
    if args.json:
        json.dump(results, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
