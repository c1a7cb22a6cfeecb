#!/usr/bin/env python
"""Do liblt_hip kernels launched on two streams overlap?  A chain of identical convolution launches (few workgroups each) captured into a graph per
stream; time of the two graphs one after the other vs at once."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "learnable-triangulation-pytorch_amd")]
import lt_engine as E
import lt_hip as H

dev = "cuda:0"


def chain(N, cin, cout, k, hw, reps, tile=0):
    b = E.PlanBuilder(dev, torch.bfloat16, tile_override=tile)
    x = E.Act(torch.randn(N, 1, hw, hw, cin, device=dev).bfloat16())
    w = torch.randn(cout, cin, k, k) * 0.05
    for _ in range(reps):
        b.conv(x, w, None, None, stride=1, pad=k // 2, relu=True)
    return b.finish()


def main():
    streams = [torch.cuda.Stream() for _ in range(2)]
    for name, args in (("3x3 256->256 @24^2, 4 images (64x64 tiles: 36 x 4 workgroups)", (4, 256, 256, 3, 24, 40)),
                       ("1x1 256->1024 @24^2, 4 images", (4, 256, 1024, 1, 24, 40)),
                       ("3x3 256->256 @24^2, 32 images", (32, 256, 256, 3, 24, 20))):
        plans = [chain(*args) for _ in range(2)]
        for p, st in zip(plans, streams):
            with torch.cuda.stream(st):
                p.run_eager(st.cuda_stream)
                st.synchronize()
                p.capture(st.cuda_stream)
        def run(conc):
            for p, st in zip(plans, streams if conc else [streams[0]] * 2):
                p.graph.launch(st.cuda_stream)
        res = []
        for conc in (False, True):
            run(conc); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                run(conc)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 5 * 1e3)
        print("%-70s one stream %.3f ms, two streams %.3f ms" % (name, res[0], res[1]))


if __name__ == "__main__":
    main()
