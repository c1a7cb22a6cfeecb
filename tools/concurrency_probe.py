#!/usr/bin/env python
"""How much of a small-batch forward is latency the GPU could hide with a second, concurrent stream?  Two model instances (same
weights), B samples each, their forwards launched (a) one after the other on one stream, (b) on two streams at once; compared with one
forward of 2 B samples.  python tools/concurrency_probe.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "learnable-triangulation-pytorch_amd")]
import bench
from mvn.models.triangulation import VolumetricTriangulationNet


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    models = [VolumetricTriangulationNet(bench.vol_config(152, 64, "bf16"), device=dev) for _ in range(2)]
    models[1].load_state_dict(models[0].state_dict())
    for m in models:
        m.to(dev).eval()
        m.copy_outputs = False
        m.use_graph = os.environ.get("PROBE_EAGER") != "1"
    images, batch, _ = bench.synthetic_batch(2 * B, 4, 384, 1000)
    images = images.to(dev)
    halves = [(images[:B].contiguous(), {"cameras": [c[:B] for c in batch["cameras"]], "pred_keypoints_3d": batch["pred_keypoints_3d"][:B]}),
              (images[B:].contiguous(), {"cameras": [c[B:] for c in batch["cameras"]], "pred_keypoints_3d": batch["pred_keypoints_3d"][B:]})]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def timed(fn, n=20):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    with torch.no_grad():
        def seq():
            for m, (im, bt) in zip(models, halves):
                m(im, None, bt)

        def conc():
            for m, (im, bt), st in zip(models, halves, streams):
                with torch.cuda.stream(st):
                    m(im, None, bt)

        def whole():
            models[0](images, None, batch)

        t_seq, t_conc, t_whole = timed(seq), timed(conc), timed(whole)
        # host time of one forward call (returns before the GPU is done unless something in it waits)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        models[0](*[halves[0][0], None, halves[0][1]])
        t_host = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        print("host time of one forward call: %.2f ms" % t_host)
    print("B = %d per forward: two forwards one after the other %.2f ms, on two streams %.2f ms, one forward of %d samples %.2f ms" % (B, t_seq, t_conc, 2 * B, t_whole))
    print("samples/s: sequential %.0f, concurrent %.0f, one plan %.0f" % (2e3 * B / t_seq, 2e3 * B / t_conc, 2e3 * B / t_whole))


if __name__ == "__main__":
    main()
