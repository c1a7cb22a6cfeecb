#!/usr/bin/env python
"""Where do the device-to-device copies of a training step come from?  (~840 __amd_rocclr_copyBuffer per mixed step.)
Runs a few steps of the small training fixture under torch.profiler and prints, per step, the aten ops that end in a copy kernel with
their Python call sites."""
import os
import sys
import collections

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "learnable-triangulation-pytorch_amd"), os.path.join(ROOT, "tests")]


def main():
    import lt_train
    from mvn.models import loss as L
    from mvn.models.triangulation import VolumetricTriangulationNet
    from test_gpu_train import _train_case
    from test_gpu_models import _cameras
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    DEV = "cuda:0"
    c, cfg, sd, inp = _train_case()
    m = VolumetricTriangulationNet(cfg, device=DEV)
    m.load_state_dict(sd, strict=True)
    m.to(DEV); m.train()
    m.train_precision = prec
    batch = {"cameras": _cameras(inp, c["B"]), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    opt = lt_train.Adam(list(m.parameters()), lr=1e-4)
    gt = torch.as_tensor(np.asarray(inp["pred_keypoints_3d"]))[:, :, :3].float().to(DEV)
    val = torch.ones(c["B"], 17, 1, device=DEV)
    images = inp["images"].to(DEV)

    def step():
        kp, _, vols, _, _, cvs, _ = m(images, None, batch)
        loss = L.KeypointsMAELoss()(kp * 0.1, gt * 0.1, val) + 0.01 * L.VolumetricCELoss()(cvs, vols, gt, val)
        opt.zero_grad(); loss.backward(); opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    ev = prof.events()
    kern = collections.Counter()
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CUDA or "emcpy" in e.name or "copyBuffer" in e.name:
            kern[e.name[:80]] += 1
    print("== device-side events of one step (top 25)")
    for k, n in kern.most_common(25):
        print("%6d  %s" % (n, k))
    ops = collections.Counter()
    for e in ev:
        if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::add_", "aten::zeros", "aten::fill_", "aten::zero_"):
            st = [f for f in (e.stack or []) if "site-packages" not in f and "<built-in" not in f][:3]
            ops[(e.name, " <- ".join(s.strip()[-90:] for s in st))] += 1
    print("== aten ops with their call sites (top 40)")
    for (n, s), k in ops.most_common(40):
        print("%5d %-16s %s" % (k, n, s))


if __name__ == "__main__":
    main()
