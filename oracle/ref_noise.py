"""TEST INFRASTRUCTURE (see oracle/__init__.py).

How far do the REFERENCE's own joints move when nothing changes but the order of its fp32 sums?  Runs /root/reference (CPU, fp32) on a volumetric fixture's
inputs with 8 threads and with 1 thread (oneDNN / ATen split their reductions differently) and prints, in the units of the north-star gate (max over
coordinates of |d| / max(|ref|, 1 mm)):  the two runs against each other, and each run against the exact (fp64) soft-argmax of the OTHER run's logits --
the quantity tests/test_gpu_models.py gates a kernel on.  A fixture on which the reference misses 1e-4 against itself cannot hold a kernel to 1e-4.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.ref_noise c2_b8_sharp
"""
import sys

import numpy as np
import torch

from . import ref_loader, spec, synth
from .make_golden import _cameras

CASES = {   # tag: (num_layers, B, NV, H, V, sharpen, seed) -- the arguments make_golden.py generates the fixture with
    "c2_b8_sharp": (152, 8, 4, 384, 64, 150.0, 11),
    "c2_sharp": (152, 2, 4, 384, 64, True, 0),
}


def run(mvn, tag, threads):
    nl, B, NV, H, V, sharpen, seed = CASES[tag]
    torch.set_num_threads(threads)
    cfg = synth.vol_config(nl, V, "softmax", 1.0, "mpii")
    sd = synth.make_state_dict(spec.vol_net_spec(nl, 17, False), seed=seed, sharpen=sharpen)
    inp = synth.make_inputs(B, NV, H, seed=seed)
    ref = mvn.models.triangulation.VolumetricTriangulationNet(cfg, device="cpu")
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    cap = {}
    ref.volume_net.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach()))
    batch = {"cameras": _cameras(mvn, inp["K"], inp["R"], inp["t"], B), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
    with torch.no_grad():
        kp, feats, vols, volc, cuboids, cvs, bps = ref(inp["images"], torch.zeros(B, NV, 3, 4), batch)
    lg = cap["logits"].double().reshape(B, 17, -1)
    kp64 = torch.einsum("bjn,bnc->bjc", torch.softmax(lg, dim=2), cvs.double().reshape(B, -1, 3))
    return kp.double(), kp64


def rel(a, b):
    return float(((a - b).abs() / b.abs().clamp(min=1.0)).max())


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "c2_b8_sharp"
    mvn = ref_loader.load()
    kp8, ex8 = run(mvn, tag, 8)
    kp1, ex1 = run(mvn, tag, 1)
    print("%s: reference, 8 threads vs 1 thread: joints max rel %.3e" % (tag, rel(kp1, kp8)))
    print("   exact soft-argmax of the 1-thread logits vs exact soft-argmax of the 8-thread logits: %.3e   <- fp32 noise of the LAYERS, amplified by the soft-argmax" % rel(ex1, ex8))
    print("   1-thread joints vs exact(8-thread logits): %.3e;  8-thread joints vs exact(1-thread logits): %.3e" % (rel(kp1, ex8), rel(kp8, ex1)))
    print("   (each run against the exact soft-argmax of its OWN logits: %.3e / %.3e)" % (rel(kp8, ex8), rel(kp1, ex1)))


if __name__ == "__main__":
    main()
