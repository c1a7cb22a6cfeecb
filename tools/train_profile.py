"""Per-op timing of the recorded training step (GPU box): python tools/train_profile.py [B] [fp32|bf16]  -> table grouped by op kind and the
most expensive weight-gradient / convolution launches."""
import os, sys, json, collections, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "learnable-triangulation-pytorch_amd")); sys.path.insert(0, R)
import bench
from mvn.models.triangulation import VolumetricTriangulationNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PREC = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda:0")
m = VolumetricTriangulationNet(bench.vol_config(152, 64, "fp32"), device=dev); m.to(dev); m.train()
m.train_precision = PREC
images, batch, geom = bench.synthetic_batch(B, 4, 384, 1000)
images = images.to(dev)
for _ in range(2):
    out = m(images, None, batch)
    out[0].sum().backward()
torch.cuda.synchronize()
tape = list(m._train_plans.values())[0].tape
res = {}
for name, ops in (("fwd", tape.fwd_ops), ("bwd", tape.bwd_ops)):
    prof = tape.profile(ops)
    kinds = collections.defaultdict(float)
    for lab, ms in prof:
        kinds[lab.split(" ")[0]] += ms
    print(name, "total %.1f ms:" % sum(ms for _, ms in prof), {k: round(v, 2) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])})
    agg = collections.defaultdict(lambda: [0, 0.0])
    for lab, ms in prof:
        agg[lab][0] += 1; agg[lab][1] += ms
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]
    for lab, (n, ms) in top:
        print("   %8.3f ms  x%-3d %s" % (ms, n, lab))
    res[name] = [(lab, n, ms) for lab, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])]
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(R, "gpurun_out", "train_ops_b%d%s.json" % (B, "" if PREC == "fp32" else "_" + PREC)), "w"), indent=0)
