"""Stand-alone timing of lt_unproject_bwd at the BASELINE config-2 shape (GPU box): python tools/unproj_bwd_bench.py [B] [resnet depth]
Builds features / projections / coordinate volumes with the model's own forward (eval), then times op.unproject_heatmaps' backward
(the deterministic gather) and, with LT_UNPROJ_BWD_ATOMICS=1, the round-2 scatter.  Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel split (unproj_bbox_kernel / unproj_dx_kernel / unproj_gather_kernel)."""
import os, sys, json, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "learnable-triangulation-pytorch_amd")); sys.path.insert(0, R)
import bench
from mvn.models.triangulation import VolumetricTriangulationNet
from mvn.utils import op
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 18
dev = torch.device("cuda:0")
m = VolumetricTriangulationNet(bench.vol_config(NL, 64, "fp32"), device=dev); m.to(dev); m.eval()
images, batch, geom = bench.synthetic_batch(B, 4, 384, 1000)
with torch.no_grad():
    kp, feats, vols, _, _, cvs, _ = m(images.to(dev), None, batch)
P = list(m._plans.values())[0]["geo"][:B * 4 * 12].reshape(B, 4, 3, 4).clone()
feats = feats.clone()
G = torch.randn(B, 32, 64, 64, 64, device=dev)
del m, vols
torch.cuda.empty_cache()
res = {}
for mode in ("gather", "scatter"):
    if mode == "scatter":
        os.environ["LT_UNPROJ_BWD_ATOMICS"] = "1"
    ts = []
    for it in range(4):
        f = feats.clone().requires_grad_(True)
        vol = op.unproject_heatmaps(f, P, cvs, "softmax")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); vol.backward(G); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res[mode] = {"ms_backward_incl_host_glue": min(ts), "all": ts, "grad_absmax": float(f.grad.abs().max())}
    os.environ.pop("LT_UNPROJ_BWD_ATOMICS", None)
print(json.dumps({"B": B, "shape": "4 views 96x96x32 fp32, 64^3 voxels", **res}))
