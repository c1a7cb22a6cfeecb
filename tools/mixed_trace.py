"""Layer-by-layer comparison of the mixed-precision training forward (bf16 MFMA convolutions) with the fp32 one at the benchmark's shape
(GPU box): python tools/mixed_trace.py [B] [resnet depth]  -- same weights, inputs and cuboid rotations; prints, for every convolution layer
in forward order, max|d| / max|ref| and rms(d) / rms(ref) of its (post-BatchNorm / activation) output.  A smooth growth with depth is rounding;
a jump at one layer would be a kernel problem.  Written for VERDICT r2 weak 7 (the mixed step's loss leaves the fp32 curve at ResNet-152 scale)."""
import os, sys, json, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "learnable-triangulation-pytorch_amd")); sys.path.insert(0, R)
import bench
from mvn.models.triangulation import VolumetricTriangulationNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 152
dev = torch.device("cuda:0")
images, batch, geom = bench.synthetic_batch(B, 4, 384, 1000)
images = images.to(dev)
outs = {}
# third run: fp32 again with the images perturbed by 2^-9 relative noise (the size of ONE bf16 rounding, applied ONCE at the input): how much of the
# divergence is the network's own sensitivity (a random-init ResNet-152 under batch-statistics BatchNorm) rather than anything the bf16 kernels add
pert = images * (1.0 + (2.0 ** -9) * torch.randn(images.shape, generator=torch.Generator().manual_seed(9)).to(dev))
for prec in ("fp32", "bf16", "fp32-perturbed"):
    torch.manual_seed(0)
    m = VolumetricTriangulationNet(bench.vol_config(NL, 64, "fp32"), device=dev)
    with torch.no_grad():
        m.volume_net.output_layer.weight.mul_(bench.SHARPEN)
    m.to(dev).train()
    m.train_precision = "bf16" if prec == "bf16" else "fp32"
    np.random.seed(1234)
    kp = m(pert if prec == "fp32-perturbed" else images, None, batch)[0]
    torch.cuda.synchronize()
    tape = list(m._train_plans.values())[0].tape
    outs[prec] = ([(lab, a.t.detach().float().clone()) for lab, a in tape.layer_outputs], kp.detach().clone())
    del m, tape
    torch.cuda.empty_cache()
a, b, c = outs["fp32"][0], outs["bf16"][0], outs["fp32-perturbed"][0]
assert len(a) == len(b) == len(c)
rows = []
for i, ((lab, x), (_, y), (_, z)) in enumerate(zip(a, b, c)):
    d = (y - x)
    rms = lambda t: float(t.pow(2).mean().sqrt())
    rows.append((i, lab, tuple(x.shape), float(d.abs().max() / x.abs().max().clamp(min=1e-30)), rms(d) / max(rms(x), 1e-30), rms(z - x) / max(rms(x), 1e-30)))
print("layer  bf16: max|d|/max|ref|  rms(d)/rms(ref)   fp32 with 2^-9 input noise: rms(d)/rms(ref)   label shape")
prev = 0.0
for i, lab, shp, e, r, rp in rows:
    flag = "  <-- jump" if r > 3 * max(prev, 1e-3) and r > 2e-2 else ""
    if i < 12 or i % 10 == 0 or flag or i > len(rows) - 12:
        print("%4d   %.3e        %.3e        %.3e        %s %s%s" % (i, e, r, rp, lab, shp, flag))
    prev = r
kp32, kp16, kpp = outs["fp32"][1], outs["bf16"][1], outs["fp32-perturbed"][1]
print("keypoints vs fp32: bf16 mean |d| %.1f mm; fp32 with 2^-9 input noise mean |d| %.1f mm" % (float((kp16 - kp32).abs().mean()), float((kpp - kp32).abs().mean())))
json.dump([{"i": i, "label": lab, "shape": shp, "bf16_max_rel": e, "bf16_rms_rel": r, "fp32_input_noise_rms_rel": rp} for i, lab, shp, e, r, rp in rows], open(os.path.join(R, "gpurun_out", "mixed_trace_b%d_r%d.json" % (B, NL)), "w"), indent=0)
