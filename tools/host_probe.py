"""Is the recorded training step bound by the host that replays it?  python tools/host_probe.py [B] [precision]
Per step, WITHOUT synchronising inside: host time to issue the forward, the loss, the backward and the optimiser step; then the wall time of
the same steps end to end.  host issue ~ wall: the GPU waits for the host (python closures + ctypes calls, ~2500 launches per step)."""
import os, sys, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "learnable-triangulation-pytorch_amd")); sys.path.insert(0, R)
import bench, lt_train
from mvn.models.triangulation import VolumetricTriangulationNet
from mvn.models import loss as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PREC = sys.argv[2] if len(sys.argv) > 2 else "act16"
dev = torch.device("cuda:0")
m = VolumetricTriangulationNet(bench.vol_config(152, 64, "fp32"), device=dev); m.to(dev); m.train()
m.train_precision = PREC
images, batch, geom = bench.synthetic_batch(B, 4, 384, 1000)
images = images.to(dev)
opt = lt_train.Adam([{"params": list(m.backbone.parameters())}, {"params": list(m.process_features.parameters()), "lr": 1e-3},
                     {"params": list(m.volume_net.parameters()), "lr": 1e-3}], lr=1e-4)
gt = (torch.as_tensor(np.asarray(batch["pred_keypoints_3d"]))[:, :, :3].float()).to(dev)
val = torch.ones(B, 17, 1, device=dev)
mae, ce = L.KeypointsMAELoss(), L.VolumetricCELoss()
pc = time.perf_counter
def step(t):
    t.append(pc()); kp, _, vols, _, _, cvs, _ = m(images, None, batch)
    t.append(pc()); loss = mae(kp * 0.1, gt * 0.1, val) + 0.01 * ce(cvs, vols, gt, val)
    t.append(pc()); opt.zero_grad(); loss.backward()
    t.append(pc()); opt.step()
    t.append(pc())
for _ in range(3):
    step([])
torch.cuda.synchronize()
N = 8
ts = []
t0 = pc()
for _ in range(N):
    t = []; step(t); ts.append(t)
t1 = pc()
torch.cuda.synchronize()
t2 = pc()
a = np.diff(np.array(ts), axis=1) * 1e3
print("host issue per step (ms): forward %.2f  loss %.2f  backward %.2f  optimiser %.2f  | sum %.2f" % (*a.mean(0), a.sum(1).mean()))
print("host loop %.2f ms/step, wall incl. final drain %.2f ms/step (GPU behind the host by %.2f ms at the end)" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3, (t2 - t1) * 1e3))
for row in a[:4]:
    print("   ", " ".join("%7.2f" % v for v in row))
