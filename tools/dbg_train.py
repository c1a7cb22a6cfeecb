import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo/learnable-triangulation-pytorch_amd"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from oracle import spec, synth
import lt_train, lt_engine as E
orig = lt_train.TrainTape._run_last
def run_last(self):
    orig(self); torch.cuda.synchronize(); print("ok", self.pb.ops[-1][1]["label"], flush=True)
lt_train.TrainTape._run_last = run_last
from test_gpu_train import _train_case
from test_gpu_models import _cameras
from mvn.models.triangulation import VolumetricTriangulationNet
c, cfg, sd, inp = _train_case()
m = VolumetricTriangulationNet(cfg, device="cuda:0"); m.load_state_dict(sd); m.to("cuda:0"); m.train()
batch = {"cameras": _cameras(inp, c["B"]), "pred_keypoints_3d": inp["pred_keypoints_3d"]}
out = m(inp["images"].to("cuda:0"), None, batch)
torch.cuda.synchronize(); print("forward ok", out[0][0, :2])
out[0].sum().backward(); torch.cuda.synchronize(); print("backward ok")
