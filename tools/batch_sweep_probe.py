import os, sys, time, json
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/learnable-triangulation-pytorch_amd"]
import torch, bench
from mvn.models.triangulation import VolumetricTriangulationNet
dev=torch.device("cuda:0")
torch.manual_seed(0)
m=VolumetricTriangulationNet(bench.vol_config(152,64,"bf16"),device=dev); m.to(dev).eval(); m.copy_outputs=True
res={}
for B in (1,2,3,5,8,10,16):
    images,batch,_=bench.synthetic_batch(B,4,384,1000); images=images.to(dev)
    for _ in range(4): m(images,None,batch)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): m(images,None,batch)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    res[B]=round(B/dt,1)
print(os.environ.get("LT_NO_XR","xr"), res)
