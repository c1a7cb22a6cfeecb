#!/usr/bin/env python
"""Headline benchmark: multi-view samples/sec of the volumetric-softmax forward path
(BASELINE.json config 2: 4 views 384x384, 64^3 voxel cube, ResNet-152 backbone) on N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" = one forward of B samples per GPU (B x 4 views -> B skeletons) on synthetic inputs already
resident in HBM, through the reference-shaped module API (VolumetricTriangulationNet.forward) whose
arithmetic is liblt_hip.so.  One process per GPU; samples are independent units, so ranks shard the batch
with NO data-path collective (weak scaling); the only communication is the barrier + MAX-reduce of the
timing.  Rank 0 prints ONE JSON line.

Besides the driver contract the line carries
  roofline      the dominant kernel family (implicit-GEMM convolutions on MFMA): algorithmic FLOP per step
                (2*MAC of every conv launch) / summed launch durations, measured with hipEvent pairs around
                every launch on the launch stream (eager pass after the timed region);
  roofline_hbm  the same for the two HBM-bound kernels (unprojection gather, 3D soft-argmax), algorithmic
                bytes from SURVEY.md section 8(d);
  cpu_baseline  the CPU oracle (torch-CPU restatement of the reference path, same weights) timed on this
                box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "learnable-triangulation-pytorch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def vol_config(num_layers, volume_size, dtype):
    """experiments/human36m/train/human36m_vol_softmax.yaml:26-53 of the reference, as a ConfigDict."""
    from mvn.utils.cfg import ConfigDict
    return ConfigDict({"model": {
        "name": "vol", "kind": "mpii", "volume_aggregation_method": "softmax", "init_weights": False, "checkpoint": "",
        "use_gt_pelvis": False, "cuboid_side": 2500.0, "volume_size": volume_size, "volume_multiplier": 1.0,
        "volume_softmax": True, "heatmap_softmax": True, "heatmap_multiplier": 100.0, "compute_dtype": dtype,
        "backbone": {"name": "resnet%d" % num_layers, "style": "simple", "init_weights": False, "checkpoint": "",
                     "num_joints": 17, "num_layers": num_layers}}})


def synthetic_batch(B, NV, image, seed):
    """randn images, NV ring cameras (r=4000 mm, h=1000 mm, f=1.2*H) looking at the origin,
    pelvis ~ N(0, 100 mm)."""
    from mvn.utils.multiview import Camera
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, NV, 3, image, image, generator=g)
    cams, Ks, Rs, ts = [], [], [], []
    for v in range(NV):
        phi = 2.0 * np.pi * v / NV
        Cc = np.array([4000.0 * np.cos(phi), 4000.0 * np.sin(phi), 1000.0])
        fwd = -Cc / np.linalg.norm(Cc)
        right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
        R = np.stack([right, np.cross(fwd, right), fwd])
        t = (-R @ Cc).reshape(3, 1)
        K = np.array([[1.2 * image, 0, image / 2.0], [0, 1.2 * image, image / 2.0], [0, 0, 1.0]])
        cams.append([Camera(R, t, K) for _ in range(B)])
        Ks.append(K); Rs.append(R); ts.append(t)
    kp = np.random.RandomState(seed).randn(B, 17, 3) * 100.0
    return images, {"cameras": cams, "pred_keypoints_3d": kp}, (np.stack(Ks), np.stack(Rs), np.stack(ts))


def cpu_baseline(state_dict, args, budget_s=20.0):
    """The oracle (oracle/vol_oracle.py: the reference's forward restated on torch-CPU fp32, same ATen ops) on the
    host cores of this box, same weights, B=1 samples of the same workload: 1 warm-up + as many timed forwards as
    fit ~budget_s (at least 2)."""
    from oracle import vol_oracle
    from oracle.synth import AttrDict
    cfg = AttrDict(vol_config(args.layers, args.volume, "fp32"))
    images, batch, (K, R, t) = synthetic_batch(1, args.views, args.image, 123)
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, args.cpu_threads))   # oneDNN on >64 threads thrashes on these small layers (measured: 256 threads -> 170 s/forward)
    torch.set_num_threads(cores)
    run = lambda: vol_oracle.volumetric_forward(sd, cfg, images, K, R, t, batch["pred_keypoints_3d"])
    run()
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 < budget_s and n < 16):
        run(); n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d forwards of B=1 (%d views %dx%d, %d^3 voxels, ResNet-%d, fp32), %.1f s" % (
                n, args.views, args.image, args.image, args.volume, args.layers, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--image", type=int, default=384)
    ap.add_argument("--volume", type=int, default=64)
    ap.add_argument("--layers", type=int, default=152)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--ops-json", default="", help="write the per-launch timing table here")
    ap.add_argument("--cpu-threads", type=int, default=32, help="host threads for the CPU baseline leg")
    ap.add_argument("--tile", type=int, default=0, help="force a conv tile id (LT_TILE_*), 0 = auto")
    args = ap.parse_args()

    import lt_dist
    world, rank, local = lt_dist.init("nccl")   # "nccl" on PyTorch-ROCm is RCCL (xGMI inside the node)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from mvn.models.triangulation import VolumetricTriangulationNet
    torch.manual_seed(0)
    model = VolumetricTriangulationNet(vol_config(args.layers, args.volume, args.dtype), device=dev)
    # random-init weights of the architecture; BatchNorm statistics randomised so that folding is not a no-op
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, buf in model.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
            elif name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
    model.eval()
    model.use_graph = not args.no_graph
    model.tile_override = args.tile
    model.copy_outputs = True
    B = args.batch
    images, batch, _ = synthetic_batch(B, args.views, args.image, 1000 + rank)   # rank r owns its own shard of samples
    images = images.to(dev)

    barrier = lt_dist.barrier

    def step():
        return model(images, None, batch)

    # setup, not a step of the contract: the first call records the plan and captures the hipGraph, the second is the graph's
    # first replay (upload); the W warm-up steps and the K timed steps below are all plain replays
    for _ in range(2):
        out = step()
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    value, total_samples, dt = lt_dist.job_throughput(B * args.steps, dt_local, dev)   # all ranks' samples / slowest rank
    assert torch.isfinite(out[0]).all()

    result = None
    if rank == 0:
        result = {
            "metric": "multi-view samples/sec (4-view vol-softmax forward)", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE config 2: volumetric-softmax forward, %d views %dx%d, %d^3 voxel cube, ResNet-%d, "
                                   "random-init weights" % (args.views, args.image, args.image, args.volume, args.layers),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": "batch-sharded replicas x%d, no data-path collective" % world,
                       "hip_graph": not args.no_graph},
        }
        if not args.no_profile:
            plan = [p for k, p in model._plans.items()][0]["plan"]
            st = model._side_stream(dev).cuda_stream
            with torch.cuda.stream(model._side_stream(dev)):
                plan.run_profiled(st, reps=1)
                ops = plan.run_profiled(st, reps=3)
            fam = {}
            for o in ops:
                f = fam.setdefault(o["kind"], {"ms": 0.0, "flops": 0, "bytes": 0, "launches": 0})
                f["ms"] += o["ms"]; f["flops"] += o["flops"]; f["bytes"] += o["bytes"]; f["launches"] += 1
            conv = fam["conv"]
            peak = PEAK_TFLOPS[args.dtype]
            ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
            # HBM traffic of the conv family per step from the committed PMC passes (tools/pmc_summary.py; same batch only)
            traffic = None
            try:
                pm = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_hbm_traffic_pmc.json")))
                if pm.get("per_gpu_batch") == B and args.dtype == "bf16":
                    traffic = pm["conv_family_bytes_per_step"]
            except (OSError, ValueError, KeyError):
                pass
            for extra in ("pwchain", "stem"):   # the fused pointwise tail and the fused stem are convolution work too
                if extra in fam:
                    conv = {k: conv[k] + fam[extra][k] for k in conv}
            ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
            result["roofline"] = {"kernel": "conv family: conv_igemm2/3/5, conv3d_halo*, conv_pw, stem_pool, pwchain (all %d launches of one step)" % conv["launches"], "bound": "mfma",
                                  "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                                  "flop_per_step": conv["flops"], "ms_per_step_in_kernel": conv["ms"]}
            hb = {}
            for k in ("unproject", "softargmax3d"):
                if k in fam:
                    a = fam[k]["bytes"] / (fam[k]["ms"] * 1e-3) / 1e9
                    hb[k] = {"bound": "hbm", "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a / PEAK_HBM_GBS,
                             "bytes_per_step": fam[k]["bytes"], "ms_per_step_in_kernel": fam[k]["ms"], "traffic": None}
            result["roofline_hbm"] = hb
            result["kernel_time_ms_per_step"] = {k: round(v["ms"], 4) for k, v in fam.items()}
            if args.ops_json:
                os.makedirs(os.path.dirname(os.path.abspath(args.ops_json)), exist_ok=True)
                json.dump(ops, open(args.ops_json, "w"), indent=0)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(model.state_dict(), args)
    barrier()
    if rank == 0:
        print(json.dumps(result))
    lt_dist.shutdown()


if __name__ == "__main__":
    main()
