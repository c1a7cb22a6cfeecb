#!/usr/bin/env python
"""Per-layer timing of the bf16 weight-gradient entry points (GPU box): python tools/wgrad_bench.py [--batch 8] [--only "l3"]
For the weight-gradient problems of the config-2 training step (4 views per sample): lt_conv_wgrad_bf16_nhwc (operands straight from the
channels-last tensors) against lt_pack_n8_from_bf16 x 2 + lt_conv_wgrad_bf16 (image-octet operands), each as the step launches it
(kernel + ordered reduce), events on one stream, best of --rounds rounds of --reps back-to-back launches.  Prints us per layer, the GEMM's
TFLOP/s and the bytes of operands per second."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learnable-triangulation-pytorch_amd"))
import torch

import lt_engine as E
import lt_hip as H


def layers(B, views=4):
    n = B * views
    L = []          # name, images, (D, H, W), Cin, Cout, taps per dim, stride, pad
    L.append(("stem 7x7/2 8->64 @384", n, (1, 384, 384), 8, 64, (1, 7, 7), 2, 3))
    for name, hw, cin, mid in (("l1", 96, 256, 64), ("l2", 48, 512, 128), ("l3", 24, 1024, 256), ("l4", 12, 2048, 512)):
        L.append(("%s reduce 1x1 %d->%d @%d" % (name, cin, mid, hw), n, (1, hw, hw), cin, mid, (1, 1, 1), 1, 0))
        L.append(("%s 3x3 %d->%d @%d" % (name, mid, mid, hw), n, (1, hw, hw), mid, mid, (1, 3, 3), 1, 1))
        L.append(("%s expand 1x1 %d->%d @%d" % (name, mid, cin, hw), n, (1, hw, hw), mid, cin, (1, 1, 1), 1, 0))
    L.append(("v2v 3^3 32->32 @64", B, (64, 64, 64), 32, 32, (3, 3, 3), 1, 1))
    L.append(("v2v 3^3 64->64 @32", B, (32, 32, 32), 64, 64, (3, 3, 3), 1, 1))
    L.append(("v2v 3^3 128->128 @16", B, (16, 16, 16), 128, 128, (3, 3, 3), 1, 1))
    L.append(("v2v 1^3 32->32 @64", B, (64, 64, 64), 32, 32, (1, 1, 1), 1, 0))
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    lib = H.lib()
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        best = None
        for _ in range(args.rounds):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record(); e1.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / args.reps
            best = t if best is None else min(best, t)
        return best

    print("%-30s %9s %9s %9s | %8s %8s" % ("layer", "nhwc us", "packed us", "packs us", "TFLOP/s", "GB/s ops"))
    for name, N, (D, Hh, W), Cin, Cout, ks, s, p in layers(args.batch):
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        pd = tuple(p if k > 1 else 0 for k in ks)
        st3 = tuple(s if k > 1 else 1 for k in ks)
        Do, Ho, Wo = [(n + 2 * q - k) // t + 1 for n, q, k, t in zip((D, Hh, W), pd, ks, st3)]
        x = torch.randn(N, D, Hh, W, Cin, device=dev).bfloat16()
        dy = torch.randn(N, Do, Ho, Wo, Cout, device=dev).bfloat16()
        taps = torch.tensor([(a, b, c, 0) for a in range(ks[0]) for b in range(ks[1]) for c in range(ks[2])], dtype=torch.int32, device=dev)
        ntaps = taps.shape[0]
        cop, kp = E.cout_pad_of(Cout), ntaps * Cin
        G = (N + 7) // 8
        ws = torch.empty(max(int(lib.lt_conv_wgrad_bf16_workspace(G * Do * Ho * Wo, cop, kp)), 16), dtype=torch.uint8, device=dev)
        dw = torch.empty(cop, kp, device=dev)
        pa = torch.empty(int(lib.lt_pack_n8_bf16_bytes(N, Do * Ho * Wo, Cout)), dtype=torch.uint8, device=dev)
        pb = torch.empty(int(lib.lt_pack_n8_bf16_bytes(N, D * Hh * W, Cin)), dtype=torch.uint8, device=dev)
        s3, p3 = H.i3(st3), H.i3(pd)

        def packs():
            H.check(lib.lt_pack_n8_from_bf16(dy.data_ptr(), pa.data_ptr(), N, Do * Ho * Wo, Cout, Cout, st), "pack dy")
            H.check(lib.lt_pack_n8_from_bf16(x.data_ptr(), pb.data_ptr(), N, D * Hh * W, Cin, Cin, st), "pack x")

        def packed():
            packs()
            H.check(lib.lt_conv_wgrad_bf16(pa.data_ptr(), pb.data_ptr(), taps.data_ptr(), dw.data_ptr(), N, D, Hh, W, Cin, Do, Ho, Wo, s3, p3, Cout, Cout, cop, kp, ntaps, 0,
                                           ws.data_ptr(), st), "lt_conv_wgrad_bf16")

        def nhwc():
            H.check(lib.lt_conv_wgrad_bf16_nhwc(dy.data_ptr(), x.data_ptr(), taps.data_ptr(), dw.data_ptr(), N, D, Hh, W, Cin, Cin, Do, Ho, Wo, s3, p3, Cout, Cout, cop, kp,
                                                ntaps, 0, ws.data_ptr(), st), "lt_conv_wgrad_bf16_nhwc")

        covered = lib.lt_conv_wgrad_bf16_nhwc_ok(N, D, Hh, W, Cin, Cin, Do, Ho, Wo, s3, p3, Cout, Cout, cop, kp, ntaps)
        t_pk, t_p = timed(packed), timed(packs)
        t_n = timed(nhwc) if covered else float("nan")
        flop = 2.0 * N * Do * Ho * Wo * Cout * kp
        byt = 2.0 * (x.numel() + dy.numel())
        t = t_n if covered else t_pk
        print("%-30s %9.1f %9.1f %9.1f | %8.1f %8.0f" % (name, t_n, t_pk, t_p, flop / t / 1e6, byt / t / 1e3))


if __name__ == "__main__":
    main()
