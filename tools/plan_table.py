#!/usr/bin/env python
"""The CURRENT state of the bf16 forward as one markdown table (VERDICT r5 "next" 8): which kernel runs which group of layers, how long it
takes and at what fraction of its roof, at 64 / 8 / 1 samples per step -- generated from ``bench.py --batch B --ops-json`` files (an event pair around
every launch of the recorded plan):

    python tools/plan_table.py profiles/r06_bench_ops_bf16_b64.json profiles/r06_bench_ops_bf16_b8.json profiles/r06_bench_ops_bf16_b1.json

Roofs: 2.5 PFLOP/s dense bf16 MFMA for the convolutions (2 x MAC / time), 8 TB/s HBM for the streams (algorithmic bytes / time)."""
import collections
import json
import re
import sys

PEAK_TF, PEAK_GBS = 2500.0, 8000.0

# (regex on the op label, group name, kernel(s), reference lines)
GROUPS = [
    (r"^stem ", "stem: conv1 + bn1 + relu + maxpool", "`stem_pool_kernel`", "pose_resnet.py:293-297"),
    (r"^conv7x7 ", "stem (fp32 / non-fused): conv1", "`conv_igemm2`", "pose_resnet.py:293-296"),
    (r"^bneck-ds", "layer1 block 0 (downsample branch inside)", "`bneck_ds_kernel`", "pose_resnet.py:75-95, 196-206"),
    (r"^bneck 256", "layer1 identity blocks (2)", "`bneck_kernel<256, 64>`", "pose_resnet.py:75-95"),
    (r"^bneck 512", "layer2 identity blocks (7)", "`bneck_kernel<512, 128>`", "pose_resnet.py:75-95"),
    (r"downsample\)$", "layer2-4 block 0: expand + strided downsample", "`conv_igemm7<3>` (`lt_conv_cat2_fwd`)", "pose_resnet.py:75-95"),
    (r"^xr ", "layer3 seams: expand(i) + reduce(i + 1) (34)", "`xr_kernel<1024, 256, FULL, 3 / 2 / 1>`", "pose_resnet.py:75-95"),
    (r"^conv3x3 256->256 @\d+x1x24x24", "layer3 3x3 256 -> 256 (36)", "`conv2d_halo_kernel<8 / 4, 9, 1>` from 5 samples; `conv_igemm2` below; the strided one of block 0: `conv_igemm7`", "pose_resnet.py:84-86"),
    (r"^conv1x1 (256->1024|1024->256) @\d+x1x24x24", "layer3 1x1 outside the seams", "`conv_igemm6 / 7`; `conv_igemm2` at 1 sample (no seam fusion below 2 samples)", "pose_resnet.py:80-92"),
    (r"^deconv4x4 256->256", "head: 4x4 / 2 transposed 256 -> 256 (2)", "`conv2d_halo_kernel<8, 4, 4>` from 60 tiles; `conv_igemm7 / 2` below", "pose_resnet.py:208-233"),
    (r"^deconv4x4 2048->256", "head: 4x4 / 2 transposed 2048 -> 256", "`conv_igemm7` (four parity phases)", "pose_resnet.py:208-233"),
    (r"^conv(1x1|3x3) \d+(\+\d+)?->\d+ @\d+x1x", "other backbone convolutions (layer2-4 block 0, layer4, process_features)", "`conv_igemm2 / 3 / 6 / 7` by tile count", "pose_resnet.py:75-95; triangulation.py:238-240"),
    (r"^unproject", "unprojection + view softmax + coordinate grid", "`unproject_qn_kernel<1, GRID>`", "op.py:99-166; triangulation.py:298-339"),
    (r"^conv7x7x7", "V2V front 7^3 32 -> 16", "`conv3d_halo7b_kernel`", "v2v.py:146"),
    (r"^conv3x3x3 32->32 .*skip", "V2V 16 -> 32 block: second 3^3 with the skip convolution inside", "`conv3d_halo_col_kernel<.., SKIP>`", "v2v.py:20-42"),
    (r"^conv3x3x3 32->32", "V2V 3^3 32 -> 32 at 64^3 (8)", "`conv3d_halo_col_kernel` (>= 256 columns) / `conv3d_halo_persist_kernel`", "v2v.py:20-42"),
    (r"^conv3x3x3 (16->32|64->64|32->64) ", "V2V 3^3 16 -> 32, 32 -> 64, 64 -> 64 (64^3 / 32^3)", "`conv3d_halo_wreg_kernel`", "v2v.py:20-42"),
    (r"^conv3x3x3 128->128 @\d+x16x16x16$", "V2V 3^3 128 -> 128 at 16^3 (5)", "`conv3d_halo_wreg_kernel<128, 128>`", "v2v.py:20-42"),
    (r"^conv3x3x3 64->128", "V2V 3^3 64 -> 128 at 16^3", "`conv3d_halo_kernel` (loader waves)", "v2v.py:20-42"),
    (r"split-K", "V2V 3^3 128 -> 128 at 8^3 / 4^3 / 2^3 (18, split-K + reduce)", "`conv_igemm2` tap-group phases + `splitk_reduce_kernel`", "v2v.py:78-90"),
    (r"^conv3x3x3 128->128", "V2V 3^3 128 -> 128 (other)", "`conv3d_halo_wreg_kernel` / `conv_igemm2`", "v2v.py:20-42"),
    (r"^(conv1x1x1|deconv2x2x2)", "V2V single-tap layers (1^3 skips, 2^3 / 2 deconvolutions + skip add)", "`conv_pw_kernel`", "v2v.py:23-37, 54-66"),
    (r"^maxpool3d", "V2V max pools (5)", "`maxpool_kernel`", "v2v.py:45-51"),
    (r"^pwchain", "V2V pointwise tail 32 -> 32 -> 32 -> 17, planar fp32 logits", "`pwchain_kernel`", "v2v.py:156-160"),
    (r"^softargmax3d", "3D soft-argmax (tail op: writes the returned tensors)", "`sa3_partial_planar` + `sa3_probs_planar`", "op.py:84-96"),
    (r"^features_out", "returned features (B, NV, 32, h, w) fp32", "`nhwc_to_nchw_kernel`", "triangulation.py:346, 355"),
]


def load(path):
    agg = collections.OrderedDict()
    ops = json.load(open(path))
    for o in ops:
        for i, (rx, name, kern, ref) in enumerate(GROUPS):
            if re.search(rx, o["label"]):
                break
        else:
            i = len(GROUPS)
        a = agg.setdefault(i, [0, 0.0, 0, 0])
        a[0] += 1; a[1] += o["ms"]; a[2] += o["flops"]; a[3] += o["bytes"]
    return agg, sum(o["ms"] for o in ops), len(ops)


def cell(a):
    if a is None:
        return "-"
    n, ms, f, b = a
    if ms <= 0 or (not f and b / ms / 1e6 > PEAK_GBS):          # (the features tail op is skipped by the profiled run: no output tensor is handed in)
        return "%d: not timed" % n
    frac = (f / ms / 1e9 / PEAK_TF) if f else (b / ms / 1e6 / PEAK_GBS)
    return "%d: %.0f us, %.3f%s" % (n, 1e3 * ms, frac, "" if f else " hbm")


def main():
    paths = sys.argv[1:]
    data = [load(p) for p in paths]
    batches = [re.search(r"_b(\d+)\.json", p).group(1) for p in paths]
    print("| layers | kernel | reference | " + " | ".join("%s samples: launches, time, fraction of the roof" % b for b in batches) + " |")
    print("|---|---|---|" + "---|" * len(paths))
    idx = sorted(set(i for agg, _, _ in data for i in agg), key=lambda i: -(data[0][0].get(i, [0, 0.0])[1]))
    for i in idx:
        name, kern, ref = (GROUPS[i][1], GROUPS[i][2], GROUPS[i][3]) if i < len(GROUPS) else ("other", "", "")
        print("| %s | %s | `%s` | " % (name, kern, ref) + " | ".join(cell(agg.get(i)) for agg, _, _ in data) + " |")
    print("| **all launches of a forward** | | | " + " | ".join("%d: %.2f ms" % (n, ms) for _, ms, n in data) + " |")


if __name__ == "__main__":
    main()
