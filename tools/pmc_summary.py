#!/usr/bin/env python
"""Turns the two rocprofv3 PMC passes of tools/gpu_session.sh pmc (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of
`bench.py --steps 3 --warmup 1 --no-graph`) into HBM bytes per step and per kernel.

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE on gfx950 counts 64 B per 128-byte request for wide coalesced
reads -> doubled; both counters are in KiB per dispatch; WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(pattern, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:70]
            per[k][0] += float(r["Counter_Value"]); per[k][1] += 1
    return per


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6       # forwards in the run: bench.py --steps 3 --warmup 1 + its 2 setup calls
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    fetch = load(os.path.join(out_dir, "pmc_fetch", "*counter_collection.csv"), "FETCH_SIZE")
    write = load(os.path.join(out_dir, "pmc_write", "*counter_collection.csv"), "WRITE_SIZE")
    per = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0])[0] * 2 + write.get(k, [0])[0])):
        f, nf = fetch.get(k, [0.0, 0]); w, _ = write.get(k, [0.0, 0])
        per[k] = {"launches_per_step": nf / steps, "fetch_bytes_per_step_corrected": f * 1024 * 2 / steps, "write_bytes_per_step": w * 1024 / steps}
    conv = [v for k, v in per.items() if k.startswith("conv") or k.startswith("pwchain") or k.startswith("stem_pool")]
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 1 --no-graph; FETCH_SIZE x2 "
                   "(gfx950 counts 64 B per 128-B request), KiB -> bytes; Infinity-Cache hits are included in both counters",
           "per_gpu_batch": batch,
           "total_fetch_bytes_per_step": sum(v["fetch_bytes_per_step_corrected"] for v in per.values()),
           "total_write_bytes_per_step": sum(v["write_bytes_per_step"] for v in per.values()),
           "conv_family_bytes_per_step": sum(v["fetch_bytes_per_step_corrected"] + v["write_bytes_per_step"] for v in conv),
           "per_kernel": per}
    dst = os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")
    json.dump(res, open(dst, "w"), indent=1)
    print("wrote", dst, "total fetch %.2f GB write %.2f GB per step; conv family %.2f GB" %
          (res["total_fetch_bytes_per_step"] / 1e9, res["total_write_bytes_per_step"] / 1e9, res["conv_family_bytes_per_step"] / 1e9))


if __name__ == "__main__":
    main()
