#!/usr/bin/env python
"""Turns the rocprofv3 PMC passes of `tools/gpu_session.sh pmc pmcmfma` into per-kernel HBM bytes per step and MFMA-busy
fractions, and writes profiles/rNN_hbm_traffic_pmc.json (the file bench.py reads `roofline.traffic` from).

    python tools/pmc_summary.py [gpurun_out] [forwards_in_run] [batch] [--round 2] [--dtype bf16] [--views 4] [--volume 64]

Passes (separate runs of `bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-profile --no-extras --preroll-s 0`,
i.e. 6 forwards: 2 setup + 1 warm-up + 3 timed):
  pmc_fetch/  --pmc FETCH_SIZE          pmc_write/  --pmc WRITE_SIZE
  pmc_mfma/   --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE on gfx950 counts 64 B per 128-byte request for wide coalesced
reads -> doubled; both counters are in KiB per dispatch; WRITE_SIZE is taken as is (uncalibrated).  MFMA busy fraction of a
kernel = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the MFMA counter is summed over every SIMD of the chip
(32 cycles per 32x32x16 bf16 MFMA, microarch guide) and GRBM_GUI_ACTIVE over the 8 XCDs; calibrated on the 7^3 convolution, whose MFMA
count is known exactly: 2.947 TFLOP / (1024 flop/cycle/SIMD x 1024 SIMDs) = 2.81 M busy cycles per SIMD against 5.64 M cycles of
kernel time = 0.50, the counters give 0.506."""
import argparse
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 256 * 4
VALU_CYCLES_PER_WAVE_INST = 2          # a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md)
N_XCD = 8        # GRBM_GUI_ACTIVE of a dispatch is summed over the 8 XCDs (45.1 M 'cycles' for a 2.56 ms kernel = 8 x 2.2 GHz)


def kernel_key(name):
    """'void (anonymous namespace)::conv_igemm6_kernel<1, 2>(Args)' and '(anonymous namespace)::unproject_q4_kernel(Args) [clone .kd]'
    -> 'conv_igemm6_kernel<1, 2>' / 'unproject_q4_kernel' (round 1 split on the first '(' and lost every kernel whose name has
    no 'void' prefix in front of '(anonymous namespace)')."""
    n = name.strip()
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in n:              # cut at the first '(' that is not inside a template argument list
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()[:90]


def family(k):
    if k.startswith(("conv_pack", "stem_pack")):
        return "setup"          # one-time weight packing at plan build, not part of a step
    if k.startswith(("conv", "pwchain", "stem_pool", "bneck", "xr_kernel")):   # (xr_kernel was missing from this list when round 5's first record was taken:
        return "conv"                                                        #  its 27 GB per step were not in that record's `traffic`)
    if k.startswith("unproject"):
        return "unproject"
    if k.startswith(("sa3_", "softargmax3d")):
        return "softargmax3d"
    if k.startswith("coord_volumes"):
        return "coord_volumes"
    return "other"


def load(pattern, counters):
    per = {c: collections.defaultdict(lambda: [0.0, 0]) for c in counters}
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c in per:
                k = kernel_key(r["Kernel_Name"])
                per[c][k][0] += float(r["Counter_Value"]); per[c][k][1] += 1
    return per


def summarise(out_dir, steps, batch=32, dtype="bf16", views=4, volume=64, suffix="", note_extra=""):
    """The three passes under ``out_dir`` (pmc_fetch*/ pmc_write*/ pmc_mfma*/; a missing pass leaves its fields empty) -> the summary dict
    (``steps`` = forwards in each profiled run).  Used by main() for the committed file and by bench.py for its in-run PMC leg."""
    fetch = load(os.path.join(out_dir, "pmc_fetch" + suffix, "*counter_collection.csv"), ["FETCH_SIZE"])["FETCH_SIZE"]
    write = load(os.path.join(out_dir, "pmc_write" + suffix, "*counter_collection.csv"), ["WRITE_SIZE"])["WRITE_SIZE"]
    sq_names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
    sq = load(os.path.join(out_dir, "pmc_mfma" + suffix, "*counter_collection.csv"), sq_names)
    per = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0])[0] * 2 + write.get(k, [0])[0])):
        f, nf = fetch.get(k, [0.0, 0]); w, nw = write.get(k, [0.0, 0])
        per[k] = {"family": family(k), "launches_per_step": max(nf, nw) / steps, "fetch_bytes_per_step_corrected": f * 1024 * 2 / steps,
                  "write_bytes_per_step": w * 1024 / steps}
    for k in sq["SQ_VALU_MFMA_BUSY_CYCLES"]:
        mf = sq["SQ_VALU_MFMA_BUSY_CYCLES"][k][0]
        gui = sq["GRBM_GUI_ACTIVE"].get(k, [0.0, 0])[0]
        wc = sq["SQ_WAVE_CYCLES"].get(k, [0.0, 0])[0]
        e = per.setdefault(k, {"family": family(k)})
        if gui > 0:
            e["mfma_busy_frac"] = mf / (N_SIMD * gui / N_XCD)
            e["shader_cycles_per_step"] = gui / N_XCD / steps
        if wc > 0:
            e["wave_wait_any_frac"] = sq["SQ_WAIT_ANY"].get(k, [0.0])[0] / wc
            e["wave_issue_stall_frac"] = sq["SQ_WAIT_INST_ANY"].get(k, [0.0])[0] / wc
            e["wave_active_frac"] = sq["SQ_ACTIVE_INST_ANY"].get(k, [0.0])[0] / wc
    # optional fourth pass (pmc_valu/): SQ_INSTS_VALU counts wave-level VALU instructions over the whole chip; a wave64 instruction occupies its
    # SIMD-32 for 2 cycles (MI355X_MICROARCH.md "Wave scheduling"), so valu_issue_frac = 2 x SQ_INSTS_VALU / (1024 SIMDs x kernel cycles) -- the roof a
    # VALU-bound kernel (the unprojection gather) is priced against
    vl = load(os.path.join(out_dir, "pmc_valu" + suffix, "*counter_collection.csv"), ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])
    for k in vl["SQ_INSTS_VALU"]:
        iv, n = vl["SQ_INSTS_VALU"][k]
        gui = vl["GRBM_GUI_ACTIVE"].get(k, [0.0, 0])[0]
        e = per.setdefault(k, {"family": family(k)})
        e["valu_wave_insts_per_step"] = iv / steps
        if gui > 0:
            e["valu_issue_frac"] = VALU_CYCLES_PER_WAVE_INST * iv / (N_SIMD * gui / N_XCD)
    tot = lambda fam, key: sum(v.get(key, 0.0) for v in per.values() if v["family"] == fam)
    hbm = {}
    for fam in ("unproject", "softargmax3d", "coord_volumes"):
        b = tot(fam, "fetch_bytes_per_step_corrected") + tot(fam, "write_bytes_per_step")
        if b:
            hbm[fam] = b
    conv_mf = sum(sq["SQ_VALU_MFMA_BUSY_CYCLES"][k][0] for k in sq["SQ_VALU_MFMA_BUSY_CYCLES"] if family(k) == "conv")
    conv_gui = sum(sq["GRBM_GUI_ACTIVE"][k][0] for k in sq["GRBM_GUI_ACTIVE"] if family(k) == "conv")
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters (separate passes, --kernel-trace only) over bench.py "
                   "--no-graph" + note_extra + "; FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), KiB -> bytes; Infinity-Cache hits are included "
                   "in both counters; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
           "per_gpu_batch": batch, "dtype": dtype, "views": views, "volume": volume, "forwards_profiled": steps,
           "total_fetch_bytes_per_step": sum(v.get("fetch_bytes_per_step_corrected", 0.0) for v in per.values()),
           "total_write_bytes_per_step": sum(v.get("write_bytes_per_step", 0.0) for v in per.values()),
           "conv_family_bytes_per_step": (tot("conv", "fetch_bytes_per_step_corrected") + tot("conv", "write_bytes_per_step")) if (fetch and write) else None,
           "conv_family_mfma_busy_frac": (conv_mf / (N_SIMD * conv_gui / N_XCD)) if conv_gui else None,
           "hbm_kernels_bytes_per_step": hbm if (fetch and write) else {},
           "valu": {fam: {"wave_insts_per_step": sum(v.get("valu_wave_insts_per_step", 0.0) for v in per.values() if v["family"] == fam),
                          "issue_frac": (VALU_CYCLES_PER_WAVE_INST * sum(vl["SQ_INSTS_VALU"][k][0] for k in vl["SQ_INSTS_VALU"] if family(k) == fam) /
                                         (N_SIMD * sum(vl["GRBM_GUI_ACTIVE"][k][0] for k in vl["GRBM_GUI_ACTIVE"] if family(k) == fam) / N_XCD))
                          if any(family(k) == fam for k in vl["GRBM_GUI_ACTIVE"]) else None}
                    for fam in ("unproject", "softargmax3d", "conv") if any(family(k) == fam for k in vl["SQ_INSTS_VALU"])},
           "per_kernel": per}
    assert "" not in per, "a kernel name parsed to the empty string"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir", nargs="?", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("forwards", nargs="?", type=int, default=6)
    ap.add_argument("batch", nargs="?", type=int, default=32)
    ap.add_argument("--round", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--volume", type=int, default=64)
    ap.add_argument("--suffix", default="")
    a = ap.parse_args()
    res = summarise(a.out_dir, a.forwards, a.batch, a.dtype, a.views, a.volume, a.suffix, " --steps 3 --warmup 1")
    hbm = res["hbm_kernels_bytes_per_step"]
    name = "r%02d_hbm_traffic_pmc%s.json" % (a.round, a.suffix)
    dst = os.path.join(ROOT, "profiles", name)
    json.dump(res, open(dst, "w"), indent=1)
    print("wrote", dst, "total fetch %.2f GB write %.2f GB per step; conv family %.2f GB, MFMA busy %s; hbm kernels %s" %
          (res["total_fetch_bytes_per_step"] / 1e9, res["total_write_bytes_per_step"] / 1e9, (res["conv_family_bytes_per_step"] or 0) / 1e9,
           res["conv_family_mfma_busy_frac"], {k: round(v / 1e9, 3) for k, v in hbm.items()}))


if __name__ == "__main__":
    main()
