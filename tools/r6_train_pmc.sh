#!/bin/bash
# per-kernel HBM traffic / MFMA-busy of the act16 training step (3 steps per pass: the first records the tape): tools/pmc_summary.py over four rocprofv3 --pmc passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
PB="--train --train-dtype act16 --batch 8 --steps 1 --warmup 2 --no-pmc-leg"
rm -rf $OUT/pmc_fetch_tr $OUT/pmc_write_tr $OUT/pmc_mfma_tr
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_tr -o bench -- python $R/bench.py $PB > /dev/null 2> $OUT/tr_pmc_fetch.err; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_tr -o bench -- python $R/bench.py $PB > /dev/null 2> $OUT/tr_pmc_write.err; echo "pmc write rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_tr -o bench -- python $R/bench.py $PB > /dev/null 2> $OUT/tr_pmc_mfma.err; echo "pmc mfma rc=$?"
cd $R
python - <<'PY'
import sys, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import pmc_summary, bench
res = pmc_summary.summarise("gpurun_out", 3, 8, "act16", 4, 64, "_tr", " --train --train-dtype act16 --batch 8 (3 steps per pass, the first one records the tape)")
fam = {}
for k, v in res["per_kernel"].items():
    f = bench.train_family(k)
    e = fam.setdefault(f, [0.0, 0.0, 0.0])
    e[0] += v.get("fetch_bytes_per_step_corrected", 0.0); e[1] += v.get("write_bytes_per_step", 0.0); e[2] += v.get("shader_cycles_per_step", 0.0)
res["by_family_gb_per_step"] = {f: {"fetch": round(e[0] / 1e9, 2), "write": round(e[1] / 1e9, 2), "kernel_cycles": round(e[2])} for f, e in sorted(fam.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))}
json.dump(res, open("gpurun_out/r06_train_hbm_traffic_pmc.json", "w"), indent=1)
print(json.dumps(res["by_family_gb_per_step"], indent=1))
rows = sorted(res["per_kernel"].items(), key=lambda kv: -(kv[1].get("fetch_bytes_per_step_corrected", 0) + kv[1].get("write_bytes_per_step", 0)))
for k, v in rows[:14]:
    print("%-60s fetch %6.2f write %6.2f GB/step  launches %6.1f  mfma %.3f" % (k[:60], v.get("fetch_bytes_per_step_corrected", 0) / 1e9, v.get("write_bytes_per_step", 0) / 1e9, v.get("launches_per_step", 0), v.get("mfma_busy_frac", 0)))
PY
rm -rf $OUT/pmc_fetch_tr $OUT/pmc_write_tr $OUT/pmc_mfma_tr
