#!/bin/bash
# Round-3 GPU sessions (from the repo root on the GPU box): bash tools/gpu_r3.sh <stage> ...   -- everything under gpurun_out/r3/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3
mkdir -p $OUT
cd $R
export PYTHONDONTWRITEBYTECODE=1
for s in "$@"; do
case $s in
tests_fast)   # the tests this round touched
  timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_data.py tests/test_gpu_backward.py -m gpu -q --tb=short -x -p no:cacheprovider > $OUT/test_train.log 2>&1
  echo "train/data rc=$?"; tail -5 $OUT/test_train.log
  timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_models.log 2>&1
  echo "models rc=$?"; tail -5 $OUT/test_models.log
  ;;
tests_all)
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_all.log 2>&1
  echo "all rc=$?"; tail -8 $OUT/test_all.log
  ;;
bench)        # the driver's command
  timeout 1500 python bench.py --ops-json $OUT/bench_ops_bf16.json > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
  echo "bench rc=$?"; python - <<PY
import json
try:
    r = json.load(open("$OUT/bench_bf16.json"))
    print({k: r[k] for k in ("value", "ms_per_step")}, "keys:", sorted(r.keys()))
    for k in ("config4", "train", "train_mixed"):
        v = r.get(k, {})
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "error", "leg_wall_s", "stderr_tail")})
except Exception as e:
    print("bench parse failed", e)
PY
  tail -3 $OUT/bench_bf16.err
  ;;
bwd)          # the unprojection backward (gather) and the whole training step, both aggregation families
  timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q --tb=short -x -p no:cacheprovider > $OUT/test_bwd.log 2>&1
  echo "backward rc=$?"; tail -15 $OUT/test_bwd.log
  timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider -k "whole_training_step or two_ranks or api_semantics or ten_steps or op_level" > $OUT/test_train_sel.log 2>&1
  echo "train selection rc=$?"; tail -25 $OUT/test_train_sel.log
  ;;
trainprof)    # per-op table of the recorded training step (B = 8 fp32) + the step rate
  timeout 600 python tools/train_profile.py 8 > $OUT/train_profile_b8.log 2>&1; echo "trainprof rc=$?"
  grep -E "^fwd|^bwd|unproject|zero" $OUT/train_profile_b8.log | head -12
  cp gpurun_out/train_ops_b8.json $OUT/ 2>/dev/null
  timeout 600 python bench.py --train --steps 10 --warmup 3 --batch 8 > $OUT/bench_train_b8.json 2> $OUT/bench_train_b8.err
  echo "bench train rc=$?"; python -c "
import json; r=json.load(open('$OUT/bench_train_b8.json')); print(r['value'], r['ms_per_step'], r['losses'])"
  ;;
aten)         # which torch (at::native / elementwise) kernels still run inside a forward / a training step?
  cd /tmp; export TMPDIR=/tmp
  rm -rf $OUT/prof_fwd $OUT/prof_train
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -o fwd -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-profile --no-extras > $OUT/prof_fwd.json 2> $OUT/prof_fwd.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $R/bench.py --train --batch 4 --steps 10 --warmup 3 > $OUT/prof_train.json 2> $OUT/prof_train.err
  cd $R
  python - <<PY
import csv, glob, json
out = {}
for tag, steps in (("fwd", 20 + 2 + 2), ("train", 10 + 3)):
    f = glob.glob("$OUT/prof_%s/*kernel_stats.csv" % tag)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    aten = [(r["Name"][:110], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in rows if "at::" in r["Name"] or "elementwise" in r["Name"].lower() or "Memcpy" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    out[tag] = {"steps_in_run": steps, "total_kernel_ms": tot, "torch_kernels": [{"name": n, "calls": c, "calls_per_step": c / steps, "ms": ms} for n, c, ms in sorted(aten, key=lambda t: -t[1])]}
    print(tag, "total kernel ms %.1f; torch kernels:" % tot)
    for n, c, ms in sorted(aten, key=lambda t: -t[1])[:12]:
        print("   %6d calls (%.1f / step) %8.3f ms  %s" % (c, c / steps, ms, n))
json.dump(out, open("$OUT/aten_kernels.json", "w"), indent=1)
PY
  for t in fwd train; do f=$(ls $OUT/prof_$t/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$t.csv; done
  ;;
mixedprof)    # mixed-precision step at B = 8: per-op table on ONE stream (weight gradients visible), then the step rate
  LT_TRAIN_NO_OVERLAP=1 timeout 600 python tools/train_profile.py 8 bf16 > $OUT/train_profile_b8_bf16.log 2>&1; echo "mixedprof rc=$?"
  grep -E "^fwd|^bwd" $OUT/train_profile_b8_bf16.log | head -4
  grep -E " wgrad | pack " $OUT/train_profile_b8_bf16.log | head -24
  cp gpurun_out/train_ops_b8_bf16.json $OUT/ 2>/dev/null
  timeout 600 python bench.py --train --train-dtype bf16 --steps 10 --warmup 3 --batch 8 > $OUT/bench_train_bf16mma_b8.json 2> $OUT/bench_train_bf16mma_b8.err
  echo "bench mixed rc=$?"; python -c "
import json; r=json.load(open('$OUT/bench_train_bf16mma_b8.json')); print(r['value'], r['ms_per_step'], r['losses'])"
  ;;
wgradtests)
  timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -x -p no:cacheprovider -k "op_level or wgrad or layer" > $OUT/test_wgrad.log 2>&1
  echo "wgrad tests rc=$?"; tail -6 $OUT/test_wgrad.log
  ;;
mixedstats)   # rocprofv3 kernel totals of the mixed-precision step at B = 8 on ONE stream
  cd /tmp; export TMPDIR=/tmp
  rm -rf $OUT/prof_mixed
  LT_TRAIN_NO_OVERLAP=1 timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mixed -o mixed -- python $R/bench.py --train --train-dtype bf16 --batch 8 --steps 5 --warmup 2 > $OUT/prof_mixed.json 2> $OUT/prof_mixed.err < /dev/null
  echo "mixedstats rc=$?"
  cd $R
  f=$(ls $OUT/prof_mixed/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_mixed_b8.csv; head -36 "$f" | cut -c1-160; else tail -5 $OUT/prof_mixed.err; fi
  rm -rf $OUT/prof_mixed
  ;;
mixedpmc)     # MFMA-busy fraction of the kernels of the mixed-precision step (one PMC pass, single stream)
  cd /tmp; export TMPDIR=/tmp
  rm -rf $OUT/pmc_mixed
  LT_TRAIN_NO_OVERLAP=1 timeout 420 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mixed -o mixed -- python $R/bench.py --train --train-dtype bf16 --batch 8 --steps 2 --warmup 1 > $OUT/pmc_mixed.json 2> $OUT/pmc_mixed.err < /dev/null
  echo "mixedpmc rc=$?"
  cd $R
  python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "tools")
from pmc_summary import kernel_key
f = glob.glob("$OUT/pmc_mixed/*counter_collection.csv")
if not f:
    print("no counter csv"); raise SystemExit
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(f[0])):
    k = kernel_key(r["Kernel_Name"]); v = float(r["Counter_Value"])
    a = acc[k]
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": a[1] += v; a[0] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE": a[2] += v
rows = []
for k, (n, mf, ga) in acc.items():
    if ga > 0:
        rows.append({"kernel": k, "dispatches": n, "gui_active_cycles_per_xcd": ga / 8, "mfma_busy_frac": mf / (1024 * ga / 8)})
rows.sort(key=lambda r: -r["gui_active_cycles_per_xcd"])
json.dump({"what": "bench.py --train --train-dtype bf16 --batch 8 (3 steps incl. the recording one), one stream; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)", "kernels": rows[:40]}, open("$OUT/pmc_mixed_summary.json", "w"), indent=1)
for r in rows[:22]:
    print("%-62s n=%5d  cycles %.3g  mfma_busy %.3f" % (r["kernel"][:62], r["dispatches"], r["gui_active_cycles_per_xcd"], r["mfma_busy_frac"]))
PY
  rm -rf $OUT/pmc_mixed
  ;;
smoke)
  timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
  ;;
*) echo "unknown stage $s";;
esac
done
