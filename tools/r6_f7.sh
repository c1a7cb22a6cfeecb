#!/bin/bash
# round 6: the fp32 7^3 halo kernels (two channel phases): kernel tests, the fp32 goldens, A/B of the fp32 forward and the fp32 training step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "halo" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan_abi.py -m gpu -q --tb=short -p no:cacheprovider -x -k "fp32 or golden or f32" 2>&1 | tail -5
for i in 1 2; do
  timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-pmc-leg --steps 5 --warmup 2 --full-line --ops-json $OUT/f7_ops_new.json > $OUT/f7_new.json 2> $OUT/f7_new.err
  LT_HALO_NO_F7=1 timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-pmc-leg --steps 5 --warmup 2 --full-line --ops-json $OUT/f7_ops_old.json > $OUT/f7_old.json 2> $OUT/f7_old.err
  python - <<'PY'
import json
for t in ("new", "old"):
    d = json.load(open("gpurun_out/f7_%s.json" % t)); ops = json.load(open("gpurun_out/f7_ops_%s.json" % t))
    o7 = [o for o in ops if o["label"].startswith("conv7x7x7")]
    print(t, "fp32 forward %.1f samples/s, roofline %.3f, parity %s | 7^3: %s" % (d["value"], d["roofline"]["frac"], json.dumps(d.get("parity", {}).get("joints_max_rel_vs_exact")),
          ", ".join("%.2f ms %.0f TFLOP/s" % (o["ms"], o["flops"] / o["ms"] / 1e9) for o in o7)))
PY
done
timeout 900 python bench.py --train --batch 4 --steps 4 --warmup 2 --no-pmc-leg > $OUT/f7_train_new.json 2> $OUT/f7_train_new.err
LT_HALO_NO_F7=1 timeout 900 python bench.py --train --batch 4 --steps 4 --warmup 2 --no-pmc-leg > $OUT/f7_train_old.json 2> $OUT/f7_train_old.err
python - <<'PY'
import json
for t in ("new", "old"):
    d = json.load(open("gpurun_out/f7_train_%s.json" % t))
    print(t, "fp32 training step %.2f samples/s, %.1f ms, losses %s" % (d["value"], d["ms_per_step"], d["losses"][:4]))
PY
