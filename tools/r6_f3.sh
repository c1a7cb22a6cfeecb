#!/bin/bash
# round 6: exact-fp32 3^3 32 -> 32 in two channel phases (two workgroups per CU): kernel tests, fp32 goldens, A/B of the fp32 forward
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "halo or conv_all_tiles" 2>&1 | tail -3
for v in 0 1 9; do
  LT_HALO_F3=$v timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-pmc-leg --steps 5 --warmup 2 --full-line --ops-json $OUT/f3_ops_$v.json > $OUT/f3_$v.json 2> $OUT/f3_$v.err
done
python - <<'PY'
import json
for v in ("0", "1", "9"):
    d = json.load(open("gpurun_out/f3_%s.json" % v)); ops = json.load(open("gpurun_out/f3_ops_%s.json" % v))
    o3 = [o for o in ops if o["label"].startswith("conv3x3x3 32->32")]
    ms = sum(o["ms"] for o in o3); fl = sum(o["flops"] for o in o3)
    print("LT_HALO_F3=%s: fp32 forward %.1f samples/s, roofline %.3f | 3^3 32->32 x%d: %.2f ms %.0f TFLOP/s" % (v, d["value"], d["roofline"]["frac"], len(o3), ms, fl / ms / 1e9))
PY
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan_abi.py -m gpu -q --tb=short -p no:cacheprovider -k "fp32 or golden or f32" 2>&1 | tail -5
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_report.json"))
for k, v in d.items():
    if "joints fp32: max rel err vs the fp64" in k: print(k, v)
PY
