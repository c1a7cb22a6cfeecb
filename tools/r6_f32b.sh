#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
for i in 1 2; do
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-pmc-leg --steps 5 --warmup 2 --full-line --ops-json $OUT/f32b_ops.json > $OUT/f32b.json 2> $OUT/f32b.err
python -c "
import json; d=json.load(open('gpurun_out/f32b.json')); print('fp32 forward %.1f samples/s, %.3f' % (d['value'], d['roofline']['frac']))"
done
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan_abi.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "fp32 or golden or f32 or conv_all_tiles" 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_report.json"))
for k, v in d.items():
    if "joints fp32: max rel err vs the fp64" in k: print(k, v)
PY
