#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 1800 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan_abi.py tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_report.json"))
for k, v in d.items():
    if "joints fp32: max rel err vs the fp64" in k or "ORDER noise" in k: print(k, v)
PY
