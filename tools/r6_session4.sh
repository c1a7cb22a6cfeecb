#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_gpu_plan_abi.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_plan_abi.log 2>&1; echo "plan abi rc=$?"; tail -40 $OUT/test_plan_abi.log
