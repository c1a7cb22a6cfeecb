#!/bin/bash
# the whole GPU suite, file by file (logs under gpurun_out/)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
for f in models kernels plan_abi; do
  timeout 1500 python -m pytest tests/test_gpu_$f.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_$f.log 2>&1; echo "$f rc=$?"; tail -4 $OUT/test_$f.log
done
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_train.log 2>&1; echo "train rc=$?"; tail -4 $OUT/test_train.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
