#!/bin/bash
# round 6, first session: the whole GPU suite (new bf16 gates in measure mode), smoke, the driver's bench command with the per-op table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_models.log 2>&1; echo "models rc=$?"; tail -5 $OUT/test_models.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/test_kernels.log
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_train.log 2>&1; echo "train rc=$?"; tail -3 $OUT/test_train.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py --ops-json $OUT/bench_ops_bf16.json > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_bf16.json
