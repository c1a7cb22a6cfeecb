#!/bin/bash
# round 6, third session: the whole GPU suite after the 64-bit / live-fragment / gate changes, fp32 per-op table, config-4 PMC of the gather
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_models.log 2>&1; echo "models rc=$?"; tail -5 $OUT/test_models.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_kernels.log 2>&1; echo "kernels rc=$?"; tail -5 $OUT/test_kernels.log
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_train.log 2>&1; echo "train rc=$?"; tail -5 $OUT/test_train.log
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-pmc-leg --steps 5 --warmup 2 --ops-json $OUT/bench_ops_fp32_b64.json > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err; echo "bench fp32 rc=$?"; cut -c1-300 $OUT/bench_fp32.json
timeout 300 python bench.py --gpus 1 --preflight > $OUT/preflight_1gpu.json 2> $OUT/preflight_1gpu.err; echo "preflight rc=$?"; cat $OUT/preflight_1gpu.json
