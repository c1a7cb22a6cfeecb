#!/bin/bash
# round 5, GPU session 1: the new parity gates (fp8v2v whole step, act16 at B = 2 / 8, the three-precision trajectory) and the driver command
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "test_mixed_precision_training_step_deviation_and_descent or test_act16_step_tracks_fp32_at_the_config2_shape or test_training_trajectory" -s > $OUT/s1_tests.log 2>&1
echo "tests rc=$?"; tail -15 $OUT/s1_tests.log
timeout 1200 python bench.py > $OUT/s1_bench.json 2> $OUT/s1_bench.err
echo "bench rc=$?"; wc -c $OUT/s1_bench.json; cat $OUT/s1_bench.json; tail -c 1500 $OUT/s1_bench.err
