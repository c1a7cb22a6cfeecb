#!/bin/bash
# round 5: the seam kernel's tile height at small batches -- tests, then the forward at 1 / 2 / 5 / 10 samples with LT_XR_NPB = 3 (the old shape) / 2 / 1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "expand_reduce" 2>&1 | tail -3
for B in "$@"; do
for v in 3 2 1 off; do
  E="LT_XR_NPB=$v LT_XR_ANY_SIZE=1"; [ $v = off ] && E="LT_NO_XR=1"
  env $E timeout 600 python bench.py --batch $B --no-extras --no-cpu-baseline --no-pmc-leg --no-profile > $OUT/xrs_bench_${B}_$v.json 2> $OUT/xrs_bench_${B}_$v.err
  echo "B=$B npb=$v rc=$?  $(python -c "import json;d=json.load(open('$OUT/xrs_bench_${B}_$v.json'));print(d['value'], d['ms_per_step'])")"
done
done
