#!/bin/bash
# per-launch tables of the forward at small batches: bash tools/r5_small_ops.sh 5 1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
for B in "$@"; do
  timeout 600 python bench.py --batch $B --no-extras --no-cpu-baseline --no-pmc-leg --ops-json $OUT/small_ops_$B.json > $OUT/small_bench_$B.json 2> $OUT/small_bench_$B.err
  echo "B=$B rc=$?  $(python -c "import json;d=json.load(open('$OUT/small_bench_$B.json'));print(d['value'], d['ms_per_step'], d['roofline']['ms_per_step_in_kernel'] if 'ms_per_step_in_kernel' in d['roofline'] else d['roofline'].get('ms_in_kernel'))")"
  python tools/ops_top.py $OUT/small_ops_$B.json 28
done
