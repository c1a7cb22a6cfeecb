#!/bin/bash
# round 5: the 2D halo kernel at small batches -- the forward at the given sample counts with the environment switch of $1 off / on:  bash tools/r5_small.sh LT_H2D_TH=8 5 10 16 32
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
SW=$1; shift
for B in "$@"; do
for v in sw def sw def; do
  E="LT_X=1"; [ $v = sw ] && E="$SW"
  env $E timeout 600 python bench.py --batch $B --no-extras --no-cpu-baseline --no-pmc-leg --no-profile > $OUT/small_bench_${B}_$v.json 2> $OUT/small_bench_${B}_$v.err
  echo "B=$B $v ($E) rc=$?  $(python -c "import json;d=json.load(open('$OUT/small_bench_${B}_$v.json'));print(d['value'], d['ms_per_step'])")"
done
done
