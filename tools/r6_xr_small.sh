#!/bin/bash
# seam-kernel tile height between the measured points of round 5 (5 and 10 samples): forward samples/s at 6 .. 9 samples with LT_XR_NPB = 2 / 3 (default rule: 2 below 224 tiles of 96 rows)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
for B in 3 4; do
for v in 1 2 1 2; do
  LT_XR_NPB=$v timeout 300 python bench.py --batch $B --no-extras --no-cpu-baseline --no-pmc-leg --no-profile --full-line --steps 40 --warmup 10 > $OUT/xs_b${B}_$v.json 2> $OUT/xs.err
  echo "B=$B npb=$v: $(python -c "import json;d=json.load(open('$OUT/xs_b${B}_$v.json'));print(round(d['value'],1), round(d['ms_per_step'],3))")"
done
done
