#!/bin/bash
# per-launch tables of the bf16 forward at 64 / 8 / 1 samples per step (the "current state" table of DESIGN.md is generated from these: tools/plan_table.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
for B in 64 8 1; do
  timeout 600 python bench.py --batch $B --no-extras --no-cpu-baseline --no-pmc-leg --full-line --ops-json $OUT/bench_ops_bf16_b$B.json > $OUT/bench_bf16_b$B.json 2> $OUT/bench_bf16_b$B.err
  echo "B=$B rc=$?: $(python -c "import json;d=json.load(open('$OUT/bench_bf16_b$B.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done
