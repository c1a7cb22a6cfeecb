#!/bin/bash
# rocprofv3 kernel statistics of the exact-fp32 forward (the tolerance-meeting mode), 64 samples per step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export PYTHONDONTWRITEBYTECODE=1
rm -rf /tmp/pf32
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf32 -o t -- python $R/bench.py --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-extras --no-pmc-leg > $OUT/prof_fp32_line.json 2> $OUT/prof_fp32.err; echo rc=$?
find /tmp/pf32 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_fp32.csv
head -14 $OUT/kernel_stats_bench_fp32.csv | cut -c1-200; cat $OUT/prof_fp32_line.json | cut -c1-200
