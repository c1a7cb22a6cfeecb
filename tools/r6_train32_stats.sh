#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export PYTHONDONTWRITEBYTECODE=1
rm -rf /tmp/kt32
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt32 -o t -- python $R/bench.py --train --batch 4 --steps 4 --warmup 2 --no-pmc-leg > $OUT/kt32_line.json 2> $OUT/kt32.err; echo rc=$?
F=$(find /tmp/kt32 -name "*kernel_stats.csv" | head -1); cp $F $OUT/train_fp32_b4_kernel_stats.csv
head -32 $F | cut -c1-230
