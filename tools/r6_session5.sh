#!/bin/bash
# round 6: kernel profile + per-kernel HBM traffic of the act16 training step (live fragment weights), MFMA-busy of the fp32 forward
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $R/bench.py --train --train-dtype act16 --steps 6 --warmup 3 --batch 8 --no-pmc-leg > $OUT/prof_train.json 2> $OUT/prof_train.err
echo "prof train rc=$?"
find $OUT/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/train_kernel_stats_act16_b8.csv
rm -rf $OUT/prof_train
cd $R
bash tools/r6_train_pmc.sh 2>&1 | tail -30
timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-extras --force-pmc-leg --steps 5 --warmup 2 --full-line > $OUT/bench_fp32_pmc.json 2> $OUT/bench_fp32_pmc.err; echo "fp32 pmc rc=$?"
python -c "import json;d=json.load(open('$OUT/bench_fp32_pmc.json'));print(d['value'], json.dumps(d['roofline']))"
