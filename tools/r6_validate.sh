#!/bin/bash
# round 6 validation session: the whole GPU suite, smoke, the driver's command, rocprofv3 kernel statistics (forward and the act16 training step), per-kernel PMC passes
# (forward at config 2 and at config 4, the training step), the fp32 forward with live PMC
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
STAGES="${@:-tests smoke bench prof pmc pmc_c4 trainpmc fp32}"
pmc_passes() {   # $1 = suffix, $2.. = bench arguments
  local SFX=$1; shift
  cd /tmp; export TMPDIR=/tmp
  local PB="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extras --no-graph --preroll-s 0 --no-pmc-leg $*"
  rm -rf $OUT/pmc_fetch$SFX $OUT/pmc_write$SFX $OUT/pmc_mfma$SFX $OUT/pmc_valu$SFX
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch$SFX -o bench -- python $R/bench.py $PB > /dev/null 2> $OUT/v_pmc_fetch$SFX.err; echo "pmc fetch$SFX rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write$SFX -o bench -- python $R/bench.py $PB > /dev/null 2> $OUT/v_pmc_write$SFX.err; echo "pmc write$SFX rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma$SFX -o bench -- python $R/bench.py $PB > /dev/null 2> $OUT/v_pmc_mfma$SFX.err; echo "pmc mfma$SFX rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu$SFX -o bench -- python $R/bench.py $PB > /dev/null 2> $OUT/v_pmc_valu$SFX.err; echo "pmc valu$SFX rc=$?"
  cd $R
}
for s in $STAGES; do
case $s in
tests)
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/v_tests.log 2>&1
  echo "gpu tests rc=$?"; tail -6 $OUT/v_tests.log
  ;;
smoke)
  timeout 600 python __graft_entry__.py smoke > $OUT/v_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/v_smoke.log
  ;;
bench)
  timeout 2400 python bench.py --ops-json $OUT/v_bench_ops.json > $OUT/v_bench.json 2> $OUT/v_bench.err
  echo "bench rc=$?"; wc -c $OUT/v_bench.json; cat $OUT/v_bench.json
  ;;
prof)
  cd /tmp; export TMPDIR=/tmp
  rm -rf $OUT/prof_fwd $OUT/prof_train
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-extras --no-pmc-leg > $OUT/v_prof_bench_line.json 2> $OUT/v_prof_fwd.err
  echo "prof fwd rc=$?"
  find $OUT/prof_fwd -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/v_kernel_stats_bench_bf16.csv
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $R/bench.py --train --train-dtype act16 --batch 8 --steps 6 --warmup 3 --no-pmc-leg > $OUT/v_prof_train_line.json 2> $OUT/v_prof_train.err
  echo "prof train rc=$?"
  find $OUT/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/v_kernel_stats_train_act16_b8.csv
  rm -rf $OUT/prof_fwd $OUT/prof_train
  cd $R
  ;;
pmc)
  pmc_passes ""
  python tools/pmc_summary.py $OUT 6 64 --round 6 2>&1 | tail -2
  cp profiles/r06_hbm_traffic_pmc.json $OUT/
  rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_valu
  ;;
pmc_c4)
  pmc_passes "_c4" --views 8 --volume 128 --batch 16
  python tools/pmc_summary.py $OUT 6 16 --round 6 --views 8 --volume 128 --suffix _c4 2>&1 | tail -2
  cp profiles/r06_hbm_traffic_pmc_c4.json $OUT/
  rm -rf $OUT/pmc_fetch_c4 $OUT/pmc_write_c4 $OUT/pmc_mfma_c4 $OUT/pmc_valu_c4
  ;;
trainpmc)
  bash tools/r6_train_pmc.sh 2>&1 | tail -12
  ;;
fp32)
  timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-extras --force-pmc-leg --steps 5 --warmup 2 --full-line --ops-json $OUT/v_bench_ops_fp32.json > $OUT/v_bench_fp32.json 2> $OUT/v_bench_fp32.err; echo "fp32 rc=$?"
  python -c "import json;d=json.load(open('$OUT/v_bench_fp32.json'));print(d['value'], json.dumps(d['roofline']))"
  ;;
esac
done
