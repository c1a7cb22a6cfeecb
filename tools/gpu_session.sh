#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel trace.  Usage (from the repo root on the GPU box):
#   bash tools/gpu_session.sh [tests] [bench] [prof] [pmc]
# Everything is written under gpurun_out/ (merged back by gpurun); each stage is time-boxed.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export PYTHONDONTWRITEBYTECODE=1
STAGES="${@:-tests bench prof}"
echo "== stages: $STAGES" | tee $OUT/session.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 | tee -a $OUT/session.log
nproc | tee -a $OUT/session.log
# on a box with >= 2 GPUs the N-rank path comes FIRST (VERDICT r5 "next" 7: it has never run on hardware): the two-rank RCCL test, then the preflight
NGPU=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
if [ "${NGPU:-0}" -ge 2 ]; then
  timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider -k "two_ranks_rccl" > $OUT/test_two_ranks_rccl.log 2>&1
  echo "two-rank RCCL test rc=$?" | tee -a $OUT/session.log; tail -3 $OUT/test_two_ranks_rccl.log | tee -a $OUT/session.log
  timeout 900 python bench.py --gpus $NGPU --preflight > $OUT/preflight.json 2> $OUT/preflight.err
  echo "preflight ($NGPU GPUs) rc=$?" | tee -a $OUT/session.log; cat $OUT/preflight.json | tee -a $OUT/session.log
fi
for s in $STAGES; do
case $s in
tests)
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_kernels.log 2>&1
  echo "kernels rc=$?" | tee -a $OUT/session.log; tail -5 $OUT/test_kernels.log | tee -a $OUT/session.log
  timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_models.log 2>&1
  echo "models rc=$?" | tee -a $OUT/session.log; tail -5 $OUT/test_models.log | tee -a $OUT/session.log
  timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_train.log 2>&1
  echo "backward+train rc=$?" | tee -a $OUT/session.log; tail -5 $OUT/test_train.log | tee -a $OUT/session.log
  ;;
train)
  # BASELINE config 5 in its present form: fp32 training step (fwd + bwd + Adam) at 4 and 8 samples per step, and its kernel profile
  for B in 4 8; do
    timeout 600 python bench.py --train --steps 10 --warmup 3 --batch $B > $OUT/bench_train_b$B.json 2> $OUT/bench_train_b$B.err
    echo "bench train B=$B rc=$?" | tee -a $OUT/session.log; cat $OUT/bench_train_b$B.json | tee -a $OUT/session.log
  done
  for B in 4 8; do
    timeout 600 python bench.py --train --train-dtype bf16 --steps 10 --warmup 3 --batch $B > $OUT/bench_train_bf16mma_b$B.json 2> $OUT/bench_train_bf16mma_b$B.err
    echo "bench train bf16-MFMA B=$B rc=$?" | tee -a $OUT/session.log; cat $OUT/bench_train_bf16mma_b$B.json | tee -a $OUT/session.log
  done
  rm -rf $OUT/prof_train
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o train -- python $R/bench.py --train --steps 5 --warmup 3 --batch 8 > $OUT/prof_train.json 2> $OUT/prof_train.err
  echo "prof train rc=$?" | tee -a $OUT/session.log
  find $OUT/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/train_kernel_stats.csv
  ;;
smoke)
  timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/session.log; tail -3 $OUT/smoke.log | tee -a $OUT/session.log
  ;;
bench)
  # the driver's command; the line carries the fp32 parity-mode leg, the parity block and the batch sweep
  timeout 900 python bench.py --ops-json $OUT/bench_ops_bf16.json > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
  echo "bench bf16 rc=$?" | tee -a $OUT/session.log; cat $OUT/bench_bf16.json | tee -a $OUT/session.log; tail -3 $OUT/bench_bf16.err
  ;;
c4)
  # BASELINE config 4: 8 views, 128^3 voxels (B = 8 per step by default)
  timeout 900 python bench.py --views 8 --volume 128 --steps 10 --warmup 3 --ops-json $OUT/bench_ops_c4.json > $OUT/bench_c4.json 2> $OUT/bench_c4.err
  echo "bench c4 rc=$?" | tee -a $OUT/session.log; cat $OUT/bench_c4.json | tee -a $OUT/session.log; tail -3 $OUT/bench_c4.err
  ;;
prof)
  cd /tmp; export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-extras > $OUT/prof_bench.json 2> $OUT/prof.err
  echo "prof rc=$?" | tee -a $OUT/session.log
  cd $R
  find $OUT/prof -name "*stats*" | head | tee -a $OUT/session.log
  ;;
ablibs)
  # per-layer A/B of library variants (lt_build.build_variant, built BEFORE the gpurun call): ABLIBS="base bpf1 noa ..."
  for round in 1 2; do
  for v in ${ABLIBS:-base}; do
    E="LT_AB=base"; [ $v != base ] && E="LT_HIP_LIB=$R/learnable-triangulation-pytorch_amd/lib/liblt_hip_$v.so"
    echo "== lib=$v round=$round" | tee -a $OUT/ablibs.log
    env $E timeout 300 python tools/conv_bench.py --batch ${AB_BATCH:-32} --only "${AB_ONLY:-rn l3}" --variants auto --residual --rounds 5 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ablibs.log
  done
  done
  ;;
convbench)
  timeout 600 python tools/conv_bench.py --json $OUT/conv_bench_bf16.json > $OUT/conv_bench_bf16.log 2>&1
  echo "convbench rc=$?" | tee -a $OUT/session.log; tail -3 $OUT/conv_bench_bf16.log
  ;;
pmcsq)
  cd /tmp; export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-graph > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
  echo "pmc sq rc=$?" | tee -a $OUT/session.log; ls $OUT/pmc_sq | head
  cd $R
  ;;
pmcconv)
  cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_conv -o cb -- python $R/tools/conv_bench.py --only v2v --variants halo,256x32/s2,256x16/s2,128x64/s2 --rounds 1 > $OUT/pmc_conv.log 2>&1
  echo "pmcconv rc=$?" | tee -a $OUT/session.log
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_conv2 -o cb -- python $R/tools/conv_bench.py --only v2v --variants halo,256x32/s2,256x16/s2,128x64/s2 --rounds 1 > $OUT/pmc_conv2.log 2>&1
  echo "pmcconv2 rc=$?" | tee -a $OUT/session.log
  cd $R
  ;;
abenv)
  # in-session A/B of env switches (two interleaved rounds each): baseline, no residual prefetch, no uniform-tap path, no halo kernel
  for round in 1 2; do
    for v in ${ABVARS:-base LT_CONV_NO_V3 LT_CONV_NO_XCD LT_HALO_NO_PERSIST LT_HALO_NO_H7 LT_HALO_NO_LDR}; do
      # LIB_<name>: an A/B build of the library (lt_build.build_variant), else an env switch read by the kernels' dispatchers
      case $v in
        base) E="LT_AB=base" ;;
        LIB_*) E="LT_HIP_LIB=$PWD/learnable-triangulation-pytorch_amd/lib/liblt_hip_${v#LIB_}.so" ;;
        *) E="$v=1" ;;
      esac
      env $E timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 10 --warmup 3 > $OUT/ab_${v}_$round.json 2> $OUT/ab_${v}_$round.err
      echo "ab $v round $round: $(python -c "import json;d=json.load(open('$OUT/ab_${v}_$round.json'));print('%.1f samples/s %.2f ms/step'%(d['value'],d['ms_per_step']))")" | tee -a $OUT/session.log
    done
  done
  ;;
ablate)
  # where does the time of the big conv layers go?  ablation builds of the library (results wrong by design, timing only):
  # no MFMAs / no epilogue / no A-side DMA (halo) / no B-side DMA (weights), for the persistent and the one-tile halo kernels
  L=$R/learnable-triangulation-pytorch_amd/lib
  python - <<'PYEOF'
import sys; sys.path.insert(0, "learnable-triangulation-pytorch_amd")
import lt_build
for n, d in (("abl_nomma", ["LT_ABL_NO_MMA"]), ("abl_noepi", ["LT_ABL_NO_EPI"]), ("abl_noa", ["LT_ABL_NO_A"]), ("abl_nob", ["LT_ABL_NO_B"])):
    lt_build.build_variant(n, d)
PYEOF
  for lib in base nomma noepi noa nob; do
    for pers in persist onetile; do
      E="LT_AB=1"; [ $lib != base ] && E="LT_HIP_LIB=$L/liblt_hip_abl_$lib.so"
      P="LT_AB2=1"; [ $pers = onetile ] && P="LT_HALO_NO_PERSIST=1 LT_HALO_NO_RING=1"
      [ $pers = onetile ] && ONLY="v2v 3^3 32->32,v2v 7^3" || ONLY="${ABL_LAYERS:-v2v 3^3 32->32,v2v 7^3,rn l3,rn l2 1x1 128,rn l1 1x1}"
      echo "== ablate lib=$lib halo=$pers" | tee -a $OUT/ablate.log
      env $E $P timeout 300 python tools/conv_bench.py --batch 16 --only "$ONLY" --variants auto --residual --rounds 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ablate.log
    done
  done
  ;;
trace)
  timeout 600 python tools/trace_kstep.py ${TRACE_ARGS:-} 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_kstep.log
  ;;
tracehalo)
  timeout 300 python tools/trace_halo.py 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_halo.log
  ;;
nst)
  # A/B: 2-stage vs 3-stage LDS-DMA ring in the v2 conv kernels
  for n in 2 3; do
    LT_CONV_NST=$n timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --ops-json $OUT/bench_ops_bf16_nst$n.json > $OUT/bench_bf16_nst$n.json 2> $OUT/bench_bf16_nst$n.err
    echo "bench nst=$n rc=$?" | tee -a $OUT/session.log; cut -c1-140 $OUT/bench_bf16_nst$n.json | tee -a $OUT/session.log
  done
  ;;
ab)
  # A/B: v1 (register staged) vs v2 (LDS-DMA) conv kernels, and batch-size sweep
  if [ -n "${WITH_V1:-}" ]; then
  LT_CONV_V1=1 timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --ops-json $OUT/bench_ops_bf16_v1.json > $OUT/bench_bf16_v1.json 2> $OUT/bench_bf16_v1.err
  echo "bench v1 rc=$?" | tee -a $OUT/session.log; cut -c1-400 $OUT/bench_bf16_v1.json | tee -a $OUT/session.log
  fi
  for b in 1 4 16 32; do
    timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 10 --warmup 3 --batch $b > $OUT/bench_bf16_b$b.json 2> $OUT/bench_bf16_b$b.err
    echo "bench B=$b rc=$?" | tee -a $OUT/session.log; cut -c1-330 $OUT/bench_bf16_b$b.json | tee -a $OUT/session.log
  done
  ;;
pmc)
  # counters in their own runs, --kernel-trace only (gpurun refuses --pmc together with the sys/hip/hsa trace domains); 6 forwards each
  cd /tmp; export TMPDIR=/tmp
  PB="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extras --no-graph --preroll-s 0 ${PMC_BENCH_ARGS:-}"
  SFX=${PMC_SUFFIX:-}
  rm -rf $OUT/pmc_fetch$SFX $OUT/pmc_write$SFX $OUT/pmc_mfma$SFX
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch$SFX -o bench -- python $R/bench.py $PB > $OUT/pmc_fetch$SFX.json 2> $OUT/pmc_fetch$SFX.err
  echo "pmc fetch rc=$?" | tee -a $OUT/session.log
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write$SFX -o bench -- python $R/bench.py $PB > $OUT/pmc_write$SFX.json 2> $OUT/pmc_write$SFX.err
  echo "pmc write rc=$?" | tee -a $OUT/session.log
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma$SFX -o bench -- python $R/bench.py $PB > $OUT/pmc_mfma$SFX.json 2> $OUT/pmc_mfma$SFX.err
  echo "pmc mfma rc=$?" | tee -a $OUT/session.log
  cd $R
  python tools/pmc_summary.py $OUT 6 ${PMC_BATCH:-32} --suffix "$SFX" ${PMC_SUMMARY_ARGS:-} 2>&1 | tee -a $OUT/session.log
  cp profiles/r02_hbm_traffic_pmc$SFX.json $OUT/ 2>/dev/null
  ;;
esac
done
ls -la $OUT | tee -a $OUT/session.log
