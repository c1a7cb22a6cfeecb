#!/bin/bash
# round 6: the pipelined fp32 K step of conv_igemm2 (PIPE32) -- tests, shader-clock trace, A/B of the fp32 forward against the -DLT_FP32_NO_PIPE build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "conv or halo" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "vs_reference_golden" 2>&1 | tail -3
timeout 600 python tools/trace_kstep.py --dtype fp32 --batch 64 --only "rn l3,rn l2 3x3" --tiles auto 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_kstep_fp32_pipe.log
L=$R/learnable-triangulation-pytorch_amd/lib
for v in pipe nopipe pipe nopipe; do
  E="LT_X=1"; [ $v = nopipe ] && E="LT_HIP_LIB=$L/liblt_hip_nopipe.so"
  env $E timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-extras --no-pmc-leg --steps 5 --warmup 2 --full-line --ops-json $OUT/ab_ops_fp32_$v.json > $OUT/ab_fp32_$v.json 2> $OUT/ab_fp32_$v.err
  echo "fp32 $v rc=$?: $(python -c "import json;d=json.load(open('$OUT/ab_fp32_$v.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done
python tools/ops_top.py $OUT/ab_ops_fp32_pipe.json 12
