#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python tools/trace_kstep.py --dtype fp32 --batch 64 --only "rn l3,rn l2 3x3,rn l1 3x3" --tiles auto,128x128,128x64 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_kstep_fp32.log
timeout 600 python tools/trace_kstep.py --dtype bf16 --batch 64 --only "rn l3" --tiles 128x128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/trace_kstep_fp32.log
