#!/bin/bash
# one iteration on lt_expand_reduce_fwd: parity, timing against the two launches, wave trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "expand_reduce" 2>&1 | tail -3
timeout 300 python tools/xr_bench.py --images 256 2>&1 | grep -v amdgpu.ids | tee $OUT/xr_iter.log
timeout 300 python tools/xr_bench.py --images 128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xr_iter.log
LT_HIP_LIB=$R/learnable-triangulation-pytorch_amd/lib/liblt_hip_xrtrace.so timeout 300 python tools/xr_bench.py --images 256 --trace 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xr_iter.log
