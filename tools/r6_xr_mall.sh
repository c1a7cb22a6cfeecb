#!/bin/bash
# does the seam kernel run faster when its streams come from the memory-side cache?  lt_expand_reduce_fwd re-run on the SAME buffers at image counts whose
# working set (t2 + residual + y + t1 = 2.95 MB per image) is below / at / above the 256 MB Infinity Cache, all whole rounds of the chip (42 images = 252 tiles of 96 rows)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
: > $OUT/xr_mall.log
for n in 42 84 126 168 256 512; do
  timeout 300 python tools/xr_bench.py --images $n --rounds 8 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xr_mall.log
done
