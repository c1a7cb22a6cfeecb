#!/bin/bash
# round 6: the gather at config 4 -- brick order (LT_UNPROJ_RASTER=1 = the old raster order) x occupancy (liblt_hip_occ3.so = the old 3 waves per SIMD), live PMC traffic
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "unproject" 2>&1 | tail -3
L=$R/learnable-triangulation-pytorch_amd/lib
for v in new raster occ3 raster_occ3 new raster; do
  E="LT_X=1"
  case $v in raster) E="LT_UNPROJ_RASTER=1";; occ3) E="LT_HIP_LIB=$L/liblt_hip_occ3.so";; raster_occ3) E="LT_UNPROJ_RASTER=1 LT_HIP_LIB=$L/liblt_hip_occ3.so";; esac
  env $E timeout 900 python bench.py --views 8 --volume 128 --batch 16 --steps 8 --warmup 3 --no-extras --full-line --no-cpu-baseline --force-pmc-leg --preroll-s 0.3 > $OUT/ab_c4_$v.json 2> $OUT/ab_c4_$v.err
  echo "c4 $v rc=$?: $(python -c "
import json;d=json.load(open('$OUT/ab_c4_$v.json'));u=d['roofline_hbm']['unproject'];print(d['value'], d['ms_per_step'], 'unproject ms', u['ms_per_step_in_kernel'], 'frac', round(u['frac'],4), 'traffic', u.get('traffic'), 'bytes', u['bytes_per_step'], 'valu', u.get('valu_issue_frac'))")"
done
for v in new raster new raster; do
  E="LT_X=1"; [ $v = raster ] && E="LT_UNPROJ_RASTER=1"
  env $E timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc-leg --full-line > $OUT/ab_c2_$v.json 2> $OUT/ab_c2_$v.err
  echo "c2 $v rc=$?: $(python -c "
import json;d=json.load(open('$OUT/ab_c2_$v.json'));u=d['roofline_hbm']['unproject'];print(d['value'], d['ms_per_step'], 'unproject ms', u['ms_per_step_in_kernel'])")"
done
