#!/bin/bash
# round 6, second session: regenerated B=8 golden, live fragment weights of the act16 tape (tests + A/B), 128-pixel seam tiles (test + A/B)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "c2_b8 or first_block" > $OUT/test_models_b8.log 2>&1; echo "models b8 rc=$?"; tail -5 $OUT/test_models_b8.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "expand_reduce or conv2d_halo or deconv4x4" > $OUT/test_kernels_xr.log 2>&1; echo "kernels xr rc=$?"; tail -3 $OUT/test_kernels_xr.log
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/test_train.log 2>&1; echo "train rc=$?"; tail -8 $OUT/test_train.log
for v in 3 4 3 4; do
  LT_XR_NPB=$v timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc-leg --ops-json $OUT/ab_ops_npb$v.json > $OUT/ab_npb$v.json 2> $OUT/ab_npb$v.err
  echo "xr npb=$v rc=$?: $(python -c "import json;d=json.load(open('$OUT/ab_npb$v.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done
python tools/ops_top.py $OUT/ab_ops_npb3.json 3; python tools/ops_top.py $OUT/ab_ops_npb4.json 3
for B in 8 16 32; do
for v in frag plain frag plain; do
  E="LT_X=1"; [ $v = plain ] && E="LT_TRAIN_NO_FRAG=1"
  env $E timeout 900 python bench.py --train --train-dtype act16 --batch $B --steps 6 --warmup 2 --no-pmc-leg > $OUT/ab_train_${v}_b$B.json 2> $OUT/ab_train_${v}_b$B.err
  echo "train $v B=$B rc=$?: $(python -c "import json;d=json.load(open('$OUT/ab_train_${v}_b$B.json'));print(d['value'], d['ms_per_step'], d['loss_first_last'])")"
done
done
