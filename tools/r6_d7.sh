#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "halo" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider -k "tape_layer_gradients and act16 and nd3" 2>&1 | tail -3
for v in d7 nod7 d7 nod7; do
  E="LT_X=1"; [ $v = nod7 ] && E="LT_HALO_NO_D7=1"
  env $E timeout 900 python bench.py --train --train-dtype act16 --batch 8 --steps 6 --warmup 2 --no-pmc-leg > $OUT/ab_train_$v.json 2> $OUT/ab_train_$v.err
  echo "train $v rc=$?: $(python -c "import json;d=json.load(open('$OUT/ab_train_$v.json'));print(d['value'], d['ms_per_step'], d['loss_first_last'])")"
done
