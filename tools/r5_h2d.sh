#!/bin/bash
# round 5: the 2D halo kernel for ResNet layer3's 3x3 -- tests, then the driver command with it off / 8-row tiles / 4-row tiles
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv2d" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
for v in off th8 th4; do
  E="LT_X=1"; [ $v = off ] && E="LT_CONV_NO_H2D=1"; [ $v = th4 ] && E="LT_H2D_TH=4"
  env $E timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc-leg --ops-json $OUT/h2d_ops_$v.json > $OUT/h2d_bench_$v.json 2> $OUT/h2d_bench_$v.err
  echo "bench $v ($E) rc=$?  $(python -c "import json;d=json.load(open('$OUT/h2d_bench_$v.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done
done
python - <<'PY'
import json
for v in ('off','th8','th4'):
    d=json.load(open('gpurun_out/h2d_ops_%s.json'%v))
    r=[o for o in d if o['label'].startswith('conv3x3 256->256 @256x1x24x24')]
    print(v, len(r), "%.1f us avg"%(1e3*sum(o['ms'] for o in r)/max(1,len(r))), "total %.2f ms"%sum(o['ms'] for o in d))
PY
