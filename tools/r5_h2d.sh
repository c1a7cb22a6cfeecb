#!/bin/bash
# round 5: the 2D halo kernel (ResNet layer3's 3x3, the head's 4x4 transposed convolutions) -- tests, then the driver command with / without the transposed part
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv2d or deconv" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
for v in off on; do
  E="LT_X=1"; [ $v = off ] && E="${1:-LT_DECONV_NO_H2D}=1"
  env $E timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc-leg --ops-json $OUT/h2d_ops_$v.json > $OUT/h2d_bench_$v.json 2> $OUT/h2d_bench_$v.err
  echo "bench $v ($E) rc=$?  $(python -c "import json;d=json.load(open('$OUT/h2d_bench_$v.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done
done
python - <<'PY'
import json
for v in ('off','on'):
    d=json.load(open('gpurun_out/h2d_ops_%s.json'%v))
    for pre in ('conv3x3 256->256 @256x1x24x24','deconv4x4 256->256 @256x1x48x48','deconv4x4 256->256 @256x1x24x24'):
        r=[o for o in d if o['label'].startswith(pre)]
        print(v, pre, len(r), "%.1f us avg"%(1e3*sum(o['ms'] for o in r)/max(1,len(r))))
    print(v, "total %.2f ms"%sum(o['ms'] for o in d))
PY
