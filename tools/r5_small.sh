#!/bin/bash
# round 5: does the 2D halo kernel pay below its two-round threshold?  The forward at 5 / 10 / 32 samples with and without LT_H2D_ANY_SIZE=1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
for B in 5 10 32; do
for v in thr any; do
  E="LT_X=1"; [ $v = any ] && E="LT_H2D_ANY_SIZE=1"
  env $E timeout 600 python bench.py --batch $B --no-extras --no-cpu-baseline --no-pmc-leg --no-profile --ops-json $OUT/small_ops_${B}_$v.json > $OUT/small_bench_${B}_$v.json 2> $OUT/small_bench_${B}_$v.err
  echo "B=$B $v rc=$?  $(python -c "import json;d=json.load(open('$OUT/small_bench_${B}_$v.json'));print(d['value'], d['ms_per_step'])")"
done
python - $B <<'PY'
import json, sys
B=sys.argv[1]
def load(f):
    d=json.load(open(f)); g={}
    for o in d:
        k=o['label'].split(' + ')[0]
        e=g.setdefault(k,[0,0.0]); e[0]+=1; e[1]+=o['ms']
    return g
a=load('gpurun_out/small_ops_%s_thr.json'%B); b=load('gpurun_out/small_ops_%s_any.json'%B)
for k in b:
    if k in a and (k.startswith('conv3x3 256->256') or k.startswith('deconv4x4 256->256')):
        print("   %-44s n=%d  %.1f -> %.1f us"%(k[:44], b[k][0], 1e3*a[k][1]/a[k][0], 1e3*b[k][1]/b[k][0]))
PY
done
