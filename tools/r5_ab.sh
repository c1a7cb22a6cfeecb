#!/bin/bash
# in-session A/B of an environment switch on the driver workload: bash tools/r5_ab.sh <ENV_VAR> [extra pytest -k expression]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
V=$1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "${2:-conv}" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
for v in on off on off; do
  E="LT_X=1"; [ $v = off ] && E="$V=1"
  env $E timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc-leg --ops-json $OUT/ab_ops_$v.json > $OUT/ab_bench_$v.json 2> $OUT/ab_bench_$v.err
  echo "bench $v ($E) rc=$?"; python -c "import json;d=json.load(open('$OUT/ab_bench_$v.json'));print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
python - <<'PY'
import json
def load(f):
    d=json.load(open(f)); g={}
    for o in d:
        e=g.setdefault(o['label'],[0,0.0]); e[0]+=1; e[1]+=o['ms']
    return g
a=load('gpurun_out/ab_ops_off.json'); b=load('gpurun_out/ab_ops_on.json')
rows=sorted(((b[k][1]-a[k][1],k) for k in b if k in a))
for d,k in rows[:6]+rows[-3:]: print("%+.3f ms  %-55s n=%d  %.1f -> %.1f us"%(d,k[:55],b[k][0],1e3*a[k][1]/a[k][0],1e3*b[k][1]/b[k][0]))
PY
