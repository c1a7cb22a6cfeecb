#!/bin/bash
# round 5, GPU session 7: lt_expand_reduce_fwd (final structure) -- parity, model goldens, per-launch timing, in-session A/B of the driver workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "expand_reduce" 2>&1 | tail -3
timeout 300 python tools/xr_bench.py --images 256 2>&1 | grep -v amdgpu.ids | tee $OUT/s7_xr_bench.log
timeout 300 python tools/xr_bench.py --images 128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/s7_xr_bench.log
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/s7_models.log 2>&1
echo "models rc=$?"; tail -4 $OUT/s7_models.log
for v in xr noxr xr noxr; do
  E="LT_X=1"; [ $v = noxr ] && E="LT_NO_XR=1"
  env $E timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc-leg --ops-json $OUT/s7_ops_$v.json > $OUT/s7_bench_$v.json 2> $OUT/s7_bench_$v.err
  echo "bench $v rc=$?"; python -c "import json;d=json.load(open('$OUT/s7_bench_$v.json'));print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
python - <<'PY'
import json
for v in ('xr','noxr'):
    d=json.load(open('gpurun_out/s7_ops_%s.json'%v))
    g={}
    for o in d:
        if '1024' in o['label'] and '24x24' in o['label']: e=g.setdefault(o['label'],[0,0.0]); e[0]+=1; e[1]+=o['ms']
    for k,e in g.items(): print(v,k,e[0],round(1e3*e[1]/e[0],1),'us')
PY
