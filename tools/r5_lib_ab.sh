#!/bin/bash
# in-session A/B of library build variants (lt_build.build_variant) on the driver workload:
#   bash tools/r5_lib_ab.sh <variant> [<variant> ...]      ("base" = the shipped liblt_hip.so); each is run twice, interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
L=$R/learnable-triangulation-pytorch_amd/lib
for rep in 1 2; do
  for v in base "$@"; do
    lib=$L/liblt_hip.so; [ $v != base ] && lib=$L/liblt_hip_$v.so
    LT_HIP_LIB=$lib timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc-leg --ops-json $OUT/lab_ops_${v}.json > $OUT/lab_bench_${v}_$rep.json 2> $OUT/lab_bench_${v}_$rep.err
    echo "bench $v rep $rep rc=$?  $(python -c "import json;d=json.load(open('$OUT/lab_bench_${v}_$rep.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
  done
done
python - "$@" <<'PY'
import json, sys
def load(f):
    d=json.load(open(f)); g={}
    for o in d:
        e=g.setdefault(o['label'],[0,0.0]); e[0]+=1; e[1]+=o['ms']
    return g
a=load('gpurun_out/lab_ops_base.json')
for v in sys.argv[1:]:
    b=load('gpurun_out/lab_ops_%s.json'%v)
    rows=sorted(((b[k][1]-a[k][1],k) for k in b if k in a))
    print("==", v, "total %+.3f ms"%sum(r[0] for r in rows)); [print("   %s: %.1f -> %.1f us"%(k[:44], 1e3*a[k][1]/a[k][0], 1e3*b[k][1]/b[k][0])) for k in b if k in a and (k.startswith("conv3x3 256->256 @256") or k.startswith("deconv4x4 256->256"))]
    for d,k in rows[:5]+rows[-3:]: print("%+.3f ms  %-55s n=%d  %.1f -> %.1f us"%(d,k[:55],b[k][0],1e3*a[k][1]/a[k][0],1e3*b[k][1]/b[k][0]))
PY
