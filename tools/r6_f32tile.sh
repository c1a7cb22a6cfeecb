#!/bin/bash
# exact-fp32 backbone: the 128 x 64 tile (two to three waves per SIMD) against the 128 x 128 tile (320 registers: one wave per SIMD), per layer group
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
for t in 0 2 5; do
  timeout 600 python bench.py --dtype fp32 --tile $t --no-cpu-baseline --no-extras --no-pmc-leg --steps 3 --warmup 1 --full-line --ops-json $OUT/ft_ops_$t.json > $OUT/ft_$t.json 2> $OUT/ft_$t.err
done
python - <<'PY'
import json, re, collections
tabs = {}
for t in ("0", "2", "5"):
    ops = json.load(open("gpurun_out/ft_ops_%s.json" % t))
    agg = collections.OrderedDict()
    for o in ops:
        l = re.sub(r"@.*", "", o["label"]).strip()
        a = agg.setdefault(l, [0, 0.0, 0]); a[0] += 1; a[1] += o["ms"]; a[2] += o["flops"]
    tabs[t] = agg
    print("tile", t, "total %.1f ms" % sum(o["ms"] for o in ops))
for l, (n, ms, f) in sorted(tabs["0"].items(), key=lambda kv: -kv[1][1])[:22]:
    print("%3d x %-22s auto %7.2f ms %5.1f TF/s | 128x64 %7.2f ms | 64x64 %7.2f ms" % (n, l[:22], ms, f / ms / 1e9 if ms else 0, tabs["2"].get(l, [0, 0, 0])[1], tabs["5"].get(l, [0, 0, 0])[1]))
PY
