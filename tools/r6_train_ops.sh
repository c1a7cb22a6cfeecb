#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python tools/train_profile.py 8 act16 > $OUT/train_ops_b8_act16.log 2> $OUT/train_ops.err; echo rc=$?
python - <<'PY'
import json, re
d = json.load(open("gpurun_out/train_ops_b8_act16.json"))
for name in ("fwd", "bwd"):
    rows = [(lab, n, ms) for lab, n, ms in d[name] if lab.startswith("bn")]
    tot = sum(ms for _, _, ms in rows)
    print(name, "bn ops: %.2f ms" % tot)
    for lab, n, ms in rows[:40]:
        m = re.search(r"(\d+)x(\d+)", lab)
        el = int(m.group(1)) * int(m.group(2)) if m else 0
        print("   %7.3f ms x%-3d %-28s %6.1f M elements  %.2f us per launch, %.1f B/ns per element-byte" % (ms, n, lab, el / 1e6, 1e3 * ms / n, el / (1e6 * ms / n) if ms else 0))
PY
