#!/bin/bash
# shader clock and socket power while the bf16 / exact-fp32 forward runs in a loop (rocm-smi polled every ~0.25 s from a second process)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export PYTHONDONTWRITEBYTECODE=1
poll() {   # $1 = tag, runs until the file $OUT/clk_stop exists
  : > $OUT/clk_$1.log
  while [ ! -e $OUT/clk_stop ]; do
    /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' ' >> $OUT/clk_$1.log; echo >> $OUT/clk_$1.log
    sleep 0.25
  done
}
for cfg in "bf16 300" "fp32 40"; do
  set -- $cfg
  rm -f $OUT/clk_stop
  poll $1 & PP=$!
  timeout 600 python bench.py --dtype $1 --steps $2 --warmup 5 --no-cpu-baseline --no-profile --no-extras --no-pmc-leg > $OUT/clk_bench_$1.json 2> $OUT/clk_bench_$1.err
  touch $OUT/clk_stop; wait $PP
  python - $1 <<'PY'
import json, re, sys
t = sys.argv[1]
d = json.loads(open("gpurun_out/clk_bench_%s.json" % t).readline())
rows = [l for l in open("gpurun_out/clk_%s.log" % t) if "sclk" in l]
sclk = [int(m.group(1)) for l in rows for m in [re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)] if m]
pw = [float(m.group(1)) for l in rows for m in [re.search(r"Power \(W\): ([\d.]+)", l)] if m]
print(t, "forward %.1f samples/s, roofline %.3f" % (d["value"], d["roofline"]["frac"]), "| %d samples of rocm-smi" % len(rows))
print("   sclk MHz: all", sorted(set(sclk)), " last third:", sclk[-max(1, len(sclk) // 3):][:12])
print("   power W: max %.0f, last third %s" % (max(pw) if pw else -1, [round(p) for p in pw[-max(1, len(pw) // 3):][:12]]))
print("   sample line:", rows[len(rows) // 2].strip()[:300] if rows else None)
PY
done
