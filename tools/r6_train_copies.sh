#!/bin/bash
# which launches of the act16 training step are __amd_rocclr_copyBuffer?  kernel trace of three steps, neighbours and sizes of every copy in the last one
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export PYTHONDONTWRITEBYTECODE=1
rm -rf /tmp/kt
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python $R/bench.py --train --train-dtype act16 --batch 8 --steps 3 --warmup 2 --no-pmc-leg > $OUT/kt_line.json 2> $OUT/kt.err; echo rc=$?
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); ls -la $F
python - "$F" <<'PY' | tee $OUT/train_copies_trace.log
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "kernels; columns:", list(rows[0].keys()))
names = [r["Kernel_Name"] for r in rows]
# the last step: after the last adam_multi_kernel but one
adam = [i for i, n in enumerate(names) if "adam_multi" in n]
print("adam launches at", adam[-6:])
i0, i1 = adam[-2] + 1, adam[-1] + 1
seg = rows[i0:i1]
print("last step:", len(seg), "kernels")
pairs = collections.Counter(); sizes = collections.Counter(); dur = collections.defaultdict(float)
for j, r in enumerate(seg):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = seg[j - 1]["Kernel_Name"][:60] if j else "-"
        nxt = seg[j + 1]["Kernel_Name"][:60] if j + 1 < len(seg) else "-"
        k = (prev, nxt)
        pairs[k] += 1
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        dur[k] += d
        sizes[(r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"), r.get("Stream_Id", r.get("Queue_Id", "?")))] += 1
print("copies in the last step:", sum(pairs.values()), "total us %.1f" % sum(dur.values()))
for k, c in pairs.most_common(25):
    print("  %4d x %7.1f us  after %-60s before %s" % (c, dur[k], k[0], k[1]))
print("grid / workgroup / stream:", sizes.most_common(12))
tot = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = r["Kernel_Name"]; n = n[n.find("::") + 2:] if n.startswith("void (anonymous") or n.startswith("(anonymous") else n
    t = tot[n[:70]]; t[0] += 1; t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
wall = (max(int(r["End_Timestamp"]) for r in seg) - min(int(r["Start_Timestamp"]) for r in seg)) / 1e3
ksum = sum(v[1] for v in tot.values())
print("last step: wall %.1f us, summed kernel time %.1f us" % (wall, ksum))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %5d x %9.1f us  %5.2f %%  %s" % (c, t, 100 * t / ksum, n))
PY
