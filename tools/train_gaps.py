#!/usr/bin/env python
"""Timeline accounting of the recorded training step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d D -o t -- python bench.py --train --train-dtype act16 --batch 8 --steps 6 --warmup 3 --no-pmc-leg
    python tools/train_gaps.py D/**/t_kernel_trace.csv
Steps are cut at adam_multi_kernel (one launch per step).  Per step (mean of the last ones): wall time, time with NO kernel resident,
busy time per queue, and -- on the queue that runs the BatchNorm forward (the step's main stream) -- the idle time in front of each
kernel kind (start - end of the previous kernel of that queue): launch latency of dependent kernels, or the host falling behind."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60]


def main():
    path = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])))
    rows.sort()
    cuts = [e for (s, e, q, n) in rows if n.startswith("adam_multi_kernel")]
    if len(cuts) < last + 1:
        print("only %d steps in the trace" % len(cuts)); return
    cuts = cuts[-(last + 1):]
    t0, t1 = cuts[0], cuts[-1]
    win = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1]
    wall = (t1 - t0) / last / 1e6
    main_q = collections.Counter(q for (s, e, q, n) in win if n.startswith("bn_act_fwd_kernel")).most_common(1)[0][0]
    # union busy time
    busy, cur_s, cur_e = 0, None, None
    for s, e, q, n in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("steps %d  wall %.2f ms/step  some kernel resident %.2f ms/step  nothing resident %.2f ms/step  launches %.0f/step" % (
        last, wall, busy / last / 1e6, wall - busy / last / 1e6, len(win) / last))
    perq = collections.defaultdict(lambda: [0, 0])
    for s, e, q, n in win:
        perq[q][0] += e - s; perq[q][1] += 1
    for q, (t, c) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
        print("  queue %-4s %8.2f ms/step of kernels  %6.0f launches/step%s" % (q, t / last / 1e6, c / last, "   <- main" if q == main_q else ""))
    # idle in front of each kernel kind on the main queue; and kernel time by kind while the other queue is busy / idle
    prev_e, idle, cnt, dur = None, collections.Counter(), collections.Counter(), collections.Counter()
    hist = collections.Counter()
    for s, e, q, n in win:
        if q != main_q:
            continue
        if prev_e is not None:
            g = max(0, s - prev_e)
            idle[n] += g; cnt[n] += 1
            hist[min(int(g / 1000), 50)] += g
        dur[n] += e - s
        prev_e = max(prev_e or 0, e)
    tot_idle = sum(idle.values()) / last / 1e6
    print("main queue: kernels %.2f ms/step, idle between kernels %.2f ms/step" % (sum(dur.values()) / last / 1e6, tot_idle))
    print("  idle by gap size (us: ms/step): " + "  ".join("%s%d: %.2f" % (">=" if k == 50 else "", k, v / last / 1e6) for k, v in sorted(hist.items()) if v / last / 1e6 >= 0.05))
    print("  %-62s %9s %9s %8s %8s" % ("kernel kind (main queue)", "ms/step", "idle ms", "n/step", "gap us"))
    for n, t in sorted(dur.items(), key=lambda kv: -(kv[1] + idle[kv[0]]))[:32]:
        print("  %-62s %9.3f %9.3f %8.1f %8.2f" % (n, t / last / 1e6, idle[n] / last / 1e6, cnt[n] / last, idle[n] / max(cnt[n], 1) / 1e3))
    oq = collections.Counter()
    for s, e, q, n in win:
        if q != main_q:
            oq[n] += e - s
    print("other queues:")
    for n, t in oq.most_common(12):
        print("  %-62s %9.3f" % (n, t / last / 1e6))


if __name__ == "__main__":
    main()
