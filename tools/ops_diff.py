#!/usr/bin/env python
"""Per-layer diff of two `bench.py --ops-json` tables (same plan, two library variants): where did the time move?
Usage: python tools/ops_diff.py A.json B.json [--top 25]"""
import collections
import json
import sys


def table(path):
    t = collections.OrderedDict()
    for o in json.load(open(path)):
        e = t.setdefault((o["kind"], o["label"]), [0, 0.0])
        e[0] += 1; e[1] += o["ms"]
    return t


def main():
    a, b = table(sys.argv[1]), table(sys.argv[2])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    rows = []
    for k in a:
        if k in b:
            rows.append((b[k][1] - a[k][1], k, a[k], b[k]))
    ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
    print("total: A %.3f ms  B %.3f ms  (B - A = %+.3f ms)" % (ta, tb, tb - ta))
    for d, (kind, label), ea, eb in sorted(rows, key=lambda r: -abs(r[0]))[:top]:
        print("%+8.3f ms  %-8s %-44s n=%3d  A %7.1f us  B %7.1f us each" % (d, kind, label[:44], ea[0], 1e3 * ea[1] / ea[0], 1e3 * eb[1] / eb[0]))


if __name__ == "__main__":
    main()
