#!/usr/bin/env python
"""Aggregate a bench.py --ops-json file by layer label: time, launches, TFLOP/s and algorithmic GB/s per group."""
import collections
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    agg = collections.OrderedDict()
    for o in d:
        a = agg.setdefault(o["label"], [0, 0.0, 0, 0])
        a[0] += 1; a[1] += o["ms"]; a[2] += o["flops"]; a[3] += o["bytes"]
    print("total %.2f ms in %d launches" % (sum(o["ms"] for o in d), len(d)))
    for k, (n, ms, f, b) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%6.2f ms x%3d %6.0f TF/s %6.0f GB/s  %s" % (ms, n, f / ms / 1e9 if ms else 0, b / ms / 1e6 if ms else 0, k))


if __name__ == "__main__":
    main()
