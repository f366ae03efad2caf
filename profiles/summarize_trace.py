#!/usr/bin/env python3
"""Per-kernel time inside ONE training step of bench.py, from a rocprofv3 --kernel-trace CSV.

usage: summarize_trace.py <kernel_trace.csv>
A step is the span between two consecutive xent_bwd_kernel dispatches (one per training step); the
last full step of the trace is used, so MIOpen's first-call solver search in the warm-up is excluded.
"""
import collections
import csv
import re
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "xent_bwd" in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]
    win = rows[a:b]
    t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
    agg = collections.OrderedDict()
    for r in win:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:100]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        e = agg.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += d
    total = sum(v[1] for v in agg.values())
    print("one training step: wall %.3f ms, kernels %d, sum of kernel time %.3f ms" % ((t1 - t0) / 1e6, len(win), total))
    print("%10s %6s %10s  %s" % ("total_ms", "calls", "avg_ms", "kernel"))
    for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%10.3f %6d %10.4f  %s" % (d, c, d / c, name))


if __name__ == "__main__":
    main(sys.argv[1])
