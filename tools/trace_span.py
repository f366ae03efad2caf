#!/usr/bin/env python3
"""Diagnostic: per-kernel totals of a rocprofv3 --kernel-trace CSV between the last two launches whose name contains
<marker> (one period of whatever repeats).  usage: trace_span.py <kernel_trace.csv> <marker> [rows]"""
import collections
import csv
import re
import sys


def main(path, marker, top=30):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    win = rows[marks[-2] + 1:marks[-1] + 1]
    t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
    agg = collections.OrderedDict()
    for r in win:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:120]
        e = agg.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    total = sum(v[1] for v in agg.values())
    print("span: wall %.3f ms, kernels %d, sum of kernel time %.3f ms" % ((t1 - t0) / 1e6, len(win), total))
    for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%10.3f %6d %10.4f  %s" % (d, c, d / c, name))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 30)
