#!/usr/bin/env python3
"""Diagnostic: per-kernel totals of the launches between two consecutive launches whose name contains <marker>, taken
at <fraction> of the way through a rocprofv3 --kernel-trace CSV.  usage: trace_window.py <csv> <marker> <fraction>"""
import collections
import csv
import re
import sys


def main(path, marker, frac):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    k = int(len(marks) * frac)
    w = rows[marks[k] + 1:marks[k + 1] + 1]
    t0, t1 = int(w[0]["Start_Timestamp"]), int(w[-1]["End_Timestamp"])
    agg = collections.OrderedDict()
    for r in w:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:100]
        e = agg.setdefault(n, [0, 0.0])
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("window: %d kernels, span %.1f us, busy %.1f us" % (len(w), (t1 - t0) / 1e3, sum(v[1] for v in agg.values())))
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print("  %8.1f %4d %7.2f  %s" % (d, c, d / c, n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]))
