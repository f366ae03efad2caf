#!/usr/bin/env python3
"""Diagnostic: per-kernel totals of the LAST actor-critic update in a rocprofv3 kernel trace of tools/prof_a3c.py
(the launches between the last two optimiser launches; a rollout replayed as a graph shows no kernels).
usage: a3c_update_summary.py <kernel_trace.csv>"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adam_tf" in r["Kernel_Name"]]
    seg = rows[ad[-2] + 1:ad[-1] + 1]
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        k = r["Kernel_Name"][:120]
        agg[k][0] += dur(r)
        agg[k][1] += 1
    print("one update: span %.3f ms, kernels %d, sum of kernel time %.3f ms" %
          ((t1 - t0) / 1e6, len(seg), sum(dur(r) for r in seg) / 1e6))
    for k, (t, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:45]:
        print("%8.3f %5d %8.4f  %s" % (t / 1e6, n, t / 1e6 / n, k))


if __name__ == "__main__":
    main(sys.argv[1])
