#!/usr/bin/env python3
"""Diagnostic: where the GPU idles inside one update of a rocprofv3 kernel trace - the launches between the last two
optimiser launches (adam_tf), the largest gaps between consecutive kernels and the idle time per 2 ms of the span.
usage: trace_gaps.py <kernel_trace.csv> [how many]"""
import collections
import csv
import re
import sys


def main(path, top=15):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adam_tf" in r["Kernel_Name"]]
    seg = rows[ad[-2] + 1:ad[-1] + 1]
    short = lambda n: re.sub(r"void at::native::|\(anonymous namespace\)::", "", n)[:70]
    t0 = int(seg[0]["Start_Timestamp"])
    prev = int(rows[ad[-2]]["End_Timestamp"])
    gaps = []
    for i, r in enumerate(seg):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gaps.append((s - prev, i, (s - t0) / 1e6, short(r["Kernel_Name"])))
        prev = max(prev, e)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("%d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms" %
          (len(seg), (prev - t0) / 1e6, busy / 1e6, sum(g[0] for g in gaps if g[0] > 0) / 1e6))
    for g in sorted(gaps, reverse=True)[:top]:
        print("%9.1f us before #%d at %6.2f ms  %s" % (g[0] / 1e3, g[1], g[2], g[3]))
    reg = collections.Counter()
    for g in gaps:
        if g[0] > 0:
            reg[int(g[2] // 2) * 2] += g[0]
    print("idle ms per 2 ms of the span:", sorted((k, round(v / 1e6, 2)) for k, v in reg.items()))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 15)
