#!/usr/bin/env python3
"""Diagnostic: idle gaps between consecutive kernels of the last training step in a rocprofv3 kernel trace
(usage: step_gaps.py <kernel_trace.csv> [min_gap_us] [marker]); a step = the span between the last two launches
whose name contains the marker (default xent_bwd: one per training step)."""
import csv
import sys


def main(path, min_gap=8.0, marker="xent_bwd"):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    win = rows[marks[-2]:marks[-1] + 1]
    total = 0.0
    for i in range(1, len(win)):
        g = (int(win[i]["Start_Timestamp"]) - int(win[i - 1]["End_Timestamp"])) / 1e3
        total += max(g, 0.0)
        if g > min_gap:
            print("%7.1f us before #%d %s (after %s)" % (g, i, win[i]["Kernel_Name"][:56], win[i - 1]["Kernel_Name"][:44]))
    span = (int(win[-1]["Start_Timestamp"]) - int(win[0]["Start_Timestamp"])) / 1e6
    print("step %.3f ms, %d kernels, idle %.3f ms" % (span, len(win) - 1, total / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0, sys.argv[3] if len(sys.argv) > 3 else "xent_bwd")
