#!/usr/bin/env python3
"""Diagnostic: the launches between the last two launches whose name contains <marker>, in start order, from a
rocprofv3 --kernel-trace CSV (one period of whatever repeats).  usage: trace_order.py <kernel_trace.csv> <marker>"""
import csv
import re
import sys


def main(path, marker):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    win = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = int(win[0]["Start_Timestamp"])
    for i, r in enumerate(win):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)[:110]
        print("%3d %9.3f %8.4f  %s" % (i, (int(r["Start_Timestamp"]) - t0) / 1e6,
                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, name))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
