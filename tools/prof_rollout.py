#!/usr/bin/env python3
"""Diagnostic: kernel breakdown of one REINFORCE rollout (2,048 end-token environments at 28 x 28, T = 10).

Without arguments: runs 1 warm-up + 3 rollouts (to be wrapped in rocprofv3 --kernel-trace).
With a kernel-trace CSV: prints per-kernel totals of the LAST rollout (span between the last two
raster launches)."""
import collections
import csv
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl.reinforce import BatchedReinforce, ExperimentPolicyEstimator
    dev = torch.device("cuda:0")
    N = 2048
    ds = data.PolygonDataset.synthetic(64, image_size=28, prediction_size=28, max_timesteps=10, history_length=2,
                                       seed=1, end_token=True)
    pol = ExperimentPolicyEstimator(28, 28, max_timesteps=10, end_token=True, compute_dtype=torch.bfloat16, device=dev)
    runner = BatchedReinforce(pol, ds, N, device_pool=True)
    fn = runner.step if "--update" in sys.argv else runner.rollout  # --update: rollout + policy update
    fn()
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        print("%s ms %.3f" % (fn.__name__, (time.perf_counter() - t0) * 1e3))


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "raster" in r["Kernel_Name"]]
    a, b = marks[-2] + 1, marks[-1] + 1
    win = rows[a:b]
    t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
    agg = collections.OrderedDict()
    for r in win:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:110]
        e = agg.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    total = sum(v[1] for v in agg.values())
    print("one rollout: wall %.3f ms, kernels %d, sum of kernel time %.3f ms" % ((t1 - t0) / 1e6, len(win), total))
    for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%10.3f %6d %10.4f  %s" % (d, c, d / c, name))


if __name__ == "__main__":
    summarize(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else run()
