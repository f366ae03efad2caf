#!/usr/bin/env python3
"""Diagnostic: kernels of ONE greedy decode + IoU (EndTokenDeep_2, 28 x 28, 10 vertices, B from ISG_DECODE_B, default
16).  Without arguments: warm-up + three decodes, each bracketed by a raster launch (wrap in rocprofv3 --kernel-trace);
with a trace CSV: the launches of the last decode in order, with durations and gaps.  ISG_NO_DECODE_GRAPH=1: the
loop launched step by step."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrnn import experiments
    dev = torch.device("cuda:0")
    B = int(os.environ.get("ISG_DECODE_B", "16"))
    model = experiments.make("EndTokenDeep_2", image_size=28, compute_dtype=torch.bfloat16, device=dev, seed=0)
    model.decode_graphs = not os.environ.get("ISG_NO_DECODE_GRAPH")
    ds = data.PolygonDataset.synthetic(64, 28, 28, model.max_timesteps, model.history_length, seed=50, end_token=True)
    idx = np.random.RandomState(60).randint(0, len(ds), [B])
    images = torch.as_tensor(ds._images[idx], device=dev).to(torch.bfloat16)
    verts = [ds._vertices[i] for i in idx]
    for _ in range(6):
        model.validate_iou(images, verts, summary_prefix=None)
    torch.cuda.synchronize()


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "raster_iou" in r["Kernel_Name"]]
    seg = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("one decode + IoU: %d kernels, span %.3f ms, busy %.3f ms" %
          (len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e6, busy / 1e6))
    prev = t0
    for i, r in enumerate(seg):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%3d  +%7.2f us  gap %6.2f  dur %6.2f  %s" % (i, (s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3,
                                                         r["Kernel_Name"][:100]))
        prev = e


if __name__ == "__main__":
    if len(sys.argv) > 1:
        summarize(sys.argv[1])
    else:
        run()
