"""Diagnostic: wall time of each of the first training steps of a fresh process (sync after every step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from imagesegmentation_amd.polyrnn import experiments
dev = torch.device("cuda:0")
model = experiments.make("RNNConv", image_size=224, compute_dtype=torch.bfloat16, device=dev, seed=0, head_size=28)
batch = bench.make_batch(224, 64, model.max_timesteps, model.history_length, 28, seed=0, device=dev, dtype=torch.bfloat16)
ts = []
for i in range(40):
    t0 = time.perf_counter()
    model.train_step(batch["images"], batch["durations"], batch["points"], batch["targets"])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print("host-issue ms / then wait ms per step:", " ".join("%.0f/%.0f" % t for t in ts))
