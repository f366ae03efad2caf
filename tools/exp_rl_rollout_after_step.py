#!/usr/bin/env python3
"""Diagnostic: time of REINFORCE rollouts that follow an update step (the order of the bench leg)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from imagesegmentation_amd import data  # noqa: E402
from imagesegmentation_amd.polyrl.reinforce import BatchedReinforce, ExperimentPolicyEstimator  # noqa: E402

dev = torch.device("cuda:0")
ds = data.PolygonDataset.synthetic(64, image_size=28, prediction_size=28, max_timesteps=10, history_length=2, seed=10,
                                   end_token=True)
pol = ExperimentPolicyEstimator(28, 28, max_timesteps=10, end_token=True, compute_dtype=torch.bfloat16, device=dev, seed=0)
runner = BatchedReinforce(pol, ds, 2048, device_pool=True, pool_seed=20)
runner.step()
torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter()
    runner.rollout()
    torch.cuda.synchronize()
    print("rollout %d: %.2f ms, reserved %.1f GB" % (i, (time.perf_counter() - t0) * 1e3,
                                                    torch.cuda.memory_reserved() / 2 ** 30))
for i in range(3):
    t0 = time.perf_counter()
    runner.step()
    torch.cuda.synchronize()
    print("step %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
