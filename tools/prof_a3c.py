#!/usr/bin/env python3
"""Diagnostic: kernel breakdown of one synchronous actor-critic update (256 environments at 32 x 32, F = 32).
Without arguments: 1 warm-up + 3 updates (wrap in rocprofv3 --kernel-trace); with a trace CSV: per-kernel
totals and idle time of the last update (span between the last two raster launches)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl import a3c
    from imagesegmentation_amd.polyrl.batched_env import BatchedPolygonEnv
    dev = torch.device("cuda:0")
    n = int(os.environ.get("ISG_A3C_ENVS", "256"))
    ds = data.PolygonDataset.synthetic(64, image_size=32, prediction_size=32, max_timesteps=10, history_length=2, seed=30)
    rs = np.random.RandomState(40)
    idx = rs.randint(0, len(ds), [n])
    images = ds._images[idx]
    polygons = [np.roll(ds._vertices[i], rs.randint(len(ds._vertices[i])), axis=0) for i in idx]
    env = BatchedPolygonEnv(ds, n, end_token=False, max_steps=a3c.A3C.TIMESTEP_LIMIT, device=dev)
    trainer = a3c.A3C(env, compute_dtype=torch.bfloat16, seed=0)
    if os.environ.get("ISG_A3C_POOL"):  # episodes drawn on the device, as the bench leg does
        from imagesegmentation_amd.polyrl.batched_env import DevicePolygonPool
        pool = DevicePolygonPool(ds, dev, dtype=torch.float32, vcap=2 + a3c.A3C.TIMESTEP_LIMIT + 1, seed=41)
        update = lambda: trainer.process(device_batch=pool.sample(n))
    else:
        update = lambda: trainer.process(images, polygons)
    update()
    torch.cuda.synchronize()
    times = []
    for _ in range(int(os.environ.get("ISG_A3C_UPDATES", "3"))):
        t0 = time.perf_counter()
        update()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        print("update ms %.3f" % times[-1])
    if len(times) > 4:
        tail = sorted(times[2:])
        print("median of the last %d: %.3f ms" % (len(tail), tail[len(tail) // 2]))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        import prof_rollout
        prof_rollout.summarize(sys.argv[1])
    else:
        run()
