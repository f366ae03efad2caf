#!/usr/bin/env python3
"""Diagnostic: does an actor-critic update ask the device for memory (the caching allocator's device allocations and
frees per update), and how long does the host spend inside loss.backward()?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from imagesegmentation_amd import data
    from imagesegmentation_amd.polyrl import a3c
    from imagesegmentation_amd.polyrl.batched_env import BatchedPolygonEnv
    dev = torch.device("cuda:0")
    n = int(os.environ.get("ISG_A3C_ENVS", "1024"))
    ds = data.PolygonDataset.synthetic(64, image_size=32, prediction_size=32, max_timesteps=10, history_length=2, seed=30)
    rs = np.random.RandomState(40)
    idx = rs.randint(0, len(ds), [n])
    images = ds._images[idx]
    polygons = [np.roll(ds._vertices[i], rs.randint(len(ds._vertices[i])), axis=0) for i in idx]
    env = BatchedPolygonEnv(ds, n, end_token=False, max_steps=a3c.A3C.TIMESTEP_LIMIT, device=dev)
    trainer = a3c.A3C(env, compute_dtype=torch.bfloat16, seed=0)
    for _ in range(3):
        trainer.process(images, polygons)
    torch.cuda.synchronize()
    for _ in range(4):
        s0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        trainer.process(images, polygons)
        torch.cuda.synchronize()
        s1 = torch.cuda.memory_stats()
        print("update %.2f ms  device allocs %d frees %d  reserved %.2f GB  allocated peak %.2f GB" % (
            (time.perf_counter() - t0) * 1e3, s1["num_device_alloc"] - s0["num_device_alloc"],
            s1["num_device_free"] - s0["num_device_free"], s1["reserved_bytes.all.current"] / 2 ** 30,
            s1["allocated_bytes.all.peak"] / 2 ** 30))
    # where the host's time goes: the reset alone, then the pieces of _process with a device drain after each
    t0 = time.perf_counter()
    env.reset(images=images, polygons=polygons)
    torch.cuda.synchronize()
    print("env.reset alone (host + device) %.2f ms" % ((time.perf_counter() - t0) * 1e3))


if __name__ == "__main__":
    main()
