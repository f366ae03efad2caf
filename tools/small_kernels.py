#!/usr/bin/env python3
"""Diagnostic: which torch operators launch the small (< 30 us) kernels of one training step at the benchmark
shape.  Runs the step under torch.profiler and prints, per operator, the number of launches and their summed device
time (14 add_ = autograd accumulating into the views of the flat gradient; copy_ = dtype casts and layout copies
around the sequence node).  With --sites: the call sites in this package of every copy / cast / add / fill operator
of the step instead (a dispatch-mode trace; autograd's own accumulation runs outside Python and shows no site).
usage (GPU box): python tools/small_kernels.py [--sites]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from imagesegmentation_amd.polyrnn import experiments
    dt = torch.bfloat16
    model = experiments.make("RNNConv", image_size=224, compute_dtype=dt, device=dev, seed=0, head_size=28)
    batch = bench.make_batch(224, 64, model.max_timesteps, model.history_length, 28, seed=0, device=dev, dtype=dt)
    step = lambda: model.train_step(batch["images"], batch["durations"], batch["points"], batch["targets"])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if "--sites" in sys.argv:
        import traceback
        from torch.utils._python_dispatch import TorchDispatchMode
        seen = collections.Counter()

        class Sites(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                name = str(func)
                if any(k in name for k in ("copy", "add", "fill", "zero", "div", "mul", "flip", "sum", "cat", "clone")):
                    site = "?"
                    for fr in reversed(traceback.extract_stack()):
                        if "imagesegmentation_amd" in fr.filename:
                            site = "%s:%d" % (fr.filename.replace(ROOT + "/", ""), fr.lineno)
                            break
                    shape = tuple(args[0].shape) if args and hasattr(args[0], "shape") else ()
                    seen[(name, site, shape)] += 1
                return func(*args, **(kwargs or {}))

        with Sites():
            step()
        torch.cuda.synchronize()
        for (name, site, shape), n in sorted(seen.items(), key=lambda kv: kv[0][1]):
            print("%2d x %-34s %-52s %s" % (n, name, site, shape))
        return
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        ktime = sum(k.duration for k in ev.kernels)
        if ktime <= 0 or ktime / max(len(ev.kernels), 1) > 30:
            continue
        rows[ev.name][0] += len(ev.kernels)
        rows[ev.name][1] += ktime
    total = 0.0
    for name, (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%7.1f us %3d launches  %s" % (us, n, name[:60]))
        total += us
    print("total %.1f us" % total)


if __name__ == "__main__":
    main()
