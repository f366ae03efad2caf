#!/usr/bin/env python3
"""Diagnostic: the ping-pong conv kernels with their MFMA loop compiled out (-DISG_NO_MFMA, a separate
library under tools/_build) against the shipped ones: how much of a launch is the memory structure."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build", "libisg_nomfma.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(ROOT, "imagesegmentation_amd", "csrc", "*.hip")))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared",
                           "-DISG_NO_MFMA", "-o", OUT] + srcs)


def main():
    if "--build" in sys.argv:
        return build()
    import torch
    from imagesegmentation_amd import _lib
    if "--nomfma" in sys.argv:
        _lib.LIB_PATH = OUT
    if "--lib" in sys.argv:  # any other variant library (A/B of kernel changes)
        _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    import bench
    from imagesegmentation_amd.polyrnn import experiments
    dev = torch.device("cuda:0")
    model = experiments.make("RNNConv", image_size=224, compute_dtype=torch.bfloat16, device=dev, seed=0, head_size=28)
    batch = {"images": torch.zeros([64, 224, 224, 3], device=dev)}
    for k in bench.time_kernels(model, batch, "bf16"):
        print("%-28s %.3f ms" % (k["kernel"], k["ms_isolated"]))


if __name__ == "__main__":
    main()
