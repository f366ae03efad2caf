"""Diagnostic: the bench's five-layer recurrence leg alone (EndTokenDeep_2's ConvLSTM stack at 224 x 224)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

w = bench.GpuWorkload.__new__(bench.GpuWorkload)
w.device, w.dtype, w.rank = torch.device("cuda:0"), torch.bfloat16, 0
print(w.recurrence_leg())
