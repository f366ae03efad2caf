"""Diagnostic: decode latency (validate_iou end to end) with the loop as a graph and launched step by step."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

class A: pass
a = A(); a.size = 28; a.config = "endtokendeep2"; a.dtype = "bf16"
w = bench.GpuWorkload.__new__(bench.GpuWorkload)
w.device = torch.device("cuda:0"); w.dtype = torch.bfloat16; w.rank = 0
print(w.decode_leg())
