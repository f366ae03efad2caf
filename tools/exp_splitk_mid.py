#!/usr/bin/env python3
"""Diagnostic: a rollout step's head at 2,048 rows, [2048, 12544] x [12544, 785] in bf16, as one library GEMM and as
batched GEMMs over slices of the reduction axis with float partial results + the summing launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagesegmentation_amd import _lib, ops  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    d = "cuda:0"
    for M in (2048, 512, 128):
        K, N = 12544, 785
        a = (torch.randn([M, K], device=d) * 0.5).to(torch.bfloat16)
        w = (torch.randn([K, N], device=d) * 0.01).to(torch.bfloat16)
        b = torch.randn([N], device=d).to(torch.bfloat16)
        out = ["M=%d addmm %.4f ms" % (M, timed(lambda: torch.addmm(b, a, w)))]
        for S in (2, 4, 8):
            a3, w3 = a.view(M, S, K // S).transpose(0, 1), w.view(S, K // S, N)
            o = torch.empty([M, N], dtype=torch.bfloat16, device=d)

            def f():
                part = torch.bmm(a3, w3, out_dtype=torch.float32)
                _lib.lib().isg_sum_slices_bias(part.data_ptr(), S, M, N, b.data_ptr(), ops.dtype_code(o.dtype), o.data_ptr(),
                                               _lib.current_stream())
            out.append("S=%d %.4f" % (S, timed(f)))
        print(" | ".join(out))


if __name__ == "__main__":
    main()
