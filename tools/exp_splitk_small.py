#!/usr/bin/env python3
"""Diagnostic: the decode step's head, [16, 12544] x [12544, 785] in bf16, as one library GEMM and as batched GEMMs
over slices of the reduction axis (run under rocprofv3 --kernel-trace --stats: the kernels' own durations)."""
import torch

d = "cuda:0"
M, K, N = 16, 12544, 785
a = (torch.randn([M, K], device=d) * 0.5).to(torch.bfloat16)
w = (torch.randn([K, N], device=d) * 0.01).to(torch.bfloat16)
b = torch.randn([N], device=d).to(torch.bfloat16)
for _ in range(20):
    torch.addmm(b, a, w)
for S in (4, 8, 16, 32):
    a3 = a.view(M, S, K // S).transpose(0, 1)
    w3 = w.view(S, K // S, N)
    for _ in range(20):
        p = torch.bmm(a3, w3, out_dtype=torch.float32)
    torch.cuda.synchronize()
    print(S, float((p.sum(0) + b.float() - torch.addmm(b.float(), a.float(), w.float())).abs().max()))
torch.cuda.synchronize()
