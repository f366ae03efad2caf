#!/usr/bin/env python3
"""Diagnostic: the acting step's action head, [1024, 32768] x [32768, 1024] in bf16, as one library GEMM and as a
batched GEMM over slices of the reduction axis summed afterwards."""
import torch


def timed(fn, n=20):
    for _ in range(3):
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
    M, K, N = 1024, 32768, 1024
    a = (torch.randn([M, K], device=d) * 0.5).to(torch.bfloat16)
    w = (torch.randn([K, N], device=d) * 0.01).to(torch.bfloat16)
    b = torch.randn([N], device=d).to(torch.bfloat16)
    ref = torch.addmm(b.float(), a.float(), w.float())
    print("addmm %.4f ms" % timed(lambda: torch.addmm(b, a, w)), float((torch.addmm(b, a, w).float() - ref).abs().max()))
    for S in (4, 8, 16, 32):
        a3 = a.view(M, S, K // S).transpose(0, 1)
        w3 = w.view(S, K // S, N)
        f_bf = lambda: torch.bmm(a3, w3).float().sum(0).add_(b)
        print("bmm S=%d bf16 partials %.4f ms" % (S, timed(f_bf)), float((f_bf() - ref).abs().max()))
        try:
            f_32 = lambda: torch.bmm(a3, w3, out_dtype=torch.float32).sum(0).add_(b)
            print("bmm S=%d f32 partials %.4f ms" % (S, timed(f_32)), float((f_32() - ref).abs().max()))
        except Exception as err:
            print("out_dtype:", str(err)[:100])


if __name__ == "__main__":
    main()
