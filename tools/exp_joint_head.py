#!/usr/bin/env python3
"""Experiment: the actor-critic update's two dense heads (1,024 + 1 columns over 32,768 inputs, 10,240 rows) as they
run now against one layer of 1,032 columns (the value column appended, padded to a multiple of 8)."""
import torch


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    M, K = 10240, 32768
    flat = torch.randn([M, K], device=dev).bfloat16()
    for N in (1024, 1032, 1040, 1152):
        w = torch.randn([K, N], device=dev).bfloat16() * 0.01
        b = torch.zeros([N], device=dev).bfloat16()
        d = torch.randn([M, N], device=dev).bfloat16()
        print("N=%d  fwd %.3f  dflat %.3f  dw %.3f ms" % (
            N, timed(lambda: torch.addmm(b, flat, w)), timed(lambda: d @ w.t()), timed(lambda: flat.t() @ d)))
    wv = torch.randn([K, 1], device=dev).bfloat16()
    dv = torch.randn([M, 1], device=dev).bfloat16()
    dflat = torch.randn([M, K], device=dev).bfloat16()
    print("value head: fwd %.3f  outer %.3f  dw %.3f  add %.3f  addmm_ in place %.3f ms" % (
        timed(lambda: flat @ wv), timed(lambda: dv @ wv.t()), timed(lambda: (dv.t() @ flat).t()),
        timed(lambda: dflat + dflat), timed(lambda: dflat.addmm_(dv, wv.t()))))
    w1 = torch.randn([K, 1024], device=dev).bfloat16()
    print("cat of the joint weights %.3f ms" % timed(lambda: torch.cat([w1, wv, w1[:, :7]], 1)))


if __name__ == "__main__":
    main()
