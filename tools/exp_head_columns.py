#!/usr/bin/env python3
"""Experiment: the REINFORCE update's dense head (20,480 rows, 12,544 inputs, 785 = 28 x 28 + 1 classes) against the
same layer with its class axis padded to a multiple of 8: forward, input gradient and weight gradient GEMMs."""
import torch

from exp_joint_head import timed


def main():
    dev = torch.device("cuda:0")
    M, K = 20480, 12544
    flat = torch.randn([M, K], device=dev).bfloat16()
    for N in (785, 792, 800, 832):
        w = torch.randn([K, N], device=dev).bfloat16() * 0.01
        b = torch.zeros([N], device=dev).bfloat16()
        d = torch.randn([M, N], device=dev).bfloat16()
        print("N=%d  fwd %.3f  dflat %.3f  dw %.3f ms" % (
            N, timed(lambda: torch.addmm(b, flat, w)), timed(lambda: d @ w.t()), timed(lambda: flat.t() @ d)))


if __name__ == "__main__":
    main()
