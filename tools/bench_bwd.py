#!/usr/bin/env python3
"""Diagnostic: the fused backward step alone at the benchmark shape (B=64, 224x224, bf16), both layer shapes,
against the three-kernel sequence on the same tensors.  ISG_LIBRARY selects a variant build of the library.
usage: python tools/bench_bwd.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagesegmentation_amd import ops  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    B, P, F = 64, 224, 16
    dt = torch.bfloat16
    out = []
    for cs, c_real, hist in ((32, 33, 1), (16, 16, 0)):
        w_x = torch.randn([3, 3, c_real, 64], device=dev) * 0.05
        w_h = torch.randn([3, 3, F, 64], device=dev) * 0.05
        packed = ops.PackedWeights(w_x, w_h, torch.zeros([64], device=dev), cs, dt, hist=hist)
        pts = torch.randint(0, P, [B, hist, 2], device=dev, dtype=torch.int32) if hist else None
        x = torch.randn([B, P, P, cs], device=dev).to(dt)
        h = (torch.randn([B, P, P, F], device=dev) * 0.5).to(dt)
        c0, c1 = torch.randn([B, P, P, F], device=dev).to(dt), torch.randn([B, P, P, F], device=dev)
        c0f = c0.float()
        gates = torch.rand([B, P, P, F, 4], device=dev).to(dt)
        factors = torch.rand([B, P, P, F // 2, 4, 2], device=dev).to(dt)
        dh_above = torch.randn([B, P, P, F], device=dev).to(dt)
        dh, dc = torch.randn([B, P, P, F], device=dev).to(dt), torch.randn([B, P, P, F], device=dev).to(dt)
        dh2, dc2 = torch.empty_like(dh), torch.empty_like(dc)
        dx = torch.zeros([B, P, P, cs if cs == 32 else F], dtype=dt, device=dev)
        dy = torch.empty_like(gates)
        ws = ops.wgrad_workspace(cs, F, dev, cached=False)
        dwx, dwh, db = torch.zeros_like(w_x), torch.zeros_like(w_h), torch.zeros([64], device=dev)
        acc = cs == 32
        fused = lambda: ops.bwd_fused(factors, c0, h, dh_above, P * P * F, dh, dc, x, h, packed, dc2, dh2, ws,
                                      ops.WGRAD_DEFER, dwx, dx=dx, dx_channels=dx.shape[-1], dx_accumulate=acc,
                                      hist_points=pts)

        def three():
            ops.pointwise_bwd(gates, c0f, c1, dh_above, P * P * F, dh, dc, dy=dy, dc_prev=dc2)
            ops.wgrad(x, h, dy, packed.c, dwx, dwh, db, True, hist_points=pts, hist=hist, mode=ops.WGRAD_DEFER, workspace=ws)
            ops.dgrad(dy, packed, dh2, dx=dx, dx_channels=dx.shape[-1], dx_accumulate=acc)

        out.append("Cs=%d fused %.4f ms  three-kernel %.4f ms" % (cs, timed(fused, reps), timed(three, reps)))
    print(os.environ.get("ISG_LIBRARY", "default"), " | ".join(out))


if __name__ == "__main__":
    main()
