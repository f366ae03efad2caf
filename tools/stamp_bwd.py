#!/usr/bin/env python3
"""Diagnostic: where the fused backward kernel spends its cycles, from in-kernel s_memtime stamps.

Builds a SEPARATE library (tools/_build/libisg_stamps.so, -DISG_STAMPS; the shipped libisg.so contains no stamps),
runs the fused backward step at the benchmark shape for both layer shapes and prints the share of each phase for
waves 0 (gate backward first) and 7 (phase B first), median over workgroups.  The fences around the stamps forbid overlaps the real kernel has: read the
shares, not the total.   usage: stamp_bwd.py --build (on the CPU box), then stamp_bwd.py on the GPU box."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stamp_conv  # noqa: E402  (shares the stamped build)


def main():
    if "--build" in sys.argv:
        stamp_conv.build()
        return
    import numpy as np
    import torch
    from imagesegmentation_amd import _lib
    _lib.LIB_PATH = stamp_conv.OUT
    from imagesegmentation_amd import ops
    lib = _lib.lib()
    lib.isg_debug_set_bwd_stamp_buffer.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    B, P, F, dt = 64, 224, 16, torch.bfloat16
    names = ["store phase ([x | h_prev] tile to LDS)", "barrier after the store phase", "gate backward of the next tile",
             "dgrad k-loop + stores", "wgrad k-loop", "barrier closing the interval", "total", "tiles"]
    for cs, c_real, hist in ((32, 33, 1), (16, 16, 0)):
        w_x = torch.randn([3, 3, c_real, 64], device=dev) * 0.05
        w_h = torch.randn([3, 3, F, 64], device=dev) * 0.05
        packed = ops.PackedWeights(w_x, w_h, torch.zeros([64], device=dev), cs, dt, hist=hist)
        pts = torch.randint(0, P, [B, hist, 2], device=dev, dtype=torch.int32) if hist else None
        x = torch.randn([B, P, P, cs], device=dev).to(dt)
        h = (torch.randn([B, P, P, F], device=dev) * 0.5).to(dt)
        c0, c1 = torch.randn([B, P, P, F], device=dev), torch.randn([B, P, P, F], device=dev)
        gates = torch.rand([B, P, P, F, 4], device=dev).to(dt)
        factors = torch.rand([B, P, P, F // 2, 4, 2], device=dev).to(dt)
        above = torch.randn([B, P, P, F], device=dev).to(dt)
        dh, dc = torch.randn([B, P, P, F], device=dev).to(dt), torch.randn([B, P, P, F], device=dev).to(dt)
        dh2, dc2 = torch.empty_like(dh), torch.empty_like(dc)
        dx = torch.zeros([B, P, P, cs if cs == 32 else F], dtype=dt, device=dev)
        ws = ops.wgrad_workspace(cs, F, dev, cached=False)
        dwx = torch.zeros_like(w_x)
        fn = lambda: ops.bwd_fused(factors, c0, h, above, P * P * F, dh, dc, x, h, packed, dc2, dh2, ws, ops.WGRAD_DEFER,
                                   dwx, dx=dx, dx_channels=dx.shape[-1], dx_accumulate=cs == 32, hist_points=pts)
        buf = torch.zeros([256 * 2 * 8], dtype=torch.int64, device=dev)
        lib.isg_debug_set_bwd_stamp_buffer(buf.data_ptr())
        fn()
        torch.cuda.synchronize()
        buf.zero_()
        fn()
        torch.cuda.synchronize()
        a = buf.cpu().numpy().reshape(256, 2, 8)
        print("== fused backward, Cs=%d" % cs)
        for g, who in ((0, "wave 0"), (1, "wave 7")):
            med = np.median(a[:, g, :], axis=0)
            print("  %s: total %8.0f cycles over %d tiles (%.0f per tile)" % (who, med[6], med[7], med[6] / med[7]))
            for k in range(6):
                print("     %-40s %8.0f  %5.1f %%  (%.0f per tile)" % (names[k], med[k], 100 * med[k] / med[6], med[k] / med[7]))


if __name__ == "__main__":
    main()
