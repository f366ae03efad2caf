#!/usr/bin/env python3
"""Diagnostic: where a ping-pong conv kernel spends its cycles, from in-kernel s_memtime stamps.

Builds a SEPARATE library (tools/_build/libisg_stamps.so, -DISG_STAMPS) - the shipped libisg.so
contains no stamps - runs the forward cell and the data gradient at the benchmark shape and prints the
share of each phase per group (median over workgroups).  Never quote this build's run time: the
fences around the stamps forbid overlaps the real kernel has; read the shares.
"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.environ.get("ISG_STAMP_LIBRARY") or os.path.join(ROOT, "tools", "_build", "libisg_stamps.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(ROOT, "imagesegmentation_amd", "csrc", "*.hip")))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared",
                           "-DISG_STAMPS", "-o", OUT] + srcs)


def main():
    if "--build" in sys.argv:
        build()
        return
    import numpy as np
    import torch
    from imagesegmentation_amd import _lib
    _lib.LIB_PATH = OUT
    from imagesegmentation_amd import ops
    from imagesegmentation_amd.convLSTM import ConvLSTMCell
    lib = _lib.lib()
    lib.isg_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    B, P, F = 64, 224, 16
    if "--rollout-shape" in sys.argv:  # 2,048 environments at 28 x 28, no gate images (decode / rollout steps)
        B, P = 2048, 28
    keep_gates = "--rollout-shape" not in sys.argv
    names = ["issue_prefetch", "k_loop", "stage_store(wait+lds)", "epilogue", "barrier_after_compute",
             "barrier_after_finish", "total", "half_steps"]
    for C, hist in ((33, 1), (16, 0)):
        cell = ConvLSTMCell([P, P], 16, [3, 3], compute_dtype=torch.bfloat16, device=dev).build(C, dev)
        packed = cell.packed(hist=hist)
        cs = packed.cs
        x = torch.randn([B, P, P, cs], device=dev).bfloat16()
        h = torch.randn([B, P, P, F], device=dev).bfloat16()
        c = torch.randn([B, P, P, F], device=dev).bfloat16()
        pts = torch.randint(0, P, [B, hist, 2], device=dev, dtype=torch.int32) if hist else None
        gates = torch.empty([B, P, P, F, 4], dtype=torch.bfloat16, device=dev)
        dy = torch.randn([B, P, P, F, 4], device=dev).bfloat16()
        dx = torch.zeros([B, P, P, cs], dtype=torch.bfloat16, device=dev)
        dhp = torch.empty_like(h)
        dwx = torch.zeros_like(cell.input_to_state_kernel)
        dwh = torch.zeros_like(cell.hidden_to_state_kernel)
        db = torch.zeros_like(cell.bias)
        # wgrad rows: "group" 0 = wave 0, 1 = wave 7; phases: load issue, k-loop, wait + LDS writes, -, barrier
        for label, fn in (("cell_fwd C=%d" % C, lambda: ops.cell_fwd(x, h, c, packed, gates=gates if keep_gates else None,
                                                                     save_gates=keep_gates, hist_points=pts,
                                                                     save_layout=ops.SAVE_FACTORS)),
                          ("dgrad C=%d" % C, lambda: ops.dgrad(dy, packed, dhp, dx=dx, dx_channels=cs)),
                          ("wgrad C=%d" % C, lambda: ops.wgrad(x, h, dy, packed.c, dwx, dwh, db, False,
                                                               hist_points=pts, hist=hist))):
            buf = torch.zeros([256 * 2 * 8], dtype=torch.int64, device=dev)
            lib.isg_debug_set_stamp_buffer(buf.data_ptr())
            fn()
            torch.cuda.synchronize()
            buf.zero_()
            fn()
            torch.cuda.synchronize()
            a = buf.cpu().numpy().reshape(256, 2, 8)
            print("== %s" % label)
            for g in (0, 1):
                med = np.median(a[:, g, :], axis=0)
                tot = med[6]
                print("  group %d: total %8.0f cycles over %d half-steps (%.0f per half-step)" % (g, tot, med[7], tot / med[7]))
                for k in range(6):
                    print("     %-24s %8.0f  %5.1f %%" % (names[k], med[k], 100 * med[k] / tot))


if __name__ == "__main__":
    main()
