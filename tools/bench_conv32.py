#!/usr/bin/env python3
"""Diagnostic: the square encoder convolutions (forward, masked data gradient, weight gradient) launched alone at
the benchmark shape (B=64, 224 x 224).
ISG_LIBRARY selects a variant library.  usage: bench_conv32.py [repeats]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from imagesegmentation_amd import _lib
    if os.environ.get("ISG_LIBRARY"):
        _lib.LIB_PATH = os.environ["ISG_LIBRARY"]
    from imagesegmentation_amd import ops
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    out = []
    for c in (32, 16):
        x = torch.randn([64, 224, 224, c], device=dev).to(torch.bfloat16)
        for ks in (5, 3):
            w = ops.conv32_pack(torch.randn([ks, ks, c, c], device=dev) * 0.05)
            b = torch.zeros([c], device=dev)
            for _ in range(3):
                ops.conv32(x, w, b, ks, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.conv32(x, w, b, ks, True)
            e1.record()
            torch.cuda.synchronize()
            t_fwd = e0.elapsed_time(e1) / n
            wd = ops.conv32_pack(torch.randn([ks, ks, c, c], device=dev) * 0.05, data_gradient=True)
            y = torch.relu(torch.randn([64, 224, 224, c], device=dev)).to(torch.bfloat16)
            times = []
            for fn in (lambda: ops.conv32_masked(x, wd, y, ks), lambda: ops.conv32_wgrad(x, y, ks)):
                for _ in range(3):
                    fn()
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / n)
            out.append("C=%d %dx%d fwd %.4f masked dgrad %.4f wgrad %.4f ms" % (c, ks, ks, t_fwd, times[0], times[1]))
    # the first layer: 3 -> 32 channels, 14 x 14 taps
    img = torch.randn([64, 224, 224, 3], device=dev).to(torch.bfloat16)
    x4 = ops.conv_in_pad_image(img)
    w1 = torch.randn([14, 14, 3, 32], device=dev) * 0.05
    wimg = ops.conv_in_pack(w1)
    b1 = torch.zeros([32], device=dev)
    dz = torch.randn([64, 224, 224, 32], device=dev).to(torch.bfloat16)
    times = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for fn in (lambda: ops.conv_in_fwd(x4, wimg, b1, 14, relu=True), lambda: ops.conv_in_wgrad(x4, dz, 14)):
        for _ in range(3):
            fn()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / n)
    out.append("first layer 14x14 fwd %.4f wgrad %.4f ms" % tuple(times))
    print(os.environ.get("ISG_LIBRARY", "default"), " | ".join(out))


if __name__ == "__main__":
    main()
