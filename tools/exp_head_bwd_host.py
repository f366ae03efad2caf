#!/usr/bin/env python3
"""Diagnostic: host time of the value head's weight-gradient product at the actor-critic update's shapes
([32768, 10240] x [10240, 1]) in several spellings (no device drain inside: what the launching thread spends)."""
import time

import torch


def main():
    d = "cuda:0"
    R, K = 10240, 32768
    flat = torch.randn([R, K], device=d).to(torch.bfloat16)
    dl = torch.randn([R, 1], device=d).to(torch.bfloat16)
    ref = (flat.float().t() @ dl.float())
    forms = {
        "flat.t() @ dl": lambda: flat.t() @ dl,
        "(dl.t() @ flat).t()": lambda: (dl.t() @ flat).t(),
        "flat.t() @ pad8(dl)": lambda: (flat.t() @ torch.nn.functional.pad(dl, (0, 7)))[:, :1],
        "mv(flat.t(), dl)": lambda: torch.mv(flat.t(), dl[:, 0])[:, None],
    }
    for name, fn in forms.items():
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        print("%-24s host %.3f ms, drain %.3f ms, rel err %.1e" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, err))
    # the forward products of the narrow head
    wc = (torch.randn([K, 1], device=d) * 0.01).to(torch.bfloat16)
    b = torch.zeros([1], device=d, dtype=torch.bfloat16)
    for name, fn in {"addmm(b, flat, w)": lambda: torch.addmm(b, flat, wc), "dl @ w.t()": lambda: dl @ wc.t()}.items():
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print("%-24s host %.3f ms, drain %.3f ms" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3))


if __name__ == "__main__":
    main()
