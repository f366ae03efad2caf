"""Diagnostic: does a process that used timing events (or big allocations) slow the process started right after it?"""
import sys
import torch
mode = sys.argv[1]
x = torch.randn([1 << 20], device="cuda")
if mode == "events":
    evs = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); x.add_(1); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    print("events", sum(a.elapsed_time(b) for a, b in evs))
elif mode == "alloc":
    big = [torch.empty([1 << 30], dtype=torch.uint8, device="cuda") for _ in range(40)]
    for t in big:
        t.zero_()
    torch.cuda.synchronize()
    print("alloc", len(big))
elif mode == "plain":
    for _ in range(200):
        x.add_(1)
    torch.cuda.synchronize()
    print("plain")
