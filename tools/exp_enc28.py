"""Experiment: library convolution time for the 28x28 REINFORCE encoder (N=2048) in several forms."""
import sys
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = "--find" in sys.argv
dev = torch.device("cuda:0")
N, S = 2048, 28


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for cin, cout, k in ((3, 16, 5), (16, 16, 5), (16, 16, 3)):
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        for cl in (True, False):
            x = torch.randn([N, cin, S, S], device=dev).to(dt)
            w = torch.randn([cout, cin, k, k], device=dev).to(dt)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last)
                w = w.contiguous(memory_format=torch.channels_last)
            t = timeit(lambda: F.conv2d(x, w, None, padding=k // 2))
            print("cin %2d cout %2d k %d %-8s %-13s %.3f ms" % (cin, cout, k, str(dt).split(".")[1],
                                                                "channels_last" if cl else "nchw", t), flush=True)
