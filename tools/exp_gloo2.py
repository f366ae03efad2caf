"""Diagnostic: two gloo ranks sharing one GPU all-reduce a CUDA tensor a few times (is the rehearsal set-up itself alive?)."""
import os, time, sys
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
x = torch.ones([n], device="cuda")
for i in range(4):
    t0 = time.perf_counter()
    y = torch.randn([4096, 4096], device="cuda") @ torch.randn([4096, 4096], device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "iter", i, "%.3f s" % (time.perf_counter() - t0), float(x[0]), flush=True)
dist.barrier()
