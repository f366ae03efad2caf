#!/usr/bin/env python3
"""Experiment: can two RCCL ranks share the box's one GPU?  (launch: python -m torch.distributed.run --nnodes=1
--nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/exp_rccl2.py)"""
import os

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl")
t = torch.full([1024], float(rank + 1), device="cuda:0")
dist.all_reduce(t)
torch.cuda.synchronize()
print("rank", rank, "sum", float(t[0]), flush=True)
dist.destroy_process_group()
