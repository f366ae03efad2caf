"""The multi-rank gradient exchange on the GPU box's one card: a process group of ONE rank over the "nccl" (= RCCL)
backend with ISG_FORCE_COLLECTIVES=1 sends every collective of the data-parallel step through the library - the
early-tail all-reduce fired from the autograd hook onto RCCL's stream, the remainder after the backward, the wait and
the averaging - and must leave loss, gradients and the Adam step bit for bit what the step without a process group
gives (a one-rank all-reduce is the identity).  Runs in a child process: the process group is global state."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, %r)
from imagesegmentation_amd import data, parallel
from imagesegmentation_amd.polyrnn import experiments
forced = os.environ.get("ISG_FORCE_COLLECTIVES") == "1"
if forced:
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
ds = data.PolygonDataset.synthetic(16, 28, 28, 10, 2, seed=1)
model = experiments.make("History2Large", image_size=28, compute_dtype=torch.bfloat16, device="cuda:0", seed=3)
model.step_graphs = os.environ.get("ISG_TEST_STEP_GRAPHS") == "1"
np.random.seed(0)
d, images, pts, t, _ = ds.get_batch_points(8)
losses = [model.train_step(images, d, pts, t).item() for _ in range(3)]
torch.cuda.synchronize()
flat = torch.cat([p.detach().float().reshape(-1) for p in model.trainable_variables])
out = dict(losses=losses, checksum=float(flat.double().sum()), absum=float(flat.double().abs().sum()),
           early=int(model._early_reduce.early_launches), on=bool(parallel._collectives_on()),
           graphs=len(model._step_graph_cache))
# the actor-critic trainer: acting loop replayed as a graph, one-piece all-reduce inside FlatAdam.step
from imagesegmentation_amd.polyrl import a3c
from imagesegmentation_amd.polyrl.batched_env import BatchedPolygonEnv
ds32 = data.PolygonDataset.synthetic(16, image_size=32, prediction_size=32, max_timesteps=10, history_length=2, seed=30)
rs = np.random.RandomState(40)
idx = rs.randint(0, len(ds32), [16])
polys = [np.roll(ds32._vertices[i], rs.randint(len(ds32._vertices[i])), axis=0) for i in idx]
env = BatchedPolygonEnv(ds32, 16, end_token=False, max_steps=10, device=torch.device("cuda:0"))
trainer = a3c.A3C(env, compute_dtype=torch.bfloat16, seed=0)
for _ in range(3):
    trainer.process(ds32._images[idx], polys)
torch.cuda.synchronize()
out["a3c"] = float(trainer.opt.flat_param.double().sum())
out["a3c_graph"] = trainer._rollout["graph"] is not None
if forced:
    # the REINFORCE / actor-critic optimiser's one-piece all-reduce, and the timing reduction of bench.py
    g = torch.arange(1000, dtype=torch.float32, device="cuda:0")
    parallel.allreduce_gradients_(g, average=True)
    out["one_piece"] = float(g.sum())
    out["max"] = parallel.allreduce_max(1.5, torch.device("cuda:0"))
    dist.destroy_process_group()
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(forced, step_graphs=False):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ISG_FORCE_COLLECTIVES", None)
    env.pop("ISG_TEST_STEP_GRAPHS", None)
    if forced:
        env["ISG_FORCE_COLLECTIVES"] = "1"
    if step_graphs:
        env["ISG_TEST_STEP_GRAPHS"] = "1"
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_one_rank_rccl_group_runs_every_collective_and_changes_nothing():
    plain = _run(False)
    forced = _run(True)
    assert not plain["on"] and plain["early"] == 0
    assert forced["on"] and forced["early"] == 3, "the head's share must go out from the autograd hook in every step"
    assert forced["losses"] == plain["losses"]
    assert forced["checksum"] == plain["checksum"] and forced["absum"] == plain["absum"]
    assert forced["one_piece"] == float(sum(range(1000))) and forced["max"] == 1.5
    assert forced["a3c"] == plain["a3c"] and forced["a3c_graph"] and plain["a3c_graph"]
    # the training step replayed as a graph: the exchange (one piece, after the replay), clip and Adam stay outside
    graphed = _run(True, step_graphs=True)
    assert graphed["graphs"] >= 1 and graphed["early"] == 0
    assert graphed["losses"] == plain["losses"] and graphed["checksum"] == plain["checksum"]
