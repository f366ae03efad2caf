"""The N > 1 path on CPU: two gloo ranks, sharded batch, one all-reduce of the flat gradient bucket.
Checks that the data-parallel step reproduces the single-process gradient (mean loss -> averaged
gradients; REINFORCE -> summed gradients) and that rollout shards cover the work exactly once."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat_grads(tp):
    out = []
    for p in tp["encoder"]:
        out += [p["w"].grad, p["b"].grad]
    for p in tp["rnn"]:
        out += [p["w_x"].grad, p["w_h"].grad, p["bias"].grad]
    out += [tp["dense"]["w"].grad, tp["dense"]["b"].grad]
    return torch.cat([g.reshape(-1) for g in out])


def _problem():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers_convlstm import make_batch, make_params
    rs = np.random.RandomState(0)
    P, hist, T, B = 8, 2, 4, 6
    params = make_params(rs, P, hist, encoder=((4, 3),), layers=2, F_=4, units=P * P + 1)
    images, histories, durations, targets = make_batch(rs, B, T, P, hist, end_token=True)
    weights = rs.uniform(0, 1, [B, T])
    return params, images, histories, durations, targets, weights, P


def _loss(tp, images, histories, durations, targets, P, weights, reduce):
    from oracle import torch_port
    logits, _ = torch_port.model_logits(tp, torch.tensor(images), torch.tensor(histories), torch.tensor(durations))
    B, T, K = logits.shape
    labels = torch.tensor(targets[..., 0].astype(np.int64) * P + targets[..., 1])
    rows = torch.nn.functional.cross_entropy(logits.reshape(B * T, K), labels.reshape(-1), reduction="none")
    rows = rows.reshape(B, T)
    if weights is not None:
        rows = rows * torch.tensor(weights)
    return rows.mean() if reduce == "mean" else rows.mean(dim=1).sum()


def _rank_grads(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return [torch.randn(3, 4, generator=g) * 30, torch.randn(5, generator=g) * 30]


def _adam_case(grad_sets, all_reduce):
    """Two steps of the flat-bucket Adam (actor-critic trainer) with clip 40 on the mean of grad_sets."""
    from imagesegmentation_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(7)
    params = [torch.nn.Parameter(torch.randn(3, 4, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g))]
    opt = FlatAdam(params, learning_rate=1e-2, clip_norm=40.0)
    for _ in range(2):
        opt.zero_grad()
        for grads in grad_sets:
            for p, gr in zip(params, grads):
                p.grad += gr / len(grad_sets)
        opt.step(all_reduce=all_reduce)
    return torch.cat([p.detach().reshape(-1) for p in params]).numpy()


def _early_tail_case(rank, reduce=True):
    """A two-layer toy net whose LAST layer closes the flat bucket: armed, its gradient is reduced from the
    autograd hook; the result must be the plain average over ranks (and a second, unarmed pass the plain sum)."""
    from imagesegmentation_amd import parallel
    g = torch.Generator().manual_seed(3)
    a = torch.nn.Parameter(torch.randn(5, 4, generator=g))
    b = torch.nn.Parameter(torch.randn(4, 3, generator=g))
    c = torch.nn.Parameter(torch.randn(3, generator=g))
    flat = torch.zeros([a.numel() + b.numel() + c.numel()])
    off = 0
    for p in (a, b, c):
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    early = parallel.EarlyTailReduce(flat, [b, c], a.numel())
    x = torch.randn(7, 5, generator=torch.Generator().manual_seed(50 + rank))
    outs = []
    for armed, average in ((True, True), (False, False)):
        flat.zero_()
        if armed:
            early.arm()
        (torch.tanh(x @ a) @ b + c).square().sum().backward()
        if reduce:
            early.finish(average=average)
        outs.append(flat.clone().numpy())
    assert early.early_launches == (1 if reduce and dist.is_initialized() else 0)
    return outs


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers_convlstm import params_to_torch
    from imagesegmentation_amd import parallel
    params, images, histories, durations, targets, weights, P = _problem()
    lo, hi = parallel.shard_range(len(images), rank, world)
    out = {}
    for mode in ("mean", "episode_sum"):
        tp = params_to_torch(params)
        w = None if mode == "mean" else weights[lo:hi]
        _loss(tp, images[lo:hi], histories[lo:hi], durations[lo:hi], targets[lo:hi], P, w, mode).backward()
        flat = _flat_grads(tp)
        parallel.allreduce_gradients_(flat, average=(mode == "mean"))
        out[mode] = flat.numpy()
    out["early"] = _early_tail_case(rank)
    out["max"] = parallel.allreduce_max(float(rank + 1), "cpu")
    out["seeds"] = [parallel.sampling_seed(b) for b in (0, 1, 7)]
    out["adam"] = _adam_case([_rank_grads(rank)], all_reduce=True)
    ret[rank] = out
    dist.destroy_process_group()


def test_two_rank_gradients_match_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers_convlstm import params_to_torch
    params, images, histories, durations, targets, weights, P = _problem()
    ref = {}
    for mode in ("mean", "episode_sum"):
        tp = params_to_torch(params)
        _loss(tp, images, histories, durations, targets, P, None if mode == "mean" else weights, mode).backward()
        ref[mode] = _flat_grads(tp).numpy()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    for rank in (0, 1):
        for mode in ("mean", "episode_sum"):
            assert np.allclose(ret[rank][mode], ref[mode], rtol=1e-10, atol=1e-12), (rank, mode)
        locals_ = [_early_tail_case(r, reduce=False) for r in (0, 1)]
        assert np.allclose(ret[rank]["early"][0], (locals_[0][0] + locals_[1][0]) / 2, rtol=1e-6, atol=1e-6)
        assert np.allclose(ret[rank]["early"][1], locals_[0][1] + locals_[1][1], rtol=1e-6, atol=1e-6)
        assert ret[rank]["max"] == 2.0
    # per-rank action noise: no two (base seed, rank) pairs share a sampling seed
    assert ret[0]["seeds"] == [0, 2, 14] and ret[1]["seeds"] == [1, 3, 15]
    for rank in (0, 1):
        assert np.allclose(ret[rank]["adam"], _adam_case([_rank_grads(0), _rank_grads(1)], all_reduce=False),
                           rtol=1e-6, atol=1e-7)


def test_shard_range_partitions_exactly():
    from imagesegmentation_amd import parallel
    assert [parallel.sampling_seed(b) for b in (0, 5)] == [0, 5]  # one process: the seed itself
    for n in (0, 1, 7, 64, 16384, 16385):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
