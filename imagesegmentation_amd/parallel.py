"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).

Both hot paths shard by sample: teacher-forced batches and REINFORCE rollouts are split across
ranks, every rank holds the full weights, and the only exchange per step is ONE all-reduce of the
flat float32 gradient bucket (the reference's closest analogue is the parameter-server push/pull of
polyrl/a3c/a3c.py:251-258).  The rasteriser / IoU path has no exchange at all.
"""
import os

import torch
import torch.distributed as dist


def _collectives_on(group=None):
    """Whether the gradient exchange is issued: more than one rank - or ISG_FORCE_COLLECTIVES=1 with an initialised
    process group of ONE rank, which sends every collective of the multi-rank path through the library (a
    one-rank all-reduce is the identity): the way to run the RCCL calls, the early-tail hook and their stream
    ordering on a machine with a single GPU."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or bool(os.environ.get("ISG_FORCE_COLLECTIVES"))


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def sampling_seed(base, group=None):
    """Seed of a rank's own random draws (sampled actions): base * world + rank, distinct per rank for every
    base, and equal to ``base`` for a single process.  Weights are initialised from the un-folded seed (every
    rank must hold the same ones); rollouts must NOT share their noise across ranks, or the all-reduced policy
    gradient is an average over correlated samples."""
    return int(base) * world_size(group) + rank(group)


def shard_range(n, rank_, world):
    """Contiguous [begin, end) of n samples owned by rank_: sizes differ by at most one."""
    base, extra = divmod(n, world)
    begin = rank_ * base + min(rank_, extra)
    return begin, begin + base + (1 if rank_ < extra else 0)


def allreduce_gradients_(flat, group=None, average=True):
    """In-place sum (or mean) of the flat gradient bucket over all ranks; a no-op for one rank.

    ``average=True`` matches a loss that is a mean over the GLOBAL batch when every rank holds an equal
    share (reference polyrnn/Model.py:98-100); ``average=False`` matches gradients that the reference
    sums over episodes (polyrl/ReinforceOverfitExperiment.py:168-176)."""
    if _collectives_on(group):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world_size(group))
    return flat


class EarlyTailReduce:
    """The same all-reduce in two pieces, so that most of it hides behind the backward pass.

    ``flat`` is the gradient bucket, ``tail_params`` the parameters whose gradients are views of
    ``flat[offset:]`` and are complete EARLY in a backward pass (the dense head: 99 % of the bytes, first to
    finish).  While ``arm()``-ed, the moment the last of them has its gradient the tail is all-reduced
    asynchronously (RCCL on its own stream, next to the recurrence's backward kernels); ``finish()`` after
    the backward reduces the remaining head of the bucket, waits for the tail and averages.  Every rank
    issues the two collectives in the same order.  Not armed, or with one rank, nothing happens early and
    ``finish()`` is ``allreduce_gradients_``."""

    def __init__(self, flat, tail_params, offset, group=None):
        self.flat, self.offset, self.group = flat, int(offset), group
        self._n = len(tail_params)
        self._seen = 0
        self._work = None
        self.armed = False
        self.early_launches = 0  # how many times the tail went out from the hook (tests, diagnostics)
        for p in tail_params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def arm(self):
        self.armed, self._seen, self._work = True, 0, None

    def _on_grad(self, _param):
        if not self.armed or not _collectives_on(self.group):
            return
        self._seen += 1
        if self._seen == self._n:
            self._work = dist.all_reduce(self.flat[self.offset:], op=dist.ReduceOp.SUM, group=self.group,
                                         async_op=True)
            self.early_launches += 1

    def finish(self, average=True):
        was_armed, self.armed = self.armed, False
        work, self._work = self._work, None
        world = world_size(self.group)
        if not _collectives_on(self.group):
            return self.flat
        if not was_armed or work is None:
            return allreduce_gradients_(self.flat, self.group, average)
        if self.offset > 0:
            dist.all_reduce(self.flat[:self.offset], op=dist.ReduceOp.SUM, group=self.group)
        work.wait()
        if average:
            self.flat.div_(world)
        return self.flat


def allreduce_max(value, device, group=None):
    """Max of a python float over ranks (timing)."""
    if not _collectives_on(group):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
