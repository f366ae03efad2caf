"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).

Both hot paths shard by sample: teacher-forced batches and REINFORCE rollouts are split across
ranks, every rank holds the full weights, and the only exchange per step is ONE all-reduce of the
flat float32 gradient bucket (the reference's closest analogue is the parameter-server push/pull of
polyrl/a3c/a3c.py:251-258).  The rasteriser / IoU path has no exchange at all.
"""
import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


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
    world = world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
    return flat


def allreduce_max(value, device, group=None):
    """Max of a python float over ranks (timing)."""
    if world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
