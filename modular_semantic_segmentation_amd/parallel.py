"""One process per GPU (torch.distributed; backend 'nccl' = RCCL over xGMI on the GPU box,
'gloo' in CPU tests).  The hot path shards by image: inference and scoring need NO data-path
collective, only the tiny result reductions below (a [C,C] confusion matrix, the Dirichlet
sufficient statistics); expert training adds the gradient all-reduce."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n, rank=None, size=None):
    """Contiguous, balanced [begin, end) share of n items for this rank."""
    if rank is None:
        rank, size = world()
    q, r = divmod(n, size)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard_data(data, rank=None, size=None):
    """This rank's share of a dict-of-arrays dataset (leading axis = samples)."""
    n = len(next(iter(data.values())))
    b, e = shard_range(n, rank, size)
    return {k: v[b:e] for k, v in data.items()}


def allreduce_sum_(*tensors):
    """In-place sum over ranks (no-op for a single process).  int64 / float64 tensors of a few KB:
    one collective each, latency-bound, nothing to bucket."""
    _, size = world()
    if size > 1:
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tensors
