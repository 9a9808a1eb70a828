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


_STATS_GROUP = {}


def stats_group():
    """A second process group (its own RCCL communicator, hence its own collective stream) for the Sync-BN moment
    all-reduces.  They sit ON the critical path of the step -- a layer cannot be normalised before its global statistics
    exist -- while the gradient buckets of GradReducer are in flight beside it: on one communicator a 2*C-double all-reduce
    issued behind a 28 MB bucket waits for the whole bucket's wire time (collectives of one communicator run in issue
    order).  Created collectively the first time any rank needs it: every rank reaches its first Sync-BN layer at the same
    point of the same program.  None for a single process."""
    _, size = world()
    if size == 1:
        return None
    key = id(dist.group.WORLD)
    if key not in _STATS_GROUP:
        _STATS_GROUP.clear()                   # (a process group of an earlier init_process_group is gone)
        _STATS_GROUP[key] = dist.new_group(backend=dist.get_backend())
    return _STATS_GROUP[key]


def allreduce_stats_(*tensors):
    """In-place sum over ranks on stats_group() (the Sync-BN moments: 2*C doubles per layer and direction)."""
    group = stats_group()
    if group is not None:
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors


def broadcast_(*tensors, src=0):
    """In-place broadcast from rank `src` (no-op for a single process)."""
    _, size = world()
    if size > 1:
        for t in tensors:
            dist.broadcast(t, src=src)
    return tensors


def require_equal_batchsize(batchsize, device):
    """Sync-BN sums per-rank moments and divides by local count x world size (ops._sync_sums), and the gradient
    all-reduce weights every rank's images equally: both need the SAME number of images per rank and step.  `fit`
    always feeds exactly config['batchsize'] samples per step, so it is enough that the ranks agree on it."""
    _, size = world()
    if size == 1:
        return
    t = torch.tensor([int(batchsize), -int(batchsize)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0]) != -int(t[1]):
        raise ValueError('data-parallel training needs the same batchsize on every rank (got %d .. %d)'
                         % (-int(t[1]), int(t[0])))


def sync_trainer_from_rank0(trainer):
    """Data-parallel replicas must start from ONE set of parameters: the reference trains a single graph
    (base_model.py:153-162), so N ranks reproduce it only if rank 0's master weights, batch-norm moving
    statistics and optimizer slots are everybody's.  Called when a trainer is created and after every
    `load_from_variables` (random initialisers are drawn per process, imported files may differ per rank).  Returns True
    if a broadcast took place: the caller's variable dict / folded inference weights are then stale on ranks != 0."""
    _, size = world()
    if size == 1:
        return False
    broadcast_(trainer.param)
    for mm, mv in getattr(trainer, 'moving', {}).values():
        broadcast_(mm, mv)
    # optimizer slots exist on a rank only once it has stepped: agree on who has them before broadcasting
    has = torch.tensor([1 if trainer.state else 0, int(getattr(trainer, 't', 0))], dtype=torch.int64,
                       device=trainer.param.device)
    broadcast_(has)
    if int(has[0]):
        if not trainer.state:
            from .trainer import init_optimizer_state
            init_optimizer_state(trainer)
        for key in sorted(trainer.state):
            broadcast_(trainer.state[key])
    else:
        trainer.state = {}
    trainer.t = int(has[1])
    trainer.repack()
    return True


def agree_any(flag, device):
    """True on every rank if `flag` is true on any rank (early-abort decisions must be collective, otherwise the
    ranks that keep training hang in the next gradient all-reduce)."""
    _, size = world()
    if size == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


class GradReducer(object):
    """Bucketed gradient all-reduce on a side HIP stream, overlapped with the rest of backward.

    `launch(flat, (b, e))` is called by the trainer as soon as the last filter gradient of a bucket has
    been enqueued: an event on the compute stream orders the collective after those kernels, the
    collective itself runs on the side stream, and `wait()` joins before the optimizer.  Every rank
    differentiates the loss normalised by the GLOBAL number of labelled pixels (all-reduced count), so
    summing the per-rank gradients reproduces the single-device gradient of the whole batch: no
    division by world size.  xGMI is point-to-point (7 links per GPU): three buckets of 10-30 MB keep
    each collective bandwidth-bound rather than latency-bound."""

    def __init__(self, device):
        self.device = device
        self.side = torch.cuda.Stream(device=device) if torch.device(device).type == 'cuda' else None
        self.pending = []

    def allreduce_now(self, tensor):
        _, size = world()
        if size > 1:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)

    def launch(self, flat, rng):
        _, size = world()
        if size == 1:
            return
        view = flat[rng[0]:rng[1]]
        if self.side is None:
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        for work in self.pending:
            work.wait()
        self.pending = []
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
