"""N > 1 path on CPU: two gloo processes shard a dataset by image, reduce the per-shard
confusion matrix / Dirichlet sufficient statistics, and must reproduce the single-process result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fcn_oracle as fo
from oracle import fusion_oracle as fu

C = 6


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dataset():
    rng = np.random.default_rng(0)
    n = 5
    return {'prob': fo.softmax(rng.standard_normal((n, 8, 9, C)).astype(np.float32)),
            'pred': rng.integers(0, C, (n, 8, 9)).astype(np.int64),
            'labels': rng.integers(-1, C, (n, 8, 9)).astype(np.int32)}


def _worker(rank, size, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    from modular_semantic_segmentation_amd import parallel
    data = parallel.shard_data(_dataset())
    cm = torch.from_numpy(fu.confusion_matrix(data['labels'], data['pred'], C).astype(np.int64))
    S, counts = fu.sufficient_statistics(data['prob'], data['labels'], C)
    S, counts = torch.from_numpy(S), torch.from_numpy(counts)
    parallel.allreduce_sum_(cm, S, counts)
    if rank == 0:
        np.savez(out, cm=cm.numpy(), S=S.numpy(), counts=counts.numpy())
    dist.destroy_process_group()


def test_two_rank_reduction_matches_single_process(tmp_path):
    out = str(tmp_path / 'r.npz')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    full = _dataset()
    assert np.array_equal(got['cm'], fu.confusion_matrix(full['labels'], full['pred'], C))
    S, counts = fu.sufficient_statistics(full['prob'], full['labels'], C)
    assert np.array_equal(got['counts'], counts)
    np.testing.assert_allclose(got['S'], S, rtol=1e-12, atol=1e-9)


def test_four_rank_reduction_with_uneven_shards(tmp_path):
    """World size 4 over 5 samples (shards of 2, 1, 1, 1 images: the remainder path of shard_range): the reduced confusion
    matrix / sufficient statistics equal the single-process ones."""
    out = str(tmp_path / 'r4.npz')
    mp.spawn(_worker, args=(4, _free_port(), out), nprocs=4, join=True)
    got = np.load(out)
    full = _dataset()
    assert np.array_equal(got['cm'], fu.confusion_matrix(full['labels'], full['pred'], C))
    S, counts = fu.sufficient_statistics(full['prob'], full['labels'], C)
    assert np.array_equal(got['counts'], counts)
    np.testing.assert_allclose(got['S'], S, rtol=1e-12, atol=1e-9)


def test_shard_range_is_a_partition():
    from modular_semantic_segmentation_amd import parallel
    for n in (0, 1, 7, 16):
        for size in (1, 2, 3, 8):
            cover = []
            for r in range(size):
                b, e = parallel.shard_range(n, r, size)
                cover += list(range(b, e))
            assert cover == list(range(n))


def _grad_worker(rank, size, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    from modular_semantic_segmentation_amd import parallel
    rng = np.random.default_rng(100 + rank)
    flat = torch.from_numpy(rng.standard_normal(1000).astype(np.float32))
    count = torch.tensor([10 + rank], dtype=torch.int64)
    red = parallel.GradReducer('cpu')
    red.allreduce_now(count)
    for rng_ in ((0, 300), (300, 640), (640, 1000)):       # buckets become ready one after the other
        red.launch(flat, rng_)
    red.wait()
    if rank == 0:
        np.savez(out, flat=flat.numpy(), count=count.numpy())
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce(tmp_path):
    out = str(tmp_path / 'g.npz')
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    ref = sum(np.random.default_rng(100 + r).standard_normal(1000).astype(np.float32) for r in range(2))
    np.testing.assert_allclose(got['flat'], ref, rtol=1e-6)
    assert got['count'][0] == 21


def test_bucketed_gradient_allreduce_four_ranks(tmp_path):
    """The same three buckets over four ranks (the first time more than two ranks meet in the bucket path), plus the
    collective early-abort vote and the equal-batchsize check of parallel.py."""
    out = str(tmp_path / 'g4.npz')
    mp.spawn(_grad_worker, args=(4, _free_port(), out), nprocs=4, join=True)
    got = np.load(out)
    ref = sum(np.random.default_rng(100 + r).standard_normal(1000).astype(np.float32) for r in range(4))
    np.testing.assert_allclose(got['flat'], ref, rtol=1e-5, atol=1e-6)
    assert got['count'][0] == 10 + 11 + 12 + 13


def _stats_worker(rank, size, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    from modular_semantic_segmentation_amd import parallel
    # the Sync-BN moments travel on a process group of their own (parallel.stats_group), interleaved with gradient buckets on
    # the default one, in the order FcnBnTrainer.step issues them: moments, bucket, moments, bucket, ...
    flat = torch.full((600,), float(rank + 1))
    red = parallel.GradReducer('cpu')
    sums = [torch.arange(8, dtype=torch.float64) * (rank + 1) for _ in range(3)]
    for i, rng_ in enumerate(((0, 100), (100, 350), (350, 600))):
        parallel.allreduce_stats_(sums[i])
        red.launch(flat, rng_)
    red.wait()
    group = parallel.stats_group()
    same = parallel.stats_group() is group and group is not dist.group.WORLD
    if rank == 0:
        np.savez(out, flat=flat.numpy(), sums=torch.stack(sums).numpy(), same=same)
    dist.destroy_process_group()


def test_sync_bn_moments_on_their_own_process_group(tmp_path):
    """parallel.allreduce_stats_ (what ops._sync_sums calls per batch norm and direction) next to the gradient buckets: three
    ranks, both groups in use alternately, every sum complete; one process: no group, nothing to do."""
    from modular_semantic_segmentation_amd import parallel
    assert parallel.stats_group() is None
    t = torch.ones(4, dtype=torch.float64)
    assert parallel.allreduce_stats_(t)[0] is t and float(t.sum()) == 4.0
    out = str(tmp_path / 'st.npz')
    mp.spawn(_stats_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out)
    assert bool(got['same'])
    assert np.array_equal(got['flat'], np.full(600, 6.0, np.float32))
    assert np.array_equal(got['sums'], np.tile(np.arange(8, dtype=np.float64) * 6, (3, 1)))


def _vote_worker(rank, size, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    from modular_semantic_segmentation_amd import parallel
    any_failed = parallel.agree_any(rank == 2, 'cpu')          # one rank fails: every rank must learn it
    none_failed = parallel.agree_any(False, 'cpu')
    parallel.require_equal_batchsize(8, 'cpu')
    try:
        parallel.require_equal_batchsize(8 if rank != 3 else 7, 'cpu')
        unequal = False
    except ValueError:
        unequal = True
    res = torch.tensor([int(any_failed), int(none_failed), int(unequal)], dtype=torch.int64)
    gathered = [torch.zeros(3, dtype=torch.int64) for _ in range(size)]
    dist.all_gather(gathered, res)
    if rank == 0:
        np.save(out, torch.stack(gathered).numpy())
    dist.destroy_process_group()


def test_collective_votes_over_four_ranks(tmp_path):
    out = str(tmp_path / 'v.npy')
    mp.spawn(_vote_worker, args=(4, _free_port(), out), nprocs=4, join=True)
    got = np.load(out)
    assert got.shape == (4, 3) and np.all(got[:, 0] == 1) and np.all(got[:, 1] == 0) and np.all(got[:, 2] == 1)


class _StubTrainer(object):
    """The attributes parallel.sync_trainer_from_rank0 touches (trainer.FcnTrainer and friends have the same)."""

    def __init__(self, seed, stepped):
        rng = np.random.default_rng(seed)
        self.kind = 'adam'
        self.param = torch.from_numpy(rng.standard_normal(257).astype(np.float32))
        self.moving = {'conv1_1': (torch.from_numpy(rng.standard_normal(8).astype(np.float32)),
                                   torch.from_numpy(rng.random(8).astype(np.float32)))}
        self.state, self.t, self.repacked = {}, 0, 0
        if stepped:
            self.state = {'m': torch.from_numpy(rng.standard_normal(257).astype(np.float32)),
                          'v': torch.from_numpy(rng.random(257).astype(np.float32))}
            self.t = 3

    def repack(self):
        self.repacked += 1


def _sync_worker(rank, size, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    from modular_semantic_segmentation_amd import parallel
    # every rank draws its own initialisers (seed=None in the models); rank 0 alone has optimizer slots (a resumed run)
    tr = _StubTrainer(seed=10 + rank, stepped=(rank == 0))
    synced = parallel.sync_trainer_from_rank0(tr)
    fresh = _StubTrainer(seed=20 + rank, stepped=False)
    parallel.sync_trainer_from_rank0(fresh)
    stop = parallel.agree_any(rank == 1, 'cpu')              # only rank 1 crossed abort_at_iou
    go_on = parallel.agree_any(False, 'cpu')
    parallel.require_equal_batchsize(4, 'cpu')
    try:
        parallel.require_equal_batchsize(4 + rank, 'cpu')
        unequal = False
    except ValueError:
        unequal = True
    np.savez(out % rank, param=tr.param.numpy(), mm=tr.moving['conv1_1'][0].numpy(), mv=tr.moving['conv1_1'][1].numpy(),
             m=tr.state['m'].numpy(), v=tr.state['v'].numpy(), t=tr.t, repacked=tr.repacked,
             fresh=fresh.param.numpy(), fresh_state=len(fresh.state), synced=synced, stop=stop, go_on=go_on, unequal=unequal)
    dist.destroy_process_group()


def test_replicas_start_from_rank0_parameters(tmp_path):
    """without a broadcast every rank trains its own Glorot draw and the replicas diverge."""
    out = str(tmp_path / 's%d.npz')
    mp.spawn(_sync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    src = _StubTrainer(seed=10, stepped=True)
    for key, ref in (('param', src.param), ('mm', src.moving['conv1_1'][0]), ('mv', src.moving['conv1_1'][1]),
                     ('m', src.state['m']), ('v', src.state['v'])):
        assert np.array_equal(r0[key], ref.numpy()) and np.array_equal(r1[key], ref.numpy()), key
    assert int(r0['t']) == int(r1['t']) == 3 and int(r1['repacked']) == 1
    assert np.array_equal(r1['fresh'], _StubTrainer(seed=20, stepped=False).param.numpy())
    assert int(r0['fresh_state']) == int(r1['fresh_state']) == 0
    assert bool(r0['stop']) and bool(r1['stop']) and not bool(r0['go_on']) and not bool(r1['go_on'])
    assert bool(r0['unequal']) and bool(r1['unequal'])
    # a broadcast is reported, and the models mark their variable dict / folded engine stale after one
    assert bool(r0['synced']) and bool(r1['synced'])


def test_rank0_sync_marks_model_stale():
    from modular_semantic_segmentation_amd import parallel
    from modular_semantic_segmentation_amd.adapnet import Adapnet
    from modular_semantic_segmentation_amd.fusion_fcn import FusionFCN
    from modular_semantic_segmentation_amd.simple_fcn import SimpleFCN
    assert parallel.sync_trainer_from_rank0(_StubTrainer(seed=1, stepped=False)) is False      # one process: nothing to do
    for cls in (SimpleFCN, Adapnet, FusionFCN):
        class Probe(object):
            _dirty = False
        p = Probe()
        cls._after_rank0_sync(p, False)
        assert p._dirty is False
        cls._after_rank0_sync(p, True)
        assert p._dirty is True
