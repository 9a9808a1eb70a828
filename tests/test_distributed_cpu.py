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
