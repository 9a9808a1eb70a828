import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from modular_semantic_segmentation_amd import host_pipeline as hp
from modular_semantic_segmentation_amd.base_model import iterate_batches
dev = torch.device('cuda', 0)
bs, n = 16, 384
net = bench.build_model(dev, 'bayes', 'fcn', bs, 'bf16')
rng = np.random.default_rng(0)
data = {'rgb': rng.integers(0, 256, (n, 384, 768, 3)).astype(np.float32),
        'depth': rng.integers(0, 65536, (n, 384, 768, 1)).astype(np.float32)}
net.predict({k: v[:64] for k, v in data.items()})     # warm + capture
torch.cuda.synchronize()
def run(variant):
    T = {'wait_next': 0.0, 'predict': 0.0, 'push': 0.0}
    hp.TRACE = {}
    t0 = time.perf_counter()
    fetch = hp.ResultFetcher(dev, n, narrow_labels=True)
    evs = []
    state = {}
    it = iter(net._device_batches(iterate_batches(data, bs), labels=False))
    while True:
        a = time.perf_counter()
        try:
            batch = next(it)
        except StopIteration:
            break
        b = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        out = net._predict_batch_auto(batch, None, state)
        c = time.perf_counter()
        if variant == 'full':
            fetch.push(out)
        elif variant == 'nocollect':
            pass
        d = time.perf_counter()
        e1 = torch.cuda.Event(enable_timing=True); e1.record(); evs.append((e0, e1))
        T['wait_next'] += b - a; T['predict'] += c - b; T['push'] += d - c
    if variant == 'full':
        fetch.finish()
    torch.cuda.synchronize()
    T['total'] = time.perf_counter() - t0
    T['gpu_busy'] = sum(a.elapsed_time(b) for a, b in evs) * 1e-3
    T['gpu_gaps'] = sum(evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(len(evs) - 1)) * 1e-3
    T['max_busy'] = max(a.elapsed_time(b) for a, b in evs) * 1e-3
    T.update(hp.TRACE)
    return {k: round(v * 1e3, 1) for k, v in T.items()}
for v in ('full', 'full', 'nocollect', 'nocollect'):
    print(v, run(v))
hp.TRACE = None
for _ in range(3):
    t0 = time.perf_counter(); r = net.predict(data); torch.cuda.synchronize(); t1 = time.perf_counter()
    del r
    t2 = time.perf_counter()
    print('net.predict', round((t1 - t0) * 1e3, 1), 'ms; free', round((t2 - t1) * 1e3, 1), 'ms')
