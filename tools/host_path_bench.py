#!/usr/bin/env python3
"""predict() / score() / fit() over HOST-resident numpy samples (the API the reference's callers use, base_model.py:265-331):
images/s including the host -> HBM copies and the label fetch.  usage: host_path_bench.py [samples] [batchsize]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device('cuda', 0)
    net = bench.build_model(dev, 'bayes', 'fcn', bs, 'bf16')
    rng = np.random.default_rng(0)
    data = {'rgb': rng.integers(0, 256, (n, 384, 768, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (n, 384, 768, 1)).astype(np.float32),
            'labels': rng.integers(-1, 12, (n, 384, 768)).astype(np.int32)}
    warm = {k: v[:2 * bs] for k, v in data.items()}
    out = {'samples': n, 'batchsize': bs}
    for name, fn in (('predict', lambda d: net.predict({k: v for k, v in d.items() if k != 'labels'})),
                     ('score', lambda d: net.score(d))):
        fn(warm)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            result = fn(data)              # kept until the clock has stopped: freeing 600 MB of label maps is the caller's
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            del result
        out[name] = {'images_per_s': round(n / min(ts), 1), 'seconds': [round(t, 4) for t in ts]}
    # fit(): 32 training steps of the RGB expert from the same host arrays (batches staged and uploaded ahead of the step)
    del net
    torch.cuda.empty_cache()
    out['fit'] = bench.host_fit_rate(dev, {k: data[k] for k in ('rgb', 'labels')}, bs)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
