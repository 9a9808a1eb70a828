#!/usr/bin/env python3
"""Host-side rates that bound predict() from host arrays: first-touch of a fresh result array, pageable -> pinned copies by
1..8 threads, pinned <-> HBM copies.  Diagnostic for host_pipeline.py (numbers in DESIGN.md)."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

out = {'cpus': len(os.sched_getaffinity(0))}
try:
    out['thp'] = open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip()
except OSError:
    pass
n, per = 256, 384 * 768
src = np.random.default_rng(0).integers(0, 12, (n, 384, 768)).astype(np.int64)   # 604 MB, touched
t0 = time.perf_counter(); dst = np.empty_like(src); np.copyto(dst, src); t1 = time.perf_counter()
out['first_touch_copy_GBps_1thr'] = round(src.nbytes / (t1 - t0) / 1e9, 2)
t0 = time.perf_counter(); np.copyto(dst, src); t1 = time.perf_counter()
out['warm_copy_GBps_1thr'] = round(src.nbytes / (t1 - t0) / 1e9, 2)
for thr in (2, 4, 8, 16):
    pool = ThreadPoolExecutor(thr)
    for fresh in (True, False):
        d = np.empty_like(src) if fresh else dst
        step = n // thr
        t0 = time.perf_counter()
        list(pool.map(lambda lo: np.copyto(d[lo:lo + step], src[lo:lo + step]), range(0, n, step)))
        t1 = time.perf_counter()
        out['%s_copy_GBps_%dthr' % ('first_touch' if fresh else 'warm', thr)] = round(src.nbytes / (t1 - t0) / 1e9, 2)
pin = torch.empty((16, 384, 768, 4), dtype=torch.float32, pin_memory=True)
dev = torch.empty((16, 384, 768, 4), dtype=torch.float32, device='cuda')
for _ in range(2):
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
out['h2d_pinned_GBps'] = round(10 * pin.numel() * 4 / (t1 - t0) / 1e9, 2)
t0 = time.perf_counter()
for _ in range(10):
    pin.copy_(dev, non_blocking=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
out['d2h_pinned_GBps'] = round(10 * pin.numel() * 4 / (t1 - t0) / 1e9, 2)
pag = torch.empty((16, 384, 768, 4), dtype=torch.float32)
pag.fill_(1.0)
t0 = time.perf_counter()
for _ in range(5):
    dev.copy_(pag); torch.cuda.synchronize()
t1 = time.perf_counter()
out['h2d_pageable_GBps'] = round(5 * pag.numel() * 4 / (t1 - t0) / 1e9, 2)
t0 = time.perf_counter(); big = torch.empty((256, 384, 768), dtype=torch.int64, pin_memory=True); t1 = time.perf_counter()
out['pinned_alloc_604MB_s'] = round(t1 - t0, 4)
print(json.dumps(out))
