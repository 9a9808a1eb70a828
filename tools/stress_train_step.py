#!/usr/bin/env python3
"""Run-to-run comparison of whole training steps: a fresh model + trainer every iteration (fresh allocations, first-step
code paths), one step on fixed data, then every activation / gradient map the step left in the trainer's arenas is
compared with the first iteration's.  The PLAIN FCN step (no batch norm) is deterministic by construction since round 3 --
every partial sum of the filter / bias gradients, the first layer's and the head's goes to slabs added in a fixed order --
and must agree BIT FOR BIT, loss, gradients and updated parameters included.  The batch-norm step's statistics and
gradient sums go through per-workgroup partials added in a fixed order too (every activation / gradient MAP of that step
is bitwise reproducible); its dense head (loss, score-layer gradients) and the joint step keep floating-point atomics:
low-bit differences there are expected and counted; anything beyond --tol of a map's largest entry is a defect.  GPU box
only."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modular_semantic_segmentation_amd import get_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=60)
ap.add_argument('--tol', type=float, default=2e-2, help='largest tolerated run-to-run difference, relative to the map maximum')
ap.add_argument('--cold', action='store_true', help='release the caching allocator between iterations')
args = ap.parse_args()
C, U, H, W = 12, 64, 32, 48
bad = 0
WEIGHTS = {}


def arenas(tr):
    out = {}
    for name in ('_g', '_a'):
        for key, v in getattr(tr, name, {}).items():
            t = v.t if hasattr(v, 'interior') else v
            if isinstance(key, tuple) and str(key[0]).endswith('_ws'):
                continue        # scratch workspaces: the part a step does not write keeps whatever the allocator handed out
            if torch.is_tensor(t):
                out[(name, str(key))] = t
    engines = [tr.e] if not hasattr(tr.e, 'trunks') else list(tr.e.trunks.values())
    for i, e in enumerate(engines):
        for key, v in e._arena.items():
            if hasattr(v, 'interior'):
                out[('arena%d' % i, str(key))] = v.t
    out[('grad', '')] = tr.grad
    out[('param', '')] = tr.param
    out[('loss', '')] = tr.loss
    return out


def run(kind, it):
    rng = np.random.default_rng(4)
    data = {'rgb': rng.integers(0, 256, (2, H, W, 3)).astype(np.float32),
            'depth': rng.integers(0, 65536, (2, H, W, 1)).astype(np.float32),
            'labels': rng.integers(-1, C, (2, H, W)).astype(np.int32)}
    if kind == 'joint':
        prefixes, nch = {'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}
        net = get_model('fusion_fcn')(prefixes, nch, U, C, trainer='rmsprop', learning_rate=1e-3, batchsize=2, seed=5)
        inputs = {m: torch.from_numpy(data[m]).cuda() for m in prefixes}
    else:
        desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, C)
        net = get_model('fcn')('rgb', desc, 'rgb', num_units=U, batch_normalization=kind == 'bn', batchsize=2, learning_rate=1e-3)
        inputs = torch.from_numpy(data['rgb']).cuda()
    # the first model's own initialisation (scaled so that the deep layers stay alive) for every iteration
    if kind not in WEIGHTS:
        w = {k: np.array(v, copy=True) for k, v in net.variables.items()}
        for k in w:
            if k.endswith('/kernel') and 'upscore' not in k:
                w[k] = w[k] * (0.02 if 'conv1_1' in k else 1.6)
        WEIGHTS[kind] = w
    net.variables.update({k: np.array(v, copy=True) for k, v in WEIGHTS[kind].items()})
    net._variables_changed()
    tr = net._ensure_trainer()
    tr.step(inputs, torch.from_numpy(data['labels']).cuda())
    torch.cuda.synchronize()
    return {k: v.detach().clone().double().cpu() for k, v in arenas(tr).items()
            if v.dtype in (torch.bfloat16, torch.float32, torch.float64)}


for kind in ('plain', 'bn', 'joint'):
    want = None
    for it in range(args.iters):
        if args.cold:
            torch.cuda.empty_cache()
        got = run(kind, it)
        if want is None:
            want = got
            continue
        small, names = 0, []
        for k in want:
            if k not in got or got[k].shape != want[k].shape:
                continue
            scale = want[k].abs().max().item() + 1e-30
            d = (got[k] - want[k]).abs().max().item() / scale
            if not d <= (0.0 if kind == 'plain' else args.tol):         # also catches NaN; the plain step: bit for bit
                bad += 1
                print('MISMATCH', kind, 'iter', it, k, 'max difference %.3g of the largest entry' % d, flush=True)
            elif d > 0:
                small += 1
                names.append(k)
        if small and it == 1:
            print(kind, ': %d maps differ in low bits run to run (order of the floating-point atomics): %s' % (small, names),
                  flush=True)
    print(kind, 'done', flush=True)
print('mismatches:', bad)
sys.exit(1 if bad else 0)
