"""Same-box A/B of two builds of the library on the training step: `python tools/lib_ab_train.py` runs itself once per library
(XV_LIB=<path> with XV_ALLOW_STALE_LIB=1 for the other build; default: tools/build/libxview_hip_old.so against the in-tree
one) and prints, per build, a SHA-256 of loss + gradients + updated parameters after one step on fixed data (equal hashes =
the same bits) and the events-timed step of the headline training shape."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(kind):
    import numpy as np
    import torch
    from modular_semantic_segmentation_amd import _lib, get_model
    if os.environ.get('XV_LIB'):
        _lib.LIB_PATH = os.environ['XV_LIB']
    C, U = 12, 64
    out = {}
    for (n, h, w, steps) in ((2, 96, 128, 1), (16, 384, 768, 12)):
        rng = np.random.default_rng(4)
        rgb = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3)).astype(np.float32)).cuda()
        lab = torch.from_numpy(rng.integers(-1, C, (n, h, w)).astype(np.int32)).cuda()
        desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, C)
        net = get_model('fcn')('rgb', desc, 'rgb', num_units=U, batch_normalization=kind == 'bn', batchsize=n,
                               learning_rate=1e-3, seed=5)
        tr = net._ensure_trainer()
        tr.step(rgb, lab)
        torch.cuda.synchronize()
        if steps == 1:
            hsh = hashlib.sha256()
            for t in (tr.loss, tr.grad, tr.param):
                hsh.update(t.detach().cpu().numpy().tobytes())
            out['sha'] = hsh.hexdigest()[:16]
            out['loss'] = float(tr.loss)
        else:
            for _ in range(3):
                tr.step(rgb, lab)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(steps):
                tr.step(rgb, lab)
            b.record()
            torch.cuda.synchronize()
            out['ms_per_step'] = round(a.elapsed_time(b) / steps, 3)
    print(json.dumps(out))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(sys.argv[2])
    else:
        other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tools', 'build', 'libxview_hip_old.so')
        for kind in ('plain', 'bn'):
            for rep in range(2):
                for name, env in (('other', {'XV_LIB': other, 'XV_ALLOW_STALE_LIB': '1'}), ('in-tree', {})):
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', kind], env={**os.environ, **env},
                                       capture_output=True, text=True)
                    print(kind, name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
