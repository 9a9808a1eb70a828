"""The two race detectors of tools/ as (short) regression tests.

`stress_cold.py` builds every operand in freshly mapped device memory and leaves foreign data in every CU's LDS before each
launch -- the conditions under which a missing wait for an LDS-DMA shows (round 2: the first-generation conv kernel passed
its item barrier with weight pieces still in flight about once in 100 such launches; warm benchmark loops never saw it).
`stress_train_step.py` runs one training step of a FRESH model per iteration and compares every map the step leaves behind
run to run."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', script)] + list(args), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert 'mismatches: 0' in r.stdout, r.stdout[-2000:]


def test_small_convs_on_cold_memory_with_dirty_lds():
    _run('stress_cold.py', '--iters', '40')


def test_training_steps_of_fresh_models_agree_run_to_run():
    _run('stress_train_step.py', '--iters', '8')
