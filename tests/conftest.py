import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def poison_free_device_memory(request):
    """GPU tests start from POISONED free memory: the caching allocator's free blocks are released, 6 GB are filled with
    0xFF bytes (NaN as bf16 / fp32 / fp64, -1 as an integer) and handed back to the allocator, so every buffer a test
    allocates with torch.empty starts as NaNs.  A kernel that reads a workspace, halo or gradient map before anything
    wrote it then fails every time -- without this it fails only on a GPU whose memory does not happen to hold the same
    test's data from an earlier process (seen as a 1-in-10 first-process-on-the-box flake).  XV_NO_POISON=1 disables."""
    if request.node.get_closest_marker('gpu') is None or os.environ.get('XV_NO_POISON') == '1':
        yield
        return
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        block = torch.empty(6 * 1024 ** 3, dtype=torch.uint8, device='cuda')
        block.fill_(0xFF)
        torch.cuda.synchronize()
        del block
    yield
