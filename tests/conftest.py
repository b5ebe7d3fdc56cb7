import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(REPO, 'tests', 'golden', 'reference_host.npz'))


@pytest.fixture(autouse=True)
def _poisoned_allocations(monkeypatch):
    """DAT_POISON=1 (a GPU-box debugging pass, with DAT_WS_POISON=1 for the C-ABI scratch): every device tensor that torch.empty hands out is
    filled with 0xFF bytes (fp32 NaN, int32 -1), so a kernel that reads rows nobody wrote -- and only works while the caching allocator
    happens to return zeroed or look-alike memory -- fails on every run instead of on some."""
    if not os.environ.get('DAT_POISON'):
        yield
        return
    import torch
    real = torch.empty

    def empty(*a, **k):
        t = real(*a, **k)
        if t.is_cuda and t.numel():
            t.view(-1).view(torch.uint8).fill_(0xFF)
            # the fill runs on the CURRENT stream, the buffer's first writer may be another one (upload buffers: the copy stream):
            # wait for it, or the poison itself races the product (captures cannot wait and need not: one stream)
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()
        return t
    monkeypatch.setattr(torch, 'empty', empty)
    yield
