import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure; builds oracle/liboracle.so on first use)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "lsb_golden.npz"))


@pytest.fixture(scope="session")
def gs():
    """The product package; importing it loads libgpusort.so (no fallback)."""
    import gpu_sort_amd
    return gpu_sort_amd


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch.device("cuda:0")


def to_dev(a, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32).copy()).to(device)


def to_u32(t):
    return t.cpu().numpy().view(np.uint32)
