import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def cpu_ops(monkeypatch):
    """Swap every HIP op for its CPU restatement from oracle/ so the HOST logic (caches, engines,
    decode loops, TP sharding) can be exercised without a GPU.  Test-only: the product has no CPU path."""
    from tests import cpu_backend
    import triforce_amd.ops as ops
    for name in cpu_backend.PATCHED:
        monkeypatch.setattr(ops, name, getattr(cpu_backend, name))
    return ops


def pytest_collection_modifyitems(config, items):
    """``pytest tests`` on a box without a HIP device: skip the gpu-marked tests instead of failing at the first one."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
