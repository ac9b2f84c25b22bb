import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0   # counts devices without initialising the GPU
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them."""
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device on this machine (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
