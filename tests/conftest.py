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


# Forward-only alignment batches have two kernels (include/poreseq_hip.h, ps_set_sweep_min): one wavefront per alignment
# (k_sweep, default from 400 alignments on) and one workgroup per alignment (k_fill).  The kernel-level parity modules run
# every gpu test under both; the other gpu tests run the default, the large golden schedules k_sweep throughout.
_BOTH_KERNELS = ("test_hip_parity", "test_hip_variant")


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.split(".")[-1]
    if mod in _BOTH_KERNELS and metafunc.definition.get_closest_marker("gpu"):
        if "fwd_kernel" not in metafunc.fixturenames:
            metafunc.fixturenames.append("fwd_kernel")
        metafunc.parametrize("fwd_kernel", ["sweep", "fill"], indirect=True)


@pytest.fixture
def fwd_kernel(request):
    from poreseq_amd import _capi
    api = _capi.load_hip()
    api.set_sweep_min(0 if request.param == "sweep" else 1 << 30)
    api.set_sweep2_min(0 if request.param == "sweep" else 1 << 30)
    api.set_sparse_min(0 if request.param == "sweep" else 1 << 30)
    yield request.param
    api.set_sweep_min(-1)
    api.set_sweep2_min(-1)
    api.set_sparse_min(-1)


@pytest.fixture
def sweep_always():
    from poreseq_amd import _capi
    api = _capi.load_hip()
    api.set_sweep_min(0)
    api.set_sweep2_min(0)
    api.set_sparse_min(0)
    yield
    api.set_sweep_min(-1)
    api.set_sweep2_min(-1)
    api.set_sparse_min(-1)
