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


# The DP fills have two kernel families (include/poreseq_hip.h, ps_set_sweep_min / ps_set_sweep_form): strip sweeps — one, two or
# four wavefronts per alignment and direction (k_sweep / k_sweeps / k_sweep2 and their _w builds, ps_sweep.hip, ps_sweepw.hip) — and
# one workgroup per alignment with skewed matrices (k_fill).  The kernel-level parity modules run every gpu test under all four;
# the other gpu tests run the library's own choice, the large golden schedules the strip sweeps throughout.
_BOTH_KERNELS = ("test_hip_parity", "test_hip_variant")
_FWD_KERNELS = {"sweep": 1, "sweep_w2": 2, "sweep_w4": 4, "fill": 0}    # name -> wavefronts per sweep (0: k_fill)


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.split(".")[-1]
    if mod in _BOTH_KERNELS and metafunc.definition.get_closest_marker("gpu"):
        if "fwd_kernel" not in metafunc.fixturenames:
            metafunc.fixturenames.append("fwd_kernel")
        metafunc.parametrize("fwd_kernel", list(_FWD_KERNELS), indirect=True)


@pytest.fixture
def fwd_kernel(request):
    from poreseq_amd import _capi
    api = _capi.load_hip()
    nw = _FWD_KERNELS[request.param]
    api.set_sweep_min(0 if nw else 1 << 30)
    api.set_sweep2_min(0 if nw else 1 << 30)
    api.set_sparse_min(0 if nw else 1 << 30)
    api.set_sweep_form(0, nw)
    yield request.param
    api.set_sweep_min(-1)
    api.set_sweep2_min(-1)
    api.set_sparse_min(-1)
    api.set_sweep_form(0, 0)


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
