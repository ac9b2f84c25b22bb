"""poreseq_amd — MI355X-native implementation of PoreSeq's event-level HMM scoring path.

`poreseq_amd.poreseqcpp` mirrors the reference's compiled module (`PSAlign`, `swalign`,
`seqtostates`); `poreseq_amd.consensus` reproduces the call schedule of the reference's
consensus / variant / train drivers on top of it (`consensus_region`, lock-step `consensus_regions`,
`polish`); `poreseq_amd.batch.RegionBatch` runs the PSAlign calls of several regions in lock-step on one
GPU; `poreseq_amd.dist` shards regions over one process per GPU.
"""
import os as _os
import sys as _sys


def _want_hw_queues():
    """Several lock-step batches per GPU (one host thread and one HIP stream each) want a hardware queue per stream; HIP's default is
    four per process and priority level, and streams that share one run their kernels one after the other.  GPU_MAX_HW_QUEUES is
    read when the HIP runtime starts, so it is set here, at import, unless the user has set it or torch has already initialised the
    GPU (then the library deals its streams over the device's priority levels instead: ps_host.cpp)."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    t = _sys.modules.get("torch")
    try:
        if t is not None and t.cuda.is_initialized():
            return
    except Exception:   # pragma: no cover
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = "12"


_want_hw_queues()

from .util import RegionInfo, MutationInfo, MutationScore, LoadParams, SaveParams, VaryParams, DEFAULT_PARAMS  # noqa: F401
from .events import PSEvent, PSModel  # noqa: F401
from . import poreseqcpp  # noqa: F401
from .poreseqcpp import PSAlign, swalign, seqtostates  # noqa: F401

__version__ = "0.1.0"
