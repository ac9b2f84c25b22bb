"""poreseq_amd — MI355X-native implementation of PoreSeq's event-level HMM scoring path.

`poreseq_amd.poreseqcpp` mirrors the reference's compiled module (`PSAlign`, `swalign`,
`seqtostates`); `poreseq_amd.consensus` reproduces the call schedule of the reference's
consensus / variant / train drivers on top of it (`consensus_region`, lock-step `consensus_regions`,
`polish`); `poreseq_amd.batch.RegionBatch` runs the PSAlign calls of several regions in lock-step on one
GPU; `poreseq_amd.dist` shards regions over one process per GPU.
"""
import os as _os


def _gpu_opened():
    """True when something in this process has already opened the GPU (the HSA runtime holds /dev/kfd from its first call on):
    GPU_MAX_HW_QUEUES exported now would come too late for HIP to read it."""
    try:
        for fd in _os.listdir("/proc/self/fd"):
            try:
                if _os.readlink("/proc/self/fd/" + fd).startswith("/dev/kfd"):
                    return True
            except OSError:
                pass
    except OSError:
        return True   # cannot tell: assume the worst
    return False


def _want_hw_queues():
    """Several lock-step batches per GPU (one host thread and one HIP stream each) want a hardware queue per stream; HIP's default is
    four per process and priority level, and streams that share one run their kernels one after the other.  GPU_MAX_HW_QUEUES is
    read when the HIP runtime starts, so it is exported here, at import — but only when nobody has set it and nothing in the process
    has opened the GPU yet (torch.cuda.is_available(), another HIP library ...: the variable would be ignored, and the library, told
    that it is in force, would put every stream on one priority level: seven streams on four queues).  PORESEQ_HWQ_SET_BY_PACKAGE=1
    tells the library that the value can be trusted; otherwise it deals its streams over the device's priority levels
    (ps_host.cpp, hwq_mode; `poreseq_amd._capi.load_hip().info()` says which mode a process is in)."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    if _gpu_opened():
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"
    _os.environ["PORESEQ_HWQ_SET_BY_PACKAGE"] = "1"


_want_hw_queues()

from .util import RegionInfo, MutationInfo, MutationScore, LoadParams, SaveParams, VaryParams, DEFAULT_PARAMS  # noqa: F401
from .events import PSEvent, PSModel  # noqa: F401
from . import poreseqcpp  # noqa: F401
from .poreseqcpp import PSAlign, swalign, seqtostates  # noqa: F401

__version__ = "0.1.0"
