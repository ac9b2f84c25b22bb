"""poreseq_amd — MI355X-native implementation of PoreSeq's event-level HMM scoring path.

`poreseq_amd.poreseqcpp` mirrors the reference's compiled module (`PSAlign`, `swalign`,
`seqtostates`); `poreseq_amd.consensus` reproduces the call schedule of the reference's
consensus / variant / train drivers on top of it (`consensus_region`, lock-step `consensus_regions`,
`polish`); `poreseq_amd.batch.RegionBatch` runs the PSAlign calls of several regions in lock-step on one
GPU; `poreseq_amd.dist` shards regions over one process per GPU.
"""
from .util import RegionInfo, MutationInfo, MutationScore, LoadParams, SaveParams, VaryParams, DEFAULT_PARAMS  # noqa: F401
from .events import PSEvent, PSModel  # noqa: F401
from . import poreseqcpp  # noqa: F401
from .poreseqcpp import PSAlign, swalign, seqtostates  # noqa: F401

__version__ = "0.1.0"
