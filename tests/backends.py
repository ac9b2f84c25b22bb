"""Test-only backends: the CPU oracle and (when built) the real reference C++ behind the same
C ABI as the product.  Nothing in poreseq_amd/ knows these paths."""
import ctypes
import os
import subprocess

from poreseq_amd import _capi
from poreseq_amd.poreseqcpp import PSAlign, swalign, seqtostates

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.environ.get("PORESEQ_ORACLE_SO") or os.path.join(ROOT, "oracle", "libps_oracle.so")   # tools/oracle_asan.sh overrides
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libps_ref.so")

_cache = {}


def build_oracle():
    if os.environ.get("PORESEQ_ORACLE_SO"):
        return
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "ps_oracle.cpp")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libps_oracle.so"])


def oracle_api():
    if "oracle" not in _cache:
        build_oracle()
        _cache["oracle"] = _capi.CApi(ORACLE_SO)
    return _cache["oracle"]


def have_ref():
    return os.path.exists(REF_SO)


def ref_api():
    if "ref" not in _cache:
        _cache["ref"] = _capi.CApi(REF_SO)
    return _cache["ref"]


class OraclePSAlign(PSAlign):
    _native = staticmethod(oracle_api)


class RefPSAlign(PSAlign):
    _native = staticmethod(ref_api)


def oracle_swalign(a, b):
    return swalign(a, b, oracle_api)


def ref_swalign(a, b):
    return swalign(a, b, ref_api)


_libc = ctypes.CDLL(None)


def reset_rand():
    """libc rand() is never seeded by the reference (Viterbi.cpp:108): srand(1) == fresh process."""
    _libc.srand(1)
    try:   # the HIP library draws from a per-thread generator (include/poreseq_hip.h, ps_srand)
        from poreseq_amd import _capi
        _capi.load_hip().srand(1)
    except Exception:
        pass


def make_pa(cls, sequence, events, params):
    pa = cls()
    pa.sequence = sequence
    pa.events = events
    pa.params = dict(params)
    return pa
