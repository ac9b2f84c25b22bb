import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_regions
from poreseq_amd.batch import RegionBatch
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
regs = [synth.make_region(10000, 10, 1002 + k, swalign, P) for k in range(16)]
api = _capi.load_hip()
def pas():
    out = []
    for d, e, t in regs:
        pa = PSAlign(); pa.sequence, pa.events, pa.params = d, copy.deepcopy(e), dict(P); out.append(pa)
    return out
consensus_regions(pas(), P)
for resident in (True, False, True, False):
    p = pas()
    t0 = time.perf_counter(); rb = RegionBatch(p, resident=resident).load(); t1 = time.perf_counter()
    consensus_regions(p, P, batch=rb); t2 = time.perf_counter()
    print("resident=%s: load %.3f s (%.1f ms per region), schedule %.3f s, total %.3f s" % (resident, t1 - t0, 1e3 * (t1 - t0) / 16, t2 - t1, t2 - t0))
