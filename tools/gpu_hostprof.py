"""Where the wall time of one lock-step batch goes on the host (not a test):  PORESEQ_TRACE=1 python tools/gpu_hostprof.py [R] [L] 2> trace.log
then  python tools/tracesum.py trace.log.  Prints wall time and the time spent inside the C ABI (ctypes calls)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth, _capi
from poreseq_amd.batch import RegionBatch
from poreseq_amd.consensus import consensus_regions
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
P = dict(DEFAULT_PARAMS, verbose=0)
api = _capi.load_hip()
regs = [synth.make_region(L, 10, 1002 + k, swalign, P) for k in range(R)]
def pas():
    out = []
    for d, ev, _ in regs:
        pa = PSAlign(); pa.sequence, pa.events, pa.params = d, copy.deepcopy(ev), dict(P); out.append(pa)
    return out
# time inside the library: wrap every ctypes function of the loaded library
incall = [0.0]
lib = api.lib
class Timed:
    def __init__(self, f): self.f = f
    def __call__(self, *a):
        t = time.perf_counter()
        try: return self.f(*a)
        finally: incall[0] += time.perf_counter() - t
for name in _capi.SYMBOLS:
    setattr(lib, name, Timed(getattr(lib, name)))
def run():
    p = pas()
    with RegionBatch(p) as rb:
        rb.load()
        incall[0] = 0.0
        t = time.perf_counter()
        consensus_regions(p, P, batch=rb)
        return time.perf_counter() - t, incall[0]
run()
sys.stderr.write("=== MEASURED RUN ===\n")
w, c = run()
print("wall %.3f s, inside the C ABI %.3f s (%.0f %%), Python %.3f s" % (w, c, 100 * c / w, w - c))
sys.stderr.write("wall %.3f s\n" % w)
