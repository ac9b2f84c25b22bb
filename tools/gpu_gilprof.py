"""How much of one consensus run is spent inside the native library (GIL released) vs in Python (not a test)."""
import copy, os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_region
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
api = _capi.load_hip()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
draft, events, truth = synth.make_region(L, 10, 1002, swalign, P)
acc = {}
class Timed:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        f = getattr(self._lib, name)
        def g(*a):
            t = time.perf_counter(); r = f(*a); dt = time.perf_counter() - t
            e = acc.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += dt
            return r
        return g
for rep in range(2):
    acc.clear()
    real = api.lib
    api.lib = Timed(real)
    pa = PSAlign(); pa.sequence = draft; pa.events = copy.deepcopy(events); pa.params = dict(P)
    t = time.perf_counter(); consensus_region(pa, P); tot = time.perf_counter() - t
    api.lib = real
nat = sum(v[1] for v in acc.values())
print("total %.3f s, native %.3f s, python %.3f s" % (tot, nat, tot - nat))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-28s calls %5d  %.3f s" % (k, v[0], v[1]))
