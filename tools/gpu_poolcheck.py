"""First GPU run of the op-level scheduler (poreseq_amd.pool; not a test, not yet part of the suite):
    python tools/gpu_poolcheck.py [R] [L] [workers] [batch_size]
1. exactness: R regions through consensus_pool (several worker threads) against the same regions through consensus_regions
   (one lock-step batch) — sequences, per-call logs, ref_align / ref_like must be identical;
2. time of both."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreseq_amd import synth
from poreseq_amd.consensus import consensus_regions
from poreseq_amd.pool import consensus_pool
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS
R = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4
BS = int(sys.argv[4]) if len(sys.argv) > 4 else 8
P = dict(DEFAULT_PARAMS, verbose=0)
regs = [synth.make_region(L if k % 3 else L // 2, 10 if k % 4 else 7, 6100 + k, swalign, P) for k in range(R)]
def pas():
    out = []
    for d, ev, _ in regs:
        pa = PSAlign(); pa.sequence, pa.events, pa.params = d, copy.deepcopy(ev), dict(P); out.append(pa)
    return out
def snap(ps, res, logs):
    return ([tuple(x) for x in res], logs, [[np.array(e.ref_align) for e in pa.events] for pa in ps], [[np.array(e.ref_like) for e in pa.events] for pa in ps])
consensus_regions(pas()[:2], P)      # warm: pools, code objects
a = pas(); la = [[] for _ in regs]
t = time.time(); ra = consensus_regions(a, P, logs=la); ta = time.time() - t
b = pas(); lb = [[] for _ in regs]
t = time.time(); rb = consensus_pool(b, P, logs=lb, workers=W, batch_size=BS); tb = time.time() - t
sa, sb = snap(a, ra, la), snap(b, rb, lb)
same = sa[0] == sb[0] and sa[1] == sb[1] and all(np.array_equal(u, v) for x, y in zip(sa[2], sb[2]) for u, v in zip(x, y)) \
    and all(np.array_equal(u, v) for x, y in zip(sa[3], sb[3]) for u, v in zip(x, y))
print("pool == lock-step: %s   (%d regions of %d / %d bases; lock-step batch %.2f s, pool with %d workers x %d regions %.2f s)" % (same, R, L, L // 2, ta, W, BS, tb))
sys.exit(0 if same else 1)
