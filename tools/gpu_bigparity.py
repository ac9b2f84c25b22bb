"""One-off validation (minutes of CPU): the full stochastic consensus schedule on a 3 kb / 10-event region, HIP vs the
CPU oracle, every intermediate sequence compared (not part of the suite: the oracle is O(L^2))."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import backends as B
from poreseq_amd import synth
from poreseq_amd.consensus import consensus_region
from poreseq_amd.poreseqcpp import PSAlign
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
draft, events, truth = synth.make_region(L, 10, 4242, B.oracle_swalign, P)
logs = {}
for name, cls in (("hip", PSAlign), ("oracle", B.OraclePSAlign)):
    B.reset_rand()
    pa = B.make_pa(cls, draft, copy.deepcopy(events), P)
    log = []
    t = time.time(); seq, acc = consensus_region(pa, P, log=log, refseq=truth); dt = time.time() - t
    logs[name] = (seq, [(c, nb, s) for c, nb, s in log], [ev.ref_align.copy() for ev in pa.events], [ev.ref_like.copy() for ev in pa.events])
    print("%-6s %.1f s, accuracy %.3f %%, %d calls" % (name, dt, acc, len(log)), flush=True)
h, o = logs["hip"], logs["oracle"]
print("final sequence identical:", h[0] == o[0])
print("per-call (name, nbases, sequence) identical:", h[1] == o[1])
print("ref_align identical:", all(np.array_equal(a, b) for a, b in zip(h[2], o[2])), " ref_like identical:", all(np.array_equal(a, b) for a, b in zip(h[3], o[3])))
