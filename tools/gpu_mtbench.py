"""Concurrent regions per GPU: R host threads, each refining its own 10 kb region (bring-up measurement)."""
import copy, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_region
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
regions = [synth.make_region(L, 10, 1002 + k, swalign, P) for k in range(16)]
def work(k, n, out):
    for r in range(n):
        draft, events, truth = regions[(k + r) % len(regions)]
        pa = PSAlign(); pa.sequence, pa.events, pa.params = draft, copy.deepcopy(events), dict(P)
        seq, _ = consensus_region(pa, P)
        out[k] = swalign(seq, truth)[0]
for R in (1, 2, 4, 8, 12, 16):
    out = [0] * R
    th = [threading.Thread(target=work, args=(k, 1, out)) for k in range(R)]   # warm-up (pools)
    [t.start() for t in th]; [t.join() for t in th]
    th = [threading.Thread(target=work, args=(k, 2, out)) for k in range(R)]
    t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]; dt = time.time() - t0
    print("R=%2d  %.3f s for %d regions -> %.2f kb/s  (min acc %.2f%%)" % (R, dt, 2 * R, 2 * R * L / 1000.0 / dt, min(out)), flush=True)
