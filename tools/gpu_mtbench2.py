import copy, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_region
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
L = 10000; R = int(sys.argv[1])
regions = [synth.make_region(L, 10, 1002 + k, swalign, P) for k in range(R)]
def work(k, n):
    for r in range(n):
        draft, events, truth = regions[k]
        pa = PSAlign(); pa.sequence, pa.events, pa.params = draft, copy.deepcopy(events), dict(P)
        consensus_region(pa, P)
th = [threading.Thread(target=work, args=(k, 1)) for k in range(R)]
[t.start() for t in th]; [t.join() for t in th]
sys.stderr.write('=== MEASURED RUN ===\n'); sys.stderr.flush()
th = [threading.Thread(target=work, args=(k, 1)) for k in range(R)]
t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]; print("wall", time.time() - t0)
