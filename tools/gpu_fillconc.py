"""Concurrent Alignment::update fills from several host threads (not a test):  python tools/gpu_fillconc.py THREADS [R] [L]
Each thread owns R regions and calls the lock-step ScoreMutations (forward + backward sweep per event, short edit list) in a loop;
prints the mean time per call and the fill launches' HIP-event time.  Shows how the fills of batches in flight slow each other."""
import copy, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import swalign
from poreseq_amd.util import DEFAULT_PARAMS
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
P = dict(DEFAULT_PARAMS, verbose=0)
api = _capi.load_hip()
regs = [synth.make_region(L, 10, 1002 + k, swalign, P) for k in range(3)]
res = [None] * T
def work(t):
    hs = [api.align_create(regs[k % 3][0], copy.deepcopy(regs[k % 3][1]), P) for k in range(R)]
    rng = np.random.default_rng(t)
    hm = [api.muts_create(synth.random_point_mutations(rng, regs[k % 3][0], 20)) for k in range(R)]
    def once():
        for m in api.batch_score_mutations(hs, hm): api.muts_destroy(m)
    once(); once()
    api.prof_reset(); api.prof_enable(2)
    N = 6
    ts = []
    for _ in range(N):
        t0 = time.time(); once(); ts.append(time.time() - t0)
    res[t] = (sum(ts) / N, api.prof_get("fill"), ts)
th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
[x.start() for x in th]; [x.join() for x in th]
print("threads %d x %d regions: %.1f ms per call; fill launch avg %.1f ms (%s)" % (
    T, R, 1e3 * sum(r[0] for r in res) / T, sum(r[1][0] for r in res) / max(1, sum(r[1][1] for r in res)),
    os.environ.get("PORESEQ_EXPERIMENT_NOSTORE") and "NO STORES" or "stores"))
print("   thread 0 calls (ms):", " ".join("%.0f" % (1e3 * x) for x in res[0][2]))
