"""Bring-up timing of the consensus schedule on the GPU (not a test)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_region
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
api = _capi.load_hip()
for L in [int(x) for x in sys.argv[1:]] or [1000, 10000]:
    t = time.time(); draft, events, truth = synth.make_region(L, 10, 1002, swalign, P); tg = time.time() - t
    for rep in range(2):
        pa = PSAlign(); pa.sequence = draft; pa.events = copy.deepcopy(events); pa.params = dict(P)
        t = time.time(); s = pa.ScoreEvents(); t1 = time.time() - t
        t = time.time(); sp = pa.ScorePoints(); t2 = time.time() - t
        log = []
        api.prof_reset(); api.prof_enable(len(sys.argv) > 1 and 'prof' in os.environ.get('PS_TIMING',''))
        t0 = time.time()
        tcalls = []
        class TL(list):
            def append(self, x): tcalls.append(time.time()); list.append(self, x)
        log = TL()
        seq, acc = consensus_region(pa, dict(P, end_trim=0), log=log)
        tc = time.time() - t0
        print("L=%d gen %.2fs ScoreEvents %.4fs ScorePoints(%d) %.4fs consensus %.3fs -> %.3f kb/s acc %.2f%% start %.2f%%" % (
            L, tg, t1, len(sp), t2, tc, L / 1000.0 / tc, swalign(seq, truth)[0], swalign(draft, truth)[0]))
        prev = t0
        for (c, nb, _), tt in zip(log, tcalls):
            print("   %-15s nb=%-4d %.3fs" % (c, nb, tt - prev)); prev = tt
        for k in ("fill", "score", "sw", "viterbi"):
            print("   prof", k, api.prof_get(k))
