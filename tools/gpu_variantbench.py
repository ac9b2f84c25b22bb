"""BASELINE config #3 shape: ScoreMutations of 10 000 point edits on a 48 kb region with 30 events (not a test)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
L, E, M = 48000, 30, 10000
t = time.time(); draft, events, truth = synth.make_region(L, E, 1003, swalign, P, draft_error=0.0); print("gen %.1fs" % (time.time() - t))
rng = np.random.default_rng(3)
muts = synth.random_point_mutations(rng, draft, M)
pa = PSAlign(); pa.sequence, pa.events, pa.params = draft, events, dict(P)
api = _capi.load_hip()
for rep in range(3):
    t = time.time(); sc = pa.ScoreMutations(copy.deepcopy(muts)); dt = time.time() - t
    print("ScoreMutations(%d edits, %d events, %d bases, scoring_width %s): %.3f s" % (M, E, L, P["scoring_width"], dt))
