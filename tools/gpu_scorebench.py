"""Time the edit scoring (k_score) of a Refine-sized list on one region (not a test):  python tools/gpu_scorebench.py [L] [scoring_width]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import swalign
from poreseq_amd.util import DEFAULT_PARAMS
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
P = dict(DEFAULT_PARAMS, verbose=0)
if len(sys.argv) > 2: P['scoring_width'] = int(sys.argv[2])
api = _capi.load_hip()
draft, events, truth = synth.make_region(L, 10, 1002, swalign, P)
h = api.align_create(draft, copy.deepcopy(events), P)
api.score_alignments(h, 10)
hm = api.find_point_mutations(h)      # Refine's list: every point edit of the sequence (8 per base)
nm = int(api.lib.ps_muts_count(hm))
api.muts_destroy(api.score_mutations(h, hm))
api.prof_reset(); api.prof_enable(1)
t = time.time()
for rep in range(3): api.muts_destroy(api.score_mutations(h, hm))
dt = (time.time() - t) / 3
ms, n, _ = api.prof_get("score"); items = api.prof_units("score")
print("scoring_width %s: %d edits x 10 events: %.1f ms per call; k_score %.2f ms per call in %d launches, %.1f ns per (event, edit) item" % (
    P['scoring_width'], nm, 1e3 * dt, ms / 3, n // 3, 1e6 * ms / max(items, 1)))
print("fill", api.prof_get("fill"))
