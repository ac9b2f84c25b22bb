"""Time the realign fill (ScoreEvents = one forward fill of 10 events) per launch (not a test)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
P = dict(DEFAULT_PARAMS, verbose=0)
if os.environ.get('PS_RW'): P['realign_width'] = int(os.environ['PS_RW'])
api = _capi.load_hip()
draft, events, truth = synth.make_region(L, 10, 1002, swalign, P)
h = api.align_create(draft, copy.deepcopy(events), P)
api.score_alignments(h, 10)
if os.environ.get("PS_MODE") == "score" or (len(sys.argv) > 2 and sys.argv[2] == "score"):     # realign with both directions (ScoreMutations on a short list)
    rng = __import__("numpy").random.default_rng(1)
    muts = synth.random_point_mutations(rng, draft, 50)
    hm = api.muts_create(muts)
    for rep in range(6):
        api.muts_destroy(api.score_mutations(h, hm))
    print("score mode done")
    sys.exit(0)
api.prof_reset(); api.prof_enable(True)
t = time.time()
for rep in range(5):
    api.score_alignments(h, 10)
print("RW=%s P>=%s  score_alignments %.2f ms/call; fill prof (ms, launches, bytes) %s" % (
    P["realign_width"], os.environ.get("PORESEQ_DEBUG_MIN_P", "auto"), 1e3 * (time.time() - t) / 5, api.prof_get("fill")))
