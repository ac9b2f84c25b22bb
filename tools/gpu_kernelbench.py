"""Run one kernel family in isolation (for rocprofv3 PMC passes): viterbi | fill | sw"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS
what = sys.argv[1]; L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
P = dict(DEFAULT_PARAMS, verbose=0)
api = _capi.load_hip()
draft, events, truth = synth.make_region(L, 10, 1002, swalign, P)
h = api.align_create(draft, copy.deepcopy(events), P)
for rep in range(3):
    t = time.time()
    if what == "viterbi":
        api.viterbi_mutate(h, 16, 0.05, 0.01, 0.33, 0.75, 0)
    elif what == "fill":
        api.score_alignments(h, 10)
    elif what == "sw":
        api.swfull(draft, truth)
    print(what, "%.2f ms" % (1e3 * (time.time() - t)))
