"""Consensus schedule of one synthetic region alone (profiling target, not a test): a warm-up run for the pools, then 3 measured."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_regions
from poreseq_amd.util import DEFAULT_PARAMS
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
P = dict(DEFAULT_PARAMS, verbose=0)
draft, events, truth = synth.make_region(L, 10, 1002, swalign, P)
for k in range(4):
    pa = PSAlign(); pa.sequence, pa.events, pa.params = draft, copy.deepcopy(events), dict(P)
    t = time.time(); consensus_regions([pa], P); print("consensus of one %d-base region: %.3f s" % (L, time.time() - t))
