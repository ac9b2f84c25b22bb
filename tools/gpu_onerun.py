"""One consensus run of a 10 kb / 10x region after a warm-up run's worth of allocation (profiling target, not a test)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_region
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
draft, events, truth = synth.make_region(10000, 10, 1002, swalign, P)
pa = PSAlign(); pa.sequence, pa.events, pa.params = draft, copy.deepcopy(events), dict(P)
t = time.time(); consensus_region(pa, P); print("one consensus run: %.3f s" % (time.time() - t))
