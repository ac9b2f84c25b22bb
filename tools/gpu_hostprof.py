import copy, os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_region
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
draft, events, truth = synth.make_region(L, 10, 1002, swalign, P)
def run():
    pa = PSAlign(); pa.sequence = draft; pa.events = copy.deepcopy(events); pa.params = dict(P)
    return consensus_region(pa, P)
run()
sys.stderr.write('=== MEASURED RUN ===\n'); sys.stderr.flush()
pr = cProfile.Profile(); t = time.time(); pr.enable(); run(); pr.disable(); print("wall", time.time() - t)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
