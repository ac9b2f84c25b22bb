"""Time the realign fill of R regions in lock-step (not a test):  python tools/gpu_fillbatch.py R [fwd|both] [L]
fwd = ScoreAlignments (one forward sweep per event), both = ScoreMutations on a short list (forward + backward per event).
PORESEQ_DEBUG_PAIR_MIN=<sweeps> moves the launch size from which two sweeps share a workgroup."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import swalign
from poreseq_amd.util import DEFAULT_PARAMS
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
L = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
P = dict(DEFAULT_PARAMS, verbose=0)
if os.environ.get('PS_RW'): P['realign_width'] = int(os.environ['PS_RW'])
api = _capi.load_hip()
regs = [synth.make_region(L, 10, 1002 + k % 3, swalign, P) for k in range(min(R, 3))]
hs = [api.align_create(regs[k % 3][0], copy.deepcopy(regs[k % 3][1]), P) for k in range(R)]
api.batch_score_alignments(hs, [10] * R)
rng = np.random.default_rng(1)
hm = [api.muts_create(synth.random_point_mutations(rng, regs[k % 3][0], 20)) for k in range(R)]
def once():
    if mode == "fwd": api.batch_score_alignments(hs, [10] * R)
    else:
        for m in api.batch_score_mutations(hs, hm): api.muts_destroy(m)
once()
api.prof_reset(); api.prof_enable(True)
t = time.time()
N = 4
for rep in range(N): once()
dt = (time.time() - t) / N
f, s = api.prof_get("fill"), api.prof_get("sweep")
print("R=%d mode=%s form=%s pair_min=%s: %.2f ms/call; fill %.2f ms x %d launches; sweep %.2f ms x %d launches (w2 %d, w4 %d)" % (
    R, mode, os.environ.get("PORESEQ_SWEEP_FORM", "auto"), os.environ.get("PORESEQ_DEBUG_PAIR_MIN", "default"), 1e3 * dt,
    f[0] / max(f[1], 1), f[1], s[0] / max(s[1], 1), s[1], api.prof_get("sweep_w2")[1], api.prof_get("sweep_w4")[1]))
