"""Smith-Waterman timing for several lengths (not a test): per-row time of the fill shows how it scales with K / waves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poreseq_amd import synth, _capi
api = _capi.load_hip()
rng = np.random.default_rng(5)
for L in [int(x) for x in sys.argv[1:]] or [1000, 2000, 4000, 8000, 10000, 16000]:
    a = synth.random_sequence(rng, L); b = synth.corrupt(rng, a, 0.03, 0.03, 0.03)
    api.swfull(a, b)
    t = time.time()
    for _ in range(3): api.swfull(a, b)
    print("L=%d  %.2f ms per swfull  (%.0f ns per row)" % (L, (time.time() - t) / 3 * 1e3, (time.time() - t) / 3 * 1e9 / L))
