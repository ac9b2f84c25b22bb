import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import backends as B
from poreseq_amd import synth
from poreseq_amd.poreseqcpp import swalign
rng = np.random.default_rng(9)
for n1, n2 in [(20000, 40000), (48500, 47000)]:
    s1 = synth.random_sequence(rng, n1)
    s2 = synth.corrupt(rng, s1, 0.05, 0.05, 0.05)
    s2 = (s2 + synth.random_sequence(rng, n2))[:n2]
    t = time.time(); a = swalign(s1, s2); tg = time.time() - t
    t = time.time(); b = B.oracle_swalign(s1, s2); tc = time.time() - t
    print(n1, n2, "gpu %.2fs oracle %.2fs" % (tg, tc), "equal:", a[1] == b[1] and a[0] == b[0], len(a[1]))
