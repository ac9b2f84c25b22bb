import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import backends as B
from poreseq_amd import _capi, synth
from poreseq_amd.util import DEFAULT_PARAMS
P0 = dict(DEFAULT_PARAMS, verbose=0)
L, E, seed = 700, 6, 12
draft, events, truth = synth.make_region(L, E, seed, B.oracle_swalign, P0)
hip, orc = _capi.load_hip(), B.oracle_api()
for d in (0, 1):
    for e in (0, E - 1):
        outs = []
        for api in (orc, hip):
            h = api.align_create(draft, copy.deepcopy(events), P0)
            outs.append(api.debug_fill(h, e, d, events[e].mean.size, len(draft) - 4))
            api.align_destroy(h)
        for k, (x, y) in enumerate(zip(*outs)):
            if d == 1 and k >= 2: continue
            bad = np.argwhere(~((x == y) | (np.isnan(x) & np.isnan(y)))) if x.dtype.kind == 'f' else np.argwhere(x != y)
            if len(bad):
                print("dir", d, "ev", e, "arr", k, "shape", x.shape, "nbad", len(bad), "first", bad[:6].tolist(), "last", bad[-3:].tolist())
                for (a, c) in bad[:4]:
                    print("   at", a, c, "oracle", x[a, c], "hip", y[a, c])
