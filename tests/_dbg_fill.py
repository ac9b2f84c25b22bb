import copy, sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
import backends as B
from poreseq_amd import _capi, synth
from poreseq_amd.util import DEFAULT_PARAMS
P0 = dict(DEFAULT_PARAMS, verbose=0)
draft, events, truth = synth.make_region(700, 6, 12, B.oracle_swalign, P0)
hip, orc = _capi.load_hip(), B.oracle_api()
names = ["main", "stay", "sm", "ss"]
for d in (0, 1):
    e = int(os.environ.get("EV","0"))
    outs = []
    for api in (orc, hip):
        h = api.align_create(draft, copy.deepcopy(events), P0)
        outs.append(api.debug_fill(h, e, d, events[e].mean.size, len(draft) - 4)); api.align_destroy(h)
    for nm, x, y in zip(names, *outs):
        x = x.astype(float); y = y.astype(float)
        bad = np.argwhere(~((x == y) | (np.isnan(x) & np.isnan(y))))
        print("dir", d, nm, "mismatches", len(bad), bad[:6].tolist(), [(x[tuple(q)], y[tuple(q)]) for q in bad[:6]])
