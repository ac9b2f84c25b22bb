"""Load tests/golden/*.npz fixtures (produced by tests/golden/make_golden.py from the real reference)."""
import os

import numpy as np

from poreseq_amd.events import PSEvent, PSModel
from poreseq_amd.util import MutationInfo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def params_of(z):
    return {str(k): float(v) for k, v in zip(z["params_keys"], z["params_vals"])}


def events_of(z):
    evs = []
    for e in range(int(z["n_events"])):
        m = PSModel()
        mdl = z["ev%d_model" % e]
        m.level_mean, m.level_stdv, m.sd_mean, m.sd_stdv = [np.array(mdl[k]) for k in range(4)]
        m.prob_skip, m.prob_stay, m.prob_extend, m.prob_insert = [float(x) for x in z["ev%d_trans" % e]]
        m.complement = bool(z["ev%d_complement" % e])
        evs.append(PSEvent(z["ev%d_mean" % e], z["ev%d_stdv" % e], z["ev%d_ref_align" % e], z["ev%d_ref_like" % e],
                           sequence=str(z["ev%d_sequence" % e]), model=m))
    return evs


def muts_of(z, prefix="muts"):
    out = []
    for s, o, m in zip(z[prefix + "_start"], z[prefix + "_orig"], z[prefix + "_mut"]):
        mi = MutationInfo()
        mi.start, mi.orig, mi.mut = int(s), str(o), str(m)
        out.append(mi)
    return out


def make(cls, z):
    pa = cls()
    pa.sequence = str(z["sequence"])
    pa.events = events_of(z)
    pa.params = params_of(z)
    return pa
