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


# ---- BASELINE-sized fixtures (tests/golden/make_golden_large.py): inputs are regenerated, outputs are stored ----
def input_digest(draft, events, truth):
    """SHA-256 over every generated input, in the order make_golden_large.py hashes them."""
    import hashlib
    h = hashlib.sha256()
    h.update(draft.encode("ascii")); h.update(b"|"); h.update(truth.encode("ascii"))
    for ev in events:
        m = ev.model
        for a in (ev.mean, ev.stdv, ev.ref_align, ev.ref_like, m.level_mean, m.level_stdv, m.sd_mean, m.sd_stdv):
            h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        h.update(np.array([m.prob_skip, m.prob_stay, m.prob_extend, m.prob_insert], dtype=np.float64).tobytes())
        h.update(ev.sequence.encode("ascii"))
    return h.hexdigest()


def regen(z, swalign):
    """(draft, events, truth, params) of a large fixture, regenerated from its seed with `swalign` (any bit-exact
    Smith-Waterman: oracle or HIP) and verified against the stored checksum of the reference-side inputs."""
    from poreseq_amd import synth
    par = params_of(z)
    draft, events, truth = synth.make_region(int(z["L"]), int(z["E"]), int(z["seed"]), swalign, par)
    got = input_digest(draft, events, truth)
    if got != str(z["input_sha256"]):
        raise AssertionError("synthetic generator drift: regenerated inputs do not match the fixture's checksum")
    return draft, events, truth, par
