"""Synthetic nanopore event generator — numpy `default_rng`, seeded.

There is no network and no fast5 data, so benchmarks and parity tests run on synthetic
reads with the statistics of R7 events: each read ("event") carries its own 1024-entry
5-mer model, walks the truth sequence's 5-mer states with stay / skip moves and emits a
(mean, stdv) level per visit.  10x coverage = 10 full-span events (5 template-like,
5 complement-like parameter sets), by `PSAlign.Coverage()`'s own definition.
"""
import numpy as np

from .events import PSEvent, PSModel
from .util import DEFAULT_PARAMS

_B = "ACGT"


def random_sequence(rng, n):
    return "".join(_B[i] for i in rng.integers(0, 4, n))


def corrupt(rng, seq, p_del, p_sub, p_ins):
    """iid deletions / substitutions / insertions."""
    out = []
    for c in seq:
        r = rng.random()
        if r < p_del:
            continue
        if r < p_del + p_sub:
            out.append(_B[(_B.index(c) + 1 + rng.integers(0, 3)) % 4])
        else:
            out.append(c)
        if rng.random() < p_ins:
            out.append(_B[rng.integers(0, 4)])
    return "".join(out)


def states_of(seq):
    """5-mer state list of an ACGT-only string (cpp/Sequence.h:84-99)."""
    code = np.frombuffer(seq.encode("ascii"), dtype=np.uint8)
    lut = np.zeros(256, dtype=np.int64)
    for i, c in enumerate(_B):
        lut[ord(c)] = i
    b = lut[code]
    if b.size < 5:
        return np.zeros(0, dtype=np.int64)
    return (b[:-4] << 8) | (b[1:-3] << 6) | (b[2:-2] << 4) | (b[3:-1] << 2) | b[4:]


def make_model(rng, complement):
    m = PSModel()
    m.level_mean = rng.normal(65.0, 12.0, 1024)
    m.level_stdv = rng.uniform(0.8, 2.0, 1024)
    m.sd_mean = rng.uniform(0.8, 1.6, 1024)
    m.sd_stdv = rng.uniform(0.2, 0.5, 1024)
    m.complement = bool(complement)
    return m


def simulate_event(rng, states, model, p_stay=0.05, p_skip=0.10):
    mean, stdv, ral = [], [], []
    j = 0
    C = len(states)
    while j < C:
        s = states[j]
        mean.append(rng.normal(model.level_mean[s], model.level_stdv[s]))
        stdv.append(max(0.3, rng.normal(model.sd_mean[s], 0.2)))
        ral.append(j + 1)
        r = rng.random()
        if r < p_stay:
            pass
        elif r < p_stay + p_skip:
            j += 2
        else:
            j += 1
    return np.array(mean), np.array(stdv), np.array(ral, dtype=np.float64)


def make_region(length, n_events, seed, swalign, params=None, draft_error=0.04, read_error=0.05, truth=None):
    """Build (draft sequence, events, truth) for one region.

    `swalign(seq1, seq2) -> (accuracy, pairs)` re-maps the events' truth alignment onto the
    draft with the reference's `mapaligns` rule; pass `poreseq_amd.poreseqcpp.swalign` (GPU)
    or, when generating golden fixtures, the reference's own.  draft_error is the per-type
    (del / sub / ins) rate: 0.04 gives a ~90 % accurate draft; 0 keeps the truth as draft.
    """
    params = dict(DEFAULT_PARAMS if params is None else params)
    root = np.random.SeedSequence(seed)
    kids = root.spawn(n_events + 2)
    rng0 = np.random.default_rng(kids[0])
    if truth is None:
        truth = random_sequence(rng0, length)   # else: a slice of a longer genome (overlapping region work-items)
    st = states_of(truth)
    events = []
    for e in range(n_events):
        rng = np.random.default_rng(kids[2 + e])
        model = make_model(rng, complement=(e % 2 == 1))
        mean, stdv, ral = simulate_event(rng, st, model)
        ev = PSEvent(mean, stdv, ral, np.zeros(mean.size),
                     sequence=corrupt(rng, truth, read_error, read_error, read_error), model=model)
        ev.setparams(params)
        events.append(ev)
    rngd = np.random.default_rng(kids[1])
    if draft_error > 0:
        draft = corrupt(rngd, truth, draft_error, draft_error, draft_error)
        pairs = np.array(swalign(truth, draft)[1])
        for ev in events:
            ev.mapaligns(pairs)
    else:
        draft = truth
    return draft, events, truth


def random_point_mutations(rng, sequence, n):
    """n random single-base edits: uniform position, del / sub / ins in ratio 1:3:4 (config #3)."""
    from .util import MutationInfo
    L = len(sequence)
    out = []
    for _ in range(n):
        m = MutationInfo()
        m.start = int(rng.integers(0, L - 4))
        k = int(rng.integers(0, 8))
        c = sequence[m.start]
        if k == 0:
            m.orig, m.mut = c, ""
        elif k < 4:
            m.orig, m.mut = c, _B[(_B.index(c) + k) % 4]
        else:
            m.orig, m.mut = "", _B[k - 4]
        out.append(m)
    return out
