"""Full-size GPU checks of BASELINE configs #3 and #4 through size-independent properties (the oracle would need
hours at these sizes)."""
import copy
import time

import numpy as np
import pytest

import backends as B
from poreseq_amd import synth
from poreseq_amd.consensus import consensus_region, consensus_regions, merge_seqs, polish, split_regions, variant_region
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS, MutationInfo

pytestmark = pytest.mark.gpu
P0 = dict(DEFAULT_PARAMS, verbose=0)


def test_config3_variant_48kb_30x_10k_point_mutations():
    L, E, M = 48000, 30, 10000
    draft, events, truth = synth.make_region(L, E, 1003, swalign, P0, draft_error=0.0)
    # plant 40 substitutions so that some of the random edits are true reversions
    rng = np.random.default_rng(1003)
    pos = np.sort(rng.choice(np.arange(500, L - 500), 40, replace=False))
    bad = list(draft)
    for p in pos:
        bad[p] = "ACGT"[("ACGT".index(bad[p]) + 1) % 4]
    bad = "".join(bad)
    muts = synth.random_point_mutations(rng, bad, M - 40)
    for p in pos:
        m = MutationInfo(); m.start, m.orig, m.mut = int(p), bad[p], draft[p]
        muts.append(m)
    pa = B.make_pa(PSAlign, bad, copy.deepcopy(events), P0)
    s1 = np.array([m.score for m in pa.ScoreMutations(copy.deepcopy(muts))])
    s2 = np.array([m.score for m in pa.ScoreMutations(copy.deepcopy(muts))])
    assert np.array_equal(s1, s2)                                   # deterministic, no state leaks between calls
    assert (s1[-40:] > 50).all()                                    # every planted error is a strongly positive edit
    assert (s1[:-40] > 0).mean() < 0.02                             # random edits of a correct sequence almost never help
    # independence: an edit's score does not depend on what else is in the list (bit-exact)
    idx = rng.choice(M, 300, replace=False)
    sub = np.array([m.score for m in pa.ScoreMutations([copy.deepcopy(muts[k]) for k in idx])])
    assert np.array_equal(sub, s1[idx])
    # additivity over events: score + 1e-6 is the event-ordered sum of per-event deltas
    parts = np.zeros(len(idx))
    for lo in (0, 10, 20):
        pe = B.make_pa(PSAlign, bad, copy.deepcopy(events[lo:lo + 10]), P0)
        parts += np.array([m.score for m in pe.ScoreMutations([copy.deepcopy(muts[k]) for k in idx])]) + 1e-6
    assert np.allclose(parts - 1e-6, sub, rtol=1e-9, atol=1e-9)
    # the variant driver prints absolute coordinates
    out = variant_region(pa, [copy.deepcopy(m) for m in muts[-3:]], region_start=0)
    assert [m.start for m in out] == [int(p) for p in pos[-3:]]


def test_config4_lambda_48kb_20x_polished_through_the_driver():
    """config #4 on one GPU: 48.5 kb truth, 20 events per region, 6 overlapping regions (split_fasta.py:94-101) refined in
    lock-step through `polish` (split -> refine_regions -> merge_seqs)"""
    Lg, E = 48500, 20
    rng = np.random.default_rng(1004)
    truth = synth.random_sequence(rng, Lg)
    P = dict(P0)
    P.pop("end_trim")                # keep the full region so neighbours still overlap by 1 kb
    made = {}

    def make(a, b):
        draft, events, t = synth.make_region(b - a, E, 2000 + a, swalign, P, truth=truth[a:b])
        made[(a, b)] = (draft, events, t)
        return B.make_pa(PSAlign, draft, events, P)

    t0 = time.perf_counter()
    whole, parts = polish(truth, make, params=None, region_length=10000, overlap=1000, batch=6)
    dt = time.perf_counter() - t0
    assert [(a, b) for a, b, _, _ in parts] == [(0, 10000), (9000, 19000), (18000, 28000), (27000, 37000), (36000, 46000), (45000, 48500)]
    for a, b, seq, acc in parts:
        assert swalign(seq, truth[a:b])[0] > 99.0, (a, b)
    acc = swalign(whole, truth)[0]
    assert abs(len(whole) - Lg) < 200 and acc > 99.5
    print("config #4 (48.5 kb, 20x, 6 regions, one GPU): %.2f s incl. synthetic data, %.1f %% identity" % (dt, acc))
    # fewer than 5 events: the driver returns the input untouched (Mutate.py:50-53)
    draft, events, t = made[(45000, 48500)]
    pa = B.make_pa(PSAlign, draft, copy.deepcopy(events[:4]), P0)
    assert consensus_region(pa, P0) == (draft, 100)


def test_config5_sample_64_regions_lock_step():
    """config #5 (4.6 Mb = 512 regions of 10 kb at 10x) on a 64-region sample: lock-step batches of 16, every region must come
    out > 99 % identical to its truth; the sample's rate is printed (the 8-GPU part cannot run here)"""
    assert len(split_regions(4600000, 10000)) == 512
    E, R = 10, 64
    regs = [synth.make_region(10000 if k % 8 else 7000 + 300 * (k // 8), E, 5000 + k, swalign, P0) for k in range(R)]   # ragged: every 8th region shorter
    t0 = time.perf_counter()
    out = []
    for k in range(0, R, 16):
        pas = [B.make_pa(PSAlign, d, copy.deepcopy(ev), P0) for d, ev, _ in regs[k:k + 16]]
        out += consensus_regions(pas, P0)
    dt = time.perf_counter() - t0
    kb = sum(len(d) for d, _, _ in regs) / 1000.0
    for (seq, _), (d, ev, t) in zip(out, regs):
        trim = int(P0["end_trim"])
        assert swalign(seq, t[trim:-trim])[0] > 99.0
    print("config #5 sample: %d regions (%.0f kb) in %.1f s = %.1f kb/s on one GPU, one lock-step batch at a time" % (R, kb, dt, kb / dt))


def test_train_on_the_gpu_lock_step_equals_replica_by_replica():
    """`poreseq train` (cmdline.py:246-267): the candidate parameter sets as ONE lock-step batch pick the same winner, with the
    same accuracies, as 16 separate runs of the oracle from fresh random streams"""
    import random
    from poreseq_amd.consensus import train
    from poreseq_amd.util import VaryParams
    draft, events, truth = synth.make_region(260, 6, 91, B.oracle_swalign, P0)

    def loader(cls):
        def make_pa(p):
            evs = copy.deepcopy(events)
            for e in evs:
                e.setparams(p)
            return B.make_pa(cls, draft, evs, p)
        return make_pa

    random.seed(11)
    sets = VaryParams(P0)[:6] + [dict(P0, skip_t=0.6, skip_c=0.6, stay_t=0.5, stay_c=0.5)]
    best, accs = train(loader(PSAlign), P0, truth, iters=1, reps=2, paramlists=[sets], lock_step=True)
    each = []
    for p in sets:
        B.reset_rand()
        each.append(consensus_region(loader(B.OraclePSAlign)(p), p, reps=2, refseq=truth)[1])
    assert accs[0] == max(each) and best == sets[int(np.argmax(each))]
