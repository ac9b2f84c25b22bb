"""Full-size GPU checks of BASELINE configs #3 and #4 through size-independent properties (the oracle would need
hours at these sizes)."""
import copy

import numpy as np
import pytest

import backends as B
from poreseq_amd import synth
from poreseq_amd.consensus import consensus_region, merge_seqs, split_regions, variant_region
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


def test_config4_lambda_size_regions_refined_and_stitched():
    """48.5 kb truth -> 6 overlapping regions (split_fasta.py:94-101) -> consensus per region -> merge_seqs."""
    Lg, E = 48500, 10
    rng = np.random.default_rng(1004)
    truth = synth.random_sequence(rng, Lg)
    regions = split_regions(Lg, 10000)
    assert regions == [(0, 10000), (9000, 19000), (18000, 28000), (27000, 37000), (36000, 46000), (45000, 48500)]
    P = dict(P0, end_trim=0.0)       # keep the full region so neighbours still overlap by 1 kb
    pieces = []
    for k, (a, b) in enumerate(regions):
        draft, events, t = synth.make_region(b - a, E, 2000 + k, swalign, P, truth=truth[a:b])
        pa = B.make_pa(PSAlign, draft, events, P)
        params = dict(P); params.pop("end_trim")
        seq, acc = consensus_region(pa, params, refseq=t)
        assert acc > 99.0, (k, acc)
        pieces.append(seq)
    whole = pieces[0]
    for nxt in pieces[1:]:
        whole = merge_seqs(whole, nxt, 1000)
    acc = swalign(whole, truth)[0]
    assert abs(len(whole) - Lg) < 200 and acc > 99.5
