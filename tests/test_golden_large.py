"""BASELINE-sized golden vectors from the real reference (tests/golden/make_golden_large.py):

  score10k_E10 / score10k_E30   ScoreEvents + ScoreMutations (2000 point + 20 multi-base edits, scoring_width 100 and 20)
                                on a 10 kb region with 10 / 30 events (config #2 / one region of config #3)
  consensus_L1000/L1500/L3000   the full Mutate.py schedule (per-call nbases, the sequence after every call, final
                                ref_align / ref_like of every event, final ScoreEvents)
  consensus_L10000_E10          the same at BASELINE config #2's size (final ref_like vectors as digests)
  mutate_seeds_L10000           one Mutate(list of seed strings) call at 10 kb

Inputs are regenerated from the fixture's seed and verified by checksum; every comparison is bit-exact (tolerance 0).
The oracle runs the cheaper cases on CPU (`not gpu`); the HIP library runs all of them (`gpu`).
"""
import copy

import numpy as np
import pytest

import backends as B
import golden_util as G
from poreseq_amd.poreseqcpp import PSAlign, swalign


def _score_case(cls, sw, name, tags=("sw100", "sw20")):
    z = G.load(name)
    draft, events, truth, par = G.regen(z, sw)
    mk = lambda p: B.make_pa(cls, draft, copy.deepcopy(events), p)
    assert np.array_equal(np.array(mk(par).ScoreEvents()), z["ScoreEvents"])
    muts = G.muts_of(z)
    for tag in tags:
        p = dict(par, scoring_width=100.0 if tag == "sw100" else 20.0)
        got = mk(p).ScoreMutations(muts)
        want = z["ScoreMutations_" + tag]
        assert np.array_equal(np.array([g.score for g in got]), want), (name, tag)


def _schedule_case(cls, sw, name):
    z = G.load(name)
    draft, events, truth, par = G.regen(z, sw)
    B.reset_rand()
    pa = B.make_pa(cls, draft, copy.deepcopy(events), par)
    for call, nb, seq in zip(z["calls"], z["nbases"], z["sequences"]):
        call = str(call)
        if call == "Mutate:self":
            got = pa.Mutate(reps=4)
        elif call == "Mutate:viterbi":
            got = pa.Mutate(seqs="viterbi")
        else:
            got = pa.Refine()
        assert got == int(nb), (name, call)
        assert pa.sequence == str(seq), (name, call)
    _final_refs(pa, z, name)
    assert np.array_equal(np.array(pa.ScoreEvents()), z["final_ScoreEvents"])


def _final_refs(pa, z, name):
    """final ref_align / ref_like of every event: stored in full, or (10 kb fixtures) as int32 alignment + SHA-256 of the likelihoods"""
    import hashlib
    for e, ev in enumerate(pa.events):
        if "final_ev%d_ref_align" % e in z.files:
            assert np.array_equal(ev.ref_align, z["final_ev%d_ref_align" % e]), (name, e)
            assert np.array_equal(ev.ref_like, z["final_ev%d_ref_like" % e]), (name, e)
        else:
            assert np.array_equal(ev.ref_align, z["final_ev%d_ref_align_i32" % e].astype(np.float64)), (name, e)
            got = hashlib.sha256(np.ascontiguousarray(ev.ref_like, dtype=np.float64).tobytes()).hexdigest()
            assert got == str(z["final_ev%d_ref_like_sha256" % e]), (name, e)


def _seed_list_case(cls, sw, name):
    """one Mutate(seqs=[caller's seed strings], reps=2) at 10 kb: Smith-Waterman strip chains, candidate batches, greedy application"""
    z = G.load(name)
    draft, events, truth, par = G.regen(z, sw)
    B.reset_rand()
    pa = B.make_pa(cls, draft, copy.deepcopy(events), par)
    got = pa.Mutate(seqs=[str(s) for s in z["seeds"]], reps=2)
    assert got == int(z["nbases"]) and pa.sequence == str(z["sequence"])
    _final_refs(pa, z, name)
    assert np.array_equal(np.array(pa.ScoreEvents()), z["final_ScoreEvents"])


# ---- CPU: the oracle against the reference's vectors (pins the oracle at sizes where the band is narrower than the columns)
def test_oracle_score10k_E10_golden():
    _score_case(B.OraclePSAlign, B.oracle_swalign, "score10k_E10", tags=("sw20",))


def test_oracle_consensus_L1000_golden():
    _schedule_case(B.OraclePSAlign, B.oracle_swalign, "consensus_L1000")


def test_oracle_mutate_seed_list_10kb_golden():
    """pins the oracle's FindMutations / Smith-Waterman / greedy loop at 10 kb against the reference's own run (~40 s)"""
    _seed_list_case(B.OraclePSAlign, B.oracle_swalign, "mutate_seeds_L10000")


# ---- GPU: the HIP library against the same vectors -------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["score10k_E10", "score10k_E30"])
def test_hip_score10k_golden(name):
    _score_case(PSAlign, swalign, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["consensus_L1000", "consensus_L1500", "consensus_L3000", "consensus_L10000_E10"])
def test_hip_consensus_schedule_golden(name, sweep_always):
    """the full Mutate.py:62-93 schedule against the reference's own run — up to BASELINE config #2's size (10 kb, 10 events:
    18 minutes of reference time) — with every forward-only batch on the strip sweep"""
    _schedule_case(PSAlign, swalign, name)


@pytest.mark.gpu
def test_hip_consensus_10kb_golden_default_kernels():
    """the same 10 kb schedule with the default kernel choice (a lone region's batches are below the strip sweep's threshold)"""
    _schedule_case(PSAlign, swalign, "consensus_L10000_E10")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sweep", "default"])
def test_hip_mutate_seed_list_10kb_golden(mode):
    from poreseq_amd import _capi
    api = _capi.load_hip()
    api.set_sweep_min(0 if mode == "sweep" else -1)
    try:
        _seed_list_case(PSAlign, swalign, "mutate_seeds_L10000")
    finally:
        api.set_sweep_min(-1)


# ---- bench.py's own regions: digests of the reference's full schedule (tests/golden/make_golden_bench.py, oracle/_ref) ----
def _bench_region_case(cls, sw, length):
    """The digest bench.py's `parity_in_run` compares every timed step's region 0 with: inputs regenerated and verified by checksum,
    the full schedule from a fresh random stream, SHA-256 of the consensus sequence + the per-call accepted-edit counts."""
    import hashlib, json, os
    from poreseq_amd import synth
    from poreseq_amd.consensus import consensus_region
    from poreseq_amd.util import DEFAULT_PARAMS
    with open(os.path.join(G.GOLDEN, "bench_regions.json")) as fh:
        cases = [g for g in json.load(fh)["regions"] if g["length"] == length]
    assert cases
    for g in cases:
        params = dict(DEFAULT_PARAMS, verbose=0)
        d, ev, tr = synth.make_region(g["length"], g["events"], g["seed"], sw, params)
        assert G.input_digest(d, ev, tr) == g["input_sha256"], "synthetic generator drift"
        pa = B.make_pa(cls, d, ev, params)
        B.reset_rand()
        log = []
        seq, acc = consensus_region(pa, params, log=log)
        assert [c for c, _, _ in log] == g["calls"] and [int(n) for _, n, _ in log] == g["nbases"], (g["seed"], log)
        assert len(seq) == g["sequence_len"] and hashlib.sha256(seq.encode("ascii")).hexdigest() == g["sequence_sha256"], g["seed"]
        assert acc == g["accuracy"]


def test_oracle_bench_region_digest_1kb():
    _bench_region_case(B.OraclePSAlign, B.oracle_swalign, 1000)


@pytest.mark.gpu
@pytest.mark.parametrize("length", [1000, 10000])
def test_hip_bench_region_digests(length):
    """the three 10 kb regions bench.py checks in every timed step (and its 1 kb region) through the HIP library, one region at a time"""
    _bench_region_case(PSAlign, swalign, length)
