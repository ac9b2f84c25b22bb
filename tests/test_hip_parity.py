"""GPU parity tests: the HIP path (through the C ABI, via poreseqcpp.PSAlign) against the oracle on
the same seeded inputs and against the reference's golden vectors.  Integer / index / max-plus work
is required bit-exact; the only tolerance is on ViterbiMutate's forward probabilities, which feed
nothing but the stochastic back-traces (checked through the resulting sequences)."""
import copy
import os

import numpy as np
import pytest

import backends as B
import golden_util as G
from poreseq_amd import _capi, synth
from poreseq_amd.poreseqcpp import PSAlign, swalign, seqtostates
from poreseq_amd.util import DEFAULT_PARAMS, MutationInfo

pytestmark = pytest.mark.gpu
P0 = dict(DEFAULT_PARAMS, verbose=0)
SCORE_CASES = ["score_L300_E5", "score_L240_E4_narrow"]


def scores(lst):
    return np.array([s.score for s in lst])


def test_backend_is_the_hip_library():
    assert _capi.load_hip().backend_name() == "hip-gfx950"


@pytest.mark.parametrize("name", SCORE_CASES)
def test_golden_score_events_points_mutations(name):
    z = G.load(name)
    assert np.array_equal(np.array(G.make(PSAlign, z).ScoreEvents()), z["ScoreEvents"])
    got = G.make(PSAlign, z).ScorePoints()
    assert [g.start for g in got] == z["ScorePoints_start"].tolist()
    assert np.array_equal(scores(got), z["ScorePoints_score"])
    got = G.make(PSAlign, z).ScoreMutations(G.muts_of(z))
    assert np.array_equal(scores(got), z["ScoreMutations_score"])


def test_golden_consensus_schedule():
    z = G.load("consensus_L400_E6")
    B.reset_rand()
    pa = G.make(PSAlign, z)
    for call, nb, seq in zip(z["calls"], z["nbases"], z["sequences"]):
        call = str(call)
        got = pa.Mutate(reps=4) if call == "Mutate:self" else pa.Mutate(seqs="viterbi") if call == "Mutate:viterbi" else pa.Refine()
        assert got == int(nb), call
        assert pa.sequence == str(seq), call
    for e, ev in enumerate(pa.events):
        assert np.array_equal(ev.ref_align, z["final_ev%d_ref_align" % e])
        assert np.array_equal(ev.ref_like, z["final_ev%d_ref_like" % e])


def test_golden_sw_states():
    z = G.load("sw_states")
    acc, pairs = swalign(str(z["s1"]), str(z["s2"]))
    assert acc == float(z["accuracy"])
    assert np.array_equal(np.array(pairs, dtype=np.int32), z["pairs"])
    assert seqtostates(str(z["odd"])) == z["odd_states"].tolist()


@pytest.mark.parametrize("L,E,seed,par", [
    (300, 5, 11, P0),
    (700, 6, 12, P0),
    (260, 4, 13, dict(P0, realign_width=33.0, scoring_width=9.0, point_width=4.0)),
    (1000, 5, 1001, P0),   # BASELINE config #1 shape
])
def test_dp_matrices_bit_exact(L, E, seed, par):
    """K1/K2 parity: full forward / backward main + stay matrices and step codes vs the oracle."""
    draft, events, truth = synth.make_region(L, E, seed, B.oracle_swalign, par)
    hip, orc = _capi.load_hip(), B.oracle_api()
    for d in (0, 1):
        for e in (0, E - 1):
            outs = []
            for api in (orc, hip):
                h = api.align_create(draft, copy.deepcopy(events), par)
                outs.append(api.debug_fill(h, e, d, events[e].mean.size, len(draft) - 4))
                api.align_destroy(h)
            for k, (x, y) in enumerate(zip(*outs)):
                if d == 1 and k >= 2:
                    continue   # backward step codes are not kept (nothing reads them)
                assert np.array_equal(x, y, equal_nan=True), (d, e, k)


@pytest.mark.parametrize("L,E,seed,par", [
    (300, 5, 21, P0),
    (500, 10, 22, P0),
    (260, 4, 23, dict(P0, realign_width=33.0, scoring_width=9.0, point_width=4.0)),
])
def test_api_parity_with_oracle(L, E, seed, par):
    draft, events, truth = synth.make_region(L, E, seed, B.oracle_swalign, par)
    mk = lambda cls: B.make_pa(cls, draft, copy.deepcopy(events), par)
    assert mk(PSAlign).ScoreEvents() == mk(B.OraclePSAlign).ScoreEvents()
    a, b = mk(PSAlign).ScorePoints(), mk(B.OraclePSAlign).ScorePoints()
    assert np.array_equal(scores(a), scores(b))
    rng = np.random.default_rng(seed)
    muts = synth.random_point_mutations(rng, draft, 60)
    for st, o, m in [(10, draft[10:13], "ACGTA"), (40, draft[40:52], ""), (60, "", "ACGTACGTACGTACGTACGTAC"),
                     (L - 3, draft[L - 3:L - 2], "G"), (len(draft), "", "A"), (len(draft) + 3, "", "A"), (0, "", "TT"),
                     (100, draft[100:101], "T" * 70), (5, draft[5:9], "G" * 30)]:
        mi = MutationInfo(); mi.start, mi.orig, mi.mut = st, o, m
        muts.append(mi)
    a, b = mk(PSAlign).ScoreMutations(muts), mk(B.OraclePSAlign).ScoreMutations(muts)
    assert np.array_equal(scores(a), scores(b))
    # in-place calls: Refine, Mutate(list), ApplyMuts
    x, y = mk(PSAlign), mk(B.OraclePSAlign)
    assert x.Refine() == y.Refine() and x.sequence == y.sequence
    for u, v in zip(x.events, y.events):
        assert np.array_equal(u.ref_align, v.ref_align) and np.array_equal(u.ref_like, v.ref_like)
    x, y = mk(PSAlign), mk(B.OraclePSAlign)
    seeds = [ev.sequence for ev in events[:3]]
    assert x.Mutate(seqs=seeds, reps=3) == y.Mutate(seqs=seeds, reps=3) and x.sequence == y.sequence
    x, y = mk(PSAlign), mk(B.OraclePSAlign)
    sc = mk(B.OraclePSAlign).ScorePoints()
    x.ApplyMuts(sc); y.ApplyMuts(sc)
    assert x.sequence == y.sequence


def test_full_consensus_schedule_matches_oracle():
    """Row H: Mutate('self') then {Mutate('viterbi'), Refine()} until no change — identical edits."""
    draft, events, truth = synth.make_region(600, 10, 31, B.oracle_swalign, P0)
    res = []
    for cls in (PSAlign, B.OraclePSAlign):
        B.reset_rand()
        pa = B.make_pa(cls, draft, copy.deepcopy(events), P0)
        log = [pa.Mutate(reps=4), pa.sequence]
        for _ in range(4):
            log += [pa.Mutate(seqs="viterbi"), pa.sequence]
            nb = pa.Refine()
            log += [nb, pa.sequence]
            if nb == 0:
                break
        res.append(log)
    assert res[0] == res[1]


def test_viterbi_seeds_match_oracle():
    draft, events, truth = synth.make_region(400, 8, 41, B.oracle_swalign, P0)
    for nkeep in (0, 16):
        out = []
        for api in (_capi.load_hip(), B.oracle_api()):
            B.reset_rand()
            h = api.align_create(draft, copy.deepcopy(events), P0)
            out.append(api.viterbi_mutate(h, nkeep, 0.05, 0.01, 0.33, 0.75, 0))
            api.align_destroy(h)
        assert out[0] == out[1], nkeep


def test_viterbi_stochastic_seeds_sweep():
    """60 (region, generator seed) pairs x 16 stochastic back-traces each: the forward weights of ViterbiMutate are computed with
    device exp / log and tree sums (a few ulp from the reference's serial sums, ps_viterbi.hip), so a back-trace could in
    principle flip where a deviate lands within an ulp of a boundary.  Every one of the 960 seed sequences must equal the oracle's."""
    import ctypes
    rng = np.random.default_rng(2024)
    hip, orc = _capi.load_hip(), B.oracle_api()
    libc = ctypes.CDLL(None)
    bad = []
    for k in range(60):
        L, E = int(rng.integers(150, 420)), int(rng.integers(5, 13))
        draft, events, truth = synth.make_region(L, E, 9000 + k, B.oracle_swalign, P0)
        out = []
        for api in (hip, orc):
            libc.srand(100 + k)          # the oracle draws from libc rand(), the HIP library from its per-thread generator
            hip.srand(100 + k)
            h = api.align_create(draft, copy.deepcopy(events), P0)
            try:
                api.score_alignments(h, E)   # a realign first, as every Mutate call has had (refs from a real backtrace)
                out.append(api.viterbi_mutate(h, 16, 0.05, 0.01, 0.33, 0.75, 0))
            finally:
                api.align_destroy(h)
        assert len(out[0]) == len(out[1]) == 16
        bad += [(k, q) for q in range(16) if out[0][q] != out[1][q]]
    assert not bad, "stochastic Viterbi seeds differ from the oracle: %s" % bad[:10]


def test_sw_parity_and_properties():
    rng = np.random.default_rng(5)
    for n1, n2 in [(1, 1), (5, 300), (300, 5), (64, 256), (65, 257), (700, 900), (2100, 1900)]:
        s1 = synth.random_sequence(rng, n1)
        s2 = synth.corrupt(rng, s1, 0.05, 0.05, 0.05) if n2 >= n1 // 2 and n1 > 10 else synth.random_sequence(rng, n2)
        a, b = swalign(s1, s2), B.oracle_swalign(s1, s2)
        assert a[1] == b[1]
        assert (a[0] == b[0]) or (np.isnan(a[0]) and np.isnan(b[0]))
    s = synth.random_sequence(rng, 3000)          # identity: full-length diagonal, 100 %
    acc, pairs = swalign(s, s)
    assert acc == 100.0 and pairs == [(i, i) for i in range(1, 3001)]
    assert swalign("", "ACGT")[1] == []


def test_sw_all_strip_widths_and_super_strips_match_oracle():
    """Both column-per-lane variants of the fill (4 / 8), several super-strips, long near-identical pairs (scores of 60 000),
    repeats (many equal maxima: the column-major first one must start the traceback) and unrelated sequences."""
    rng = np.random.default_rng(77)
    cases = []
    for n1, n2 in [(4100, 4096), (4096, 4097), (6000, 8192), (3000, 8193), (9000, 9500), (2500, 16400), (1300, 17000)]:
        s1 = synth.random_sequence(rng, n1)
        s2 = synth.corrupt(rng, s1, 0.04, 0.04, 0.04)
        s2 = (s2 + synth.random_sequence(rng, n2))[:n2]
        cases.append((s1, s2))
    unit = synth.random_sequence(rng, 37)
    cases.append((unit * 60, unit * 130))                       # tandem repeat: ties everywhere
    cases.append((synth.random_sequence(rng, 5000), synth.random_sequence(rng, 5200)))   # unrelated
    cases.append(("ACGT" * 300, "TGCA" * 1100))
    s1 = synth.random_sequence(rng, 12084)                      # long, nearly identical: scores of 60 000
    cases.append((s1, synth.corrupt(rng, s1, 0.002, 0.002, 0.002)))
    cases.append((s1 + "ACGTT", s1 + "ACGTT"))
    cases.append(("A" * 2600, "A" * 2700))                      # a whole super-strip of ties and a perfect diagonal
    want = [B.oracle_swalign(s1, s2) for s1, s2 in cases]
    # the 4-, 8- and 16-columns-per-lane builds of the chained form, the packed 16-bit fill (the 8-column build's default: two columns per
    # register; the 12 084-base pairs reach scores of 60 000) and its 32-bit twin, and one 16-wave workgroup per pair (8 / 16 columns per
    # lane by length; the 17 000-column case exceeds it and takes the chain)
    for env in ({"PORESEQ_SW_K": "4"}, {"PORESEQ_SW_K": "8"}, {"PORESEQ_SW_K": "8", "PORESEQ_SW_PK": "0"}, {"PORESEQ_SW_K": "16"}, {"PORESEQ_SW_FORM": "one"}):
        os.environ.update(env)
        try:
            for (s1, s2), b in zip(cases, want):
                a = swalign(s1, s2)
                assert a[1] == b[1], (len(s1), len(s2), env)
                assert (a[0] == b[0]) or (np.isnan(a[0]) and np.isnan(b[0]))
        finally:
            for k in env:
                os.environ.pop(k, None)


def _edit(st, o, m):
    mi = MutationInfo(); mi.start, mi.orig, mi.mut = st, o, m
    return mi


@pytest.mark.parametrize("L,E,par", [(3000, 6, P0), (900, 5, dict(P0, realign_width=45.0, scoring_width=12.0)), (10000, 4, P0)])
def test_column_sparse_score_mutations_equals_full_matrices_and_oracle(L, E, par):
    """ScoreMutations with a short edit list keeps only the matrix columns the edits read (k_sweeps): same scores as the
    full-matrix path (k_fill) and the oracle — edits at both ends of the sequence, past the end (skipped: -1e-6), multi-base
    insertions / deletions, one longer than a 64-column chunk, columns next to an invalid 5-mer, an inert event, an event that
    barely aligns, and ref_align / ref_like after the call."""
    draft, events, truth = synth.make_region(L, E, 4100 + L, B.oracle_swalign, par)
    ev = copy.deepcopy(events)
    ev[1].ref_align[:] = 0                       # inert event
    ev[2].ref_align[40:] = 0                     # barely aligned event
    seq = draft[:L // 2] + "N" + draft[L // 2 + 1:]
    n = len(seq)
    rng = np.random.default_rng(L)
    muts = synth.random_point_mutations(rng, seq, 24)
    muts += [_edit(0, "", "TT"), _edit(0, seq[0:1], ""), _edit(1, seq[1:2], "G"), _edit(3, seq[3:5], "A"), _edit(4, "", "C"), _edit(5, seq[5:9], "G" * 30),
             _edit(n - 1, seq[n - 1:], "A"), _edit(n - 3, seq[n - 3:n - 2], "G"), _edit(n - 6, seq[n - 6:n - 2], ""), _edit(n, "", "A"), _edit(n + 3, "", "A"),
             _edit(n - 9, "", "ACGTACG"), _edit(L // 2 - 2, seq[L // 2 - 2:L // 2 - 1], "T"), _edit(L // 2 + 1, "", "GG"), _edit(L // 2 - 7, seq[L // 2 - 7:L // 2 + 3], "ACG"),
             _edit(L // 3, seq[L // 3:L // 3 + 1], "T" * 70), _edit(L // 4, seq[L // 4:L // 4 + 40], ""), _edit(2 * L // 3, "", "ACGTACGTACGTACGTACGTAC")]
    api = _capi.load_hip()
    mk = lambda cls: B.make_pa(cls, seq, copy.deepcopy(ev), par)
    res = []
    for mode in ("sparse", "full", "oracle"):
        api.set_sparse_min(0 if mode == "sparse" else 1 << 30)
        try:
            pa = mk(B.OraclePSAlign if mode == "oracle" else PSAlign)
            if mode != "oracle":
                api.prof_enable(1); api.prof_reset()
            got = pa.ScoreMutations(muts)
            if mode != "oracle":
                sweeps, fills = api.prof_get("sweep")[1], api.prof_get("fill")[1]
                api.prof_enable(0)
                assert (sweeps, fills) == ((1, 0) if mode == "sparse" else (0, 1)), (mode, sweeps, fills)   # the path under test did run
            res.append((scores(got), [e.ref_align.copy() for e in pa.events], [e.ref_like.copy() for e in pa.events]))
        finally:
            api.set_sparse_min(-1)
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0])
        for x, y in zip(res[0][1] + res[0][2], other[1] + other[2]):
            assert np.array_equal(x, y)
    assert res[0][0][24 + 10] == -1e-6           # the edit past the end is skipped (cpp/MakeMutations.cpp:46-47)


def test_mutate_with_unalignable_and_duplicate_seeds_matches_oracle():
    """explicit seed lists: a seed that does not align at all (its events get no ref_index: inert sweeps next to live ones in one
    workgroup pair), the same seed twice (likelihood cache), an odd number of seeds (an unpaired sweep)"""
    draft, events, truth = synth.make_region(500, 6, 501, B.oracle_swalign, P0)
    rng = np.random.default_rng(9)
    seeds = [truth, "ACGT" * 6, synth.corrupt(rng, truth, 0.03, 0.03, 0.03), truth, synth.random_sequence(rng, 300)]
    seeds += [synth.corrupt(rng, truth, 0.02, 0.02, 0.02) for _ in range(22)]      # 27 distinct-ish seeds x 6 events = 162 sweeps: paired workgroups
    res = []
    for cls in (PSAlign, B.OraclePSAlign):
        pa = B.make_pa(cls, draft, copy.deepcopy(events), P0)
        nb = pa.Mutate(seqs=list(seeds), reps=3)
        res.append((nb, pa.sequence, [ev.ref_align.copy() for ev in pa.events], [ev.ref_like.copy() for ev in pa.events]))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    for x, y in zip(res[0][2] + res[0][3], res[1][2] + res[1][3]):
        assert np.array_equal(x, y)


def test_tiny_sequences_match_oracle():
    """sequences of 5 .. 40 bases (1 .. 36 states; fewer states than the four-state prefetch window of k_fill)"""
    for L, seed in ((5, 1), (6, 2), (8, 3), (12, 4), (40, 5)):
        draft, events, truth = synth.make_region(max(L, 30), 3, 400 + seed, B.oracle_swalign, P0, draft_error=0.0)
        short = draft[:L]
        ev = copy.deepcopy(events)
        for e in ev:
            e.ref_align[e.ref_align > L - 4] = 0
        mk = lambda cls: B.make_pa(cls, short, copy.deepcopy(ev), P0)
        assert mk(PSAlign).ScoreEvents() == mk(B.OraclePSAlign).ScoreEvents(), L
        assert np.array_equal(scores(mk(PSAlign).ScorePoints()), scores(mk(B.OraclePSAlign).ScorePoints())), L


@pytest.mark.parametrize("layout", ["auto", "pairs", "plain", "plain-pairs"])
def test_bands_that_jump_and_resume_match_oracle(layout, monkeypatch):
    """Narrow bands on events whose ref_align jumps (levels cut out, alignments shifted): the band leaves empty anti-diagonals
    behind and resumes further down — k_fill's SLOW bodies in the middle of a sweep, rows jumping by more than P — against the oracle,
    matrices included.  Every build of k_fill sees them: what the launcher picks for sweeps this small (lone, compact LDS layout,
    slow bodies derived from the streamed band ends), two sweeps per workgroup, and the plain layout (LDS bitmap of slow bodies) that
    sweeps of more than 256 lanes take, alone and in pairs."""
    if "pairs" in layout:
        monkeypatch.setenv("PORESEQ_DEBUG_PAIR_MIN", "0")
    if "plain" in layout:
        monkeypatch.setenv("PORESEQ_DEBUG_MIN_P", "320")
    rng = np.random.default_rng(5)
    for W, seed in ((3, 1), (5, 2), (8, 3), (17, 4)):
        Pn = dict(P0, realign_width=float(W), scoring_width=4.0, point_width=3.0)
        draft, events, truth = synth.make_region(420, 4, 300 + seed, B.oracle_swalign, Pn)
        ev = copy.deepcopy(events)
        for e in ev:
            n = e.ref_align.size
            a, b = sorted(rng.integers(20, n - 20, 2))
            e.ref_align[a:b] = 0                                   # an unaligned stretch: interpolated ref_index
            c = int(rng.integers(20, n - 60))
            e.ref_align[c:c + 40] = np.minimum(e.ref_align[c:c + 40] + 90, len(draft) - 5) * (e.ref_align[c:c + 40] > 0)   # a jump forward
        mk = lambda cls: B.make_pa(cls, draft, copy.deepcopy(ev), Pn)
        assert mk(PSAlign).ScoreEvents() == mk(B.OraclePSAlign).ScoreEvents(), W
        assert np.array_equal(scores(mk(PSAlign).ScorePoints()), scores(mk(B.OraclePSAlign).ScorePoints())), W
        hip, orc = _capi.load_hip(), B.oracle_api()
        for d in (0, 1):
            out = []
            for api in (hip, orc):
                h = api.align_create(draft, copy.deepcopy(ev), Pn)
                out.append(api.debug_fill(h, 1, d, ev[1].mean.size, len(draft) - 4))
                api.align_destroy(h)
            for x, y in zip(out[0][:2], out[1][:2]):
                assert np.array_equal(x, y, equal_nan=True), (W, d)
            if d == 0:
                assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3]), W


def test_wide_band_and_many_events_match_oracle():
    """realign_width 600 (footprint ~620 rows: a 640-lane sweep) and 80 events through ViterbiMutate (the deep-stack build
    of k_vit_obs), both against the oracle"""
    Pw = dict(P0, realign_width=600.0)
    draft, events, truth = synth.make_region(1500, 3, 81, B.oracle_swalign, Pw)
    mk = lambda cls: B.make_pa(cls, draft, copy.deepcopy(events), Pw)
    assert mk(PSAlign).ScoreEvents() == mk(B.OraclePSAlign).ScoreEvents()
    rng = np.random.default_rng(3)
    muts = synth.random_point_mutations(rng, draft, 60)
    assert np.array_equal(scores(mk(PSAlign).ScoreMutations(muts)), scores(mk(B.OraclePSAlign).ScoreMutations(muts)))
    # realign_width 1000 on a 3 kb region: ~1030 rows per anti-diagonal, more than one lane per row -> k_fill_wide (two slots per thread)
    Pv = dict(P0, realign_width=1000.0)
    draft, events, truth = synth.make_region(3000, 2, 83, B.oracle_swalign, Pv)
    mk = lambda cls: B.make_pa(cls, draft, copy.deepcopy(events), Pv)
    assert mk(PSAlign).ScoreEvents() == mk(B.OraclePSAlign).ScoreEvents()
    muts = synth.random_point_mutations(rng, draft, 40)
    assert np.array_equal(scores(mk(PSAlign).ScoreMutations(muts)), scores(mk(B.OraclePSAlign).ScoreMutations(muts)))
    hip, orc = _capi.load_hip(), B.oracle_api()
    for d in (0, 1):
        out = []
        for api in (hip, orc):
            h = api.align_create(draft, copy.deepcopy(events), Pv)
            out.append(api.debug_fill(h, 0, d, events[0].mean.size, len(draft) - 4))
            api.align_destroy(h)
        assert np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1], equal_nan=True), d
        if d == 0:
            assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])
    draft, events, truth = synth.make_region(180, 80, 82, B.oracle_swalign, P0)
    res = []
    for cls in (PSAlign, B.OraclePSAlign):
        B.reset_rand()
        pa = B.make_pa(cls, draft, copy.deepcopy(events), P0)
        res.append((pa.Mutate(seqs="viterbi", reps=1), pa.sequence))
    assert res[0] == res[1]


@pytest.mark.parametrize("width,L", [(45, 400), (150, 900), (430, 1600), (800, 2600)])
def test_every_strip_height_matches_oracle(width, L):
    """the strip sweeps' heights K = 4 / 6 / 16 / 32 rows per lane (chosen from the band's widest window; the default width
    300 -> K = 10 and width 600 -> K = 24 run elsewhere): DP matrices with step codes through the debug hook (forward +
    backward strip records when the strip sweeps are on), ScoreEvents, ScoreMutations and a Mutate against the oracle"""
    Pw = dict(P0, realign_width=float(width))
    draft, events, truth = synth.make_region(L, 4, 300 + width, B.oracle_swalign, Pw)
    mk = lambda cls: B.make_pa(cls, draft, copy.deepcopy(events), Pw)
    assert mk(PSAlign).ScoreEvents() == mk(B.OraclePSAlign).ScoreEvents()
    muts = synth.random_point_mutations(np.random.default_rng(width), draft, 80)
    assert np.array_equal(scores(mk(PSAlign).ScoreMutations(muts)), scores(mk(B.OraclePSAlign).ScoreMutations(muts)))
    hip, orc = _capi.load_hip(), B.oracle_api()
    for d in (0, 1):
        out = []
        for api in (hip, orc):
            h = api.align_create(draft, copy.deepcopy(events), Pw)
            out.append(api.debug_fill(h, 1, d, events[1].mean.size, len(draft) - 4))
            api.align_destroy(h)
        assert np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1], equal_nan=True), d
        if d == 0:
            assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])
    res = []
    for cls in (PSAlign, B.OraclePSAlign):
        B.reset_rand()
        pa = mk(cls)
        res.append((pa.Mutate(reps=2), pa.sequence, [e.ref_like.copy() for e in pa.events]))
    assert res[0][:2] == res[1][:2]
    assert all(np.array_equal(a, b) for a, b in zip(res[0][2], res[1][2]))


def test_realign_to_and_copy():
    draft, events, truth = synth.make_region(300, 4, 61, B.oracle_swalign, P0)
    x, y = B.make_pa(PSAlign, draft, copy.deepcopy(events), P0), B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(events), P0)
    xc = x.Copy()
    x.RealignTo(truth); y.RealignTo(truth)
    assert x.sequence == truth and xc.sequence == draft
    assert x.ScoreEvents() == y.ScoreEvents()
    assert np.array_equal(x.Coverage(), y.Coverage())


def test_large_properties_config2_shape():
    """BASELINE config #2 size (10 kb, 10 events): size-independent properties instead of the oracle."""
    draft, events, truth = synth.make_region(10000, 10, 1002, swalign, P0, draft_error=0.0)
    pa = B.make_pa(PSAlign, draft, copy.deepcopy(events), P0)
    s1 = pa.ScoreEvents()
    assert s1 == pa.ScoreEvents()                      # idempotent / deterministic
    assert all(2.0 * len(draft) < s < 3.0 * len(draft) for s in s1)   # ~2.2-2.4 per base
    # additivity over events: scoring a subset of events gives the same per-event scores
    sub = B.make_pa(PSAlign, draft, copy.deepcopy(events[3:6]), P0).ScoreEvents()
    assert sub == s1[3:6]
    # planted errors are found: corrupt 3 bases, the reverting edits get the top positive scores
    bad = list(draft)
    for p in (2000, 5000, 8000):
        bad[p] = "ACGT"[("ACGT".index(bad[p]) + 1) % 4]
    bad = "".join(bad)
    pb = B.make_pa(PSAlign, bad, copy.deepcopy(events), P0)
    muts = []
    for p in (2000, 5000, 8000):
        m = MutationInfo(); m.start, m.orig, m.mut = p, bad[p], draft[p]
        muts.append(m)
    m = MutationInfo(); m.start, m.orig, m.mut = 3000, bad[3000], "ACGT"[("ACGT".index(bad[3000]) + 1) % 4]
    muts.append(m)
    sc = scores(pb.ScoreMutations(muts))
    assert (sc[:3] > 20).all() and sc[3] < 0
    nb = pb.Refine()
    assert nb >= 3 and draft[300:-300] in pb.sequence      # the planted errors are repaired (ends may legitimately move)


def test_profile_modes_count_the_same_launches():
    """ps_prof_enable(1) (event pair read after every launch) and (2) (pairs queued, read by ps_prof_get: what bench.py uses
    inside its timed steps) see the same launches, algorithmic bytes and work units; both measure a positive time"""
    draft, events, truth = synth.make_region(500, 6, 31, B.oracle_swalign, P0)
    api = _capi.load_hip()
    rng = np.random.default_rng(3)
    muts = synth.random_point_mutations(rng, draft, 40)
    seen = {}
    for mode in (1, 2):
        h = api.align_create(draft, copy.deepcopy(events), P0)
        hm = api.muts_create(muts)
        api.prof_reset(); api.prof_enable(mode)
        want = api.score_alignments(h, len(events))
        api.muts_destroy(api.score_mutations(h, hm))
        ms, n, nbytes = api.prof_get("fill")
        ms2, n2, _ = api.prof_get("score")
        seen[mode] = (n, nbytes, api.prof_units("fill"), n2, api.prof_units("score"), list(want))
        api.prof_enable(0)
        assert ms > 0 and ms2 > 0 and n >= 2 and n2 >= 1
        api.muts_destroy(hm); api.align_destroy(h)
    assert seen[1] == seen[2]
