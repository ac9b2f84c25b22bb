"""CPU tests: the oracle against the reference's golden vectors (and the live reference build
when oracle/_ref is present), the C ABI exports of the product library, host-side helpers."""
import copy
import ctypes
import os

import numpy as np
import pytest

import backends as B
import golden_util as G
from poreseq_amd import _capi, synth
from poreseq_amd.util import DEFAULT_PARAMS, MutationInfo, MutationScore, RegionInfo, LoadParams, SaveParams

SCORE_CASES = ["score_L300_E5", "score_L240_E4_narrow"]


@pytest.mark.parametrize("name", SCORE_CASES)
def test_oracle_score_events_golden(name):
    z = G.load(name)
    got = G.make(B.OraclePSAlign, z).ScoreEvents()
    assert np.array_equal(np.array(got), z["ScoreEvents"])  # bit-exact


@pytest.mark.parametrize("name", SCORE_CASES)
def test_oracle_score_points_golden(name):
    z = G.load(name)
    got = G.make(B.OraclePSAlign, z).ScorePoints()
    assert [g.start for g in got] == z["ScorePoints_start"].tolist()
    assert [g.orig for g in got] == [str(x) for x in z["ScorePoints_orig"]]
    assert [g.mut for g in got] == [str(x) for x in z["ScorePoints_mut"]]
    assert np.array_equal(np.array([g.score for g in got]), z["ScorePoints_score"])


@pytest.mark.parametrize("name", SCORE_CASES)
def test_oracle_score_mutations_golden(name):
    z = G.load(name)
    got = G.make(B.OraclePSAlign, z).ScoreMutations(G.muts_of(z))
    assert np.array_equal(np.array([g.score for g in got]), z["ScoreMutations_score"])
    assert [g.start for g in got] == z["ScoreMutations_start"].tolist()


def test_oracle_consensus_schedule_golden():
    """Mutate.py:70-85 schedule: per-call nbases, sequence after every call, final refs."""
    z = G.load("consensus_L400_E6")
    B.reset_rand()
    pa = G.make(B.OraclePSAlign, z)
    for call, nb, seq in zip(z["calls"], z["nbases"], z["sequences"]):
        call = str(call)
        if call == "Mutate:self":
            got = pa.Mutate(reps=4)
        elif call == "Mutate:viterbi":
            got = pa.Mutate(seqs="viterbi")
        else:
            got = pa.Refine()
        assert got == int(nb), call
        assert pa.sequence == str(seq), call
    for e, ev in enumerate(pa.events):
        assert np.array_equal(ev.ref_align, z["final_ev%d_ref_align" % e])
        assert np.array_equal(ev.ref_like, z["final_ev%d_ref_like" % e])


def test_oracle_sw_and_states_golden():
    z = G.load("sw_states")
    acc, pairs = B.oracle_swalign(str(z["s1"]), str(z["s2"]))
    assert acc == float(z["accuracy"])
    assert np.array_equal(np.array(pairs, dtype=np.int32), z["pairs"])
    assert B.seqtostates(str(z["odd"]), B.oracle_api) == z["odd_states"].tolist()
    assert B.seqtostates(str(z["s1"]), B.oracle_api) == z["s1_states"].tolist()
    assert B.seqtostates("ACG", B.oracle_api) == []


@pytest.mark.skipif(not B.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_matches_live_reference_full_api():
    P = dict(DEFAULT_PARAMS, verbose=0)
    draft, events, truth = synth.make_region(350, 5, 4242, B.ref_swalign, P)
    logs = {}
    for cls in (B.RefPSAlign, B.OraclePSAlign):
        B.reset_rand()
        pa = B.make_pa(cls, draft, copy.deepcopy(events), P)
        log = [pa.ScoreEvents(), [s.score for s in pa.ScorePoints()]]
        log.append(pa.Mutate(reps=2)); log.append(pa.sequence)
        log.append(pa.Mutate(seqs="viterbi", reps=2)); log.append(pa.sequence)
        log.append(pa.Refine()); log.append(pa.sequence)
        log.append([e.ref_align.tolist() for e in pa.events])
        logs[cls.__name__] = log
    assert logs["RefPSAlign"] == logs["OraclePSAlign"]
    for d in (0, 1):
        outs = []
        for api in (B.ref_api(), B.oracle_api()):
            h = api.align_create(draft, copy.deepcopy(events), P)
            outs.append(api.debug_fill(h, 1, d, events[1].mean.size, len(draft) - 4))
            api.align_destroy(h)
        for x, y in zip(*outs):
            assert np.array_equal(x, y, equal_nan=True)


def test_oracle_edge_cases():
    P = dict(DEFAULT_PARAMS, verbose=0)
    draft, events, truth = synth.make_region(120, 3, 7, B.oracle_swalign, P, draft_error=0.0)
    # an event with no alignment at all is inert: scores 0, refs untouched (Alignment.cpp:51-59)
    ev = copy.deepcopy(events)
    ev[1].ref_align[:] = 0
    pa = B.make_pa(B.OraclePSAlign, draft, ev, P)
    s = pa.ScoreEvents()
    assert s[1] == 0.0 and s[0] > 0
    # no events, tiny sequence
    pa0 = B.make_pa(B.OraclePSAlign, "ACGTACGTAC", [], P)
    assert pa0.ScoreEvents() == []
    assert len(pa0.ScorePoints()) == 8 * 6  # 1 deletion + 3 substitutions + 4 insertions per state
    # edit past the end is skipped: score stays at the -1e-6 seed (MakeMutations.cpp:46-47, AlignUtil.h:86)
    m = MutationInfo(); m.start = len(draft) + 5; m.mut = "A"
    out = B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(events), P).ScoreMutations([m])
    assert out[0].score == -1e-6


def test_library_exports_every_declared_symbol():
    """The shipped .so must load (no GPU needed) and export every entry point of include/poreseq_hip.h."""
    import re
    hdr = open(os.path.join(B.ROOT, "include", "poreseq_hip.h")).read()
    declared = set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    lib = ctypes.CDLL(_capi.HIP_LIB)
    for name in declared:
        assert hasattr(lib, name), name
    api = _capi.load_hip()
    assert api.backend_name() == "hip-gfx950"


def test_hip_library_rand_is_libc_rand_per_thread():
    """ps_viterbi_mutate's deviates: a per-thread glibc random_r() state == rand() of a fresh process (no GPU needed)."""
    import ctypes, threading
    from poreseq_amd import _capi
    api = _capi.load_hip()
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 1234, 0, 2**32 - 1):
        libc.srand(seed)
        want = np.array([libc.rand() for _ in range(20000)]) / (2.0 ** 31)
        api.srand(seed)
        assert np.array_equal(api.rand_draw(20000), want)
    libc.srand(1)
    fresh = np.array([libc.rand() for _ in range(1000)]) / (2.0 ** 31)
    got = {}
    def work(k):
        got[k] = api.rand_draw(1000)      # never seeded in this thread: the state of an unseeded process
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    [t.start() for t in th]; [t.join() for t in th]
    assert all(np.array_equal(got[k], fresh) for k in range(3))


def test_product_has_no_path_to_the_oracle():
    pkg = os.path.join(B.ROOT, "poreseq_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "libps_oracle" not in txt and "oracle/" not in txt.replace("oracle/ ", ""), fn


def test_util_types_and_params(tmp_path):
    m = MutationInfo("12\tA\t.")
    assert (m.start, m.orig, m.mut) == (12, "A", "") and str(m) == "12\tA\t."
    assert MutationInfo("# comment").start == -1 and MutationInfo("1 2").start == -1
    s = MutationScore(); s.start, s.orig, s.mut, s.score = 3, "", "GT", 1.5
    assert str(s) == "3\t.\tGT\t1.5"
    r = RegionInfo("chr1:100:200")
    assert (r.name, r.start, r.end) == ("chr1", 100, 200)
    assert RegionInfo("10:20").name is None and RegionInfo("abc").start is None
    p = tmp_path / "p.conf"
    SaveParams(str(p), {"a": 1.5, "b": 2.0})
    assert LoadParams(str(p)) == {"a": 1.5, "b": 2.0}
    assert LoadParams(None) == {}


def test_event_mapaligns_and_setparams():
    draft, events, truth = synth.make_region(100, 2, 3, B.oracle_swalign, dict(DEFAULT_PARAMS), draft_error=0.0)
    ev = events[0]
    before = ev.ref_align.copy()
    pairs = np.array([(i, i + 2) for i in range(1, 200)])
    ev.mapaligns(pairs)
    assert np.array_equal(ev.ref_align[before > 0], before[before > 0] + 2)
    assert events[0].model.prob_skip == DEFAULT_PARAMS["skip_t"] and events[1].model.prob_skip == DEFAULT_PARAMS["skip_c"]


@pytest.mark.skipif(not B.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", list(range(11, 35)))
def test_oracle_matches_live_reference_random_sweep(seed):
    """Differential sweep of the restatement against the reference's own C++ (oracle/_ref) on small random cases the fixed
    vectors do not hold: narrow and odd band widths, events whose alignment has holes and jumps, unaligned events, random
    multi-base edits at both ends of the sequence, a non-zero lik_offset — every API result and the realigned events."""
    rng = np.random.default_rng(seed)
    W = float(rng.choice([3, 7, 20, 45, 300]))
    P = dict(DEFAULT_PARAMS, verbose=0, realign_width=W, scoring_width=float(rng.choice([2, 9, 30, 100])),
             point_width=float(rng.choice([2, 5, 20])), lik_offset=float(rng.choice([0.0, 0.5, 4.5])))
    L, E = int(rng.integers(60, 320)), int(rng.integers(2, 7))
    draft, events, truth = synth.make_region(L, E, 900 + seed, B.ref_swalign, P)
    ev = copy.deepcopy(events)
    for k, e in enumerate(ev):
        n = e.ref_align.size
        if n > 60 and rng.random() < 0.6:
            a, b = sorted(rng.integers(5, n - 5, 2))
            e.ref_align[a:b] = 0                                               # a hole
        if n > 80 and rng.random() < 0.5:
            c = int(rng.integers(10, n - 50))
            e.ref_align[c:c + 30] = np.minimum(e.ref_align[c:c + 30] + 40, len(draft) - 5) * (e.ref_align[c:c + 30] > 0)   # a jump
        if k == E - 1 and rng.random() < 0.3:
            e.ref_align[:] = 0                                                 # an event that never aligned
    muts = synth.random_point_mutations(rng, draft, 25)
    for _ in range(12):
        st = int(rng.integers(0, len(draft) + 2))
        no = int(rng.integers(0, 6)); nm = int(rng.integers(0, 9))
        mi = MutationInfo(); mi.start, mi.orig, mi.mut = st, draft[st:st + no], "".join(rng.choice(list("ACGT"), nm))
        if mi.orig or mi.mut:
            muts.append(mi)
    logs = []
    for cls in (B.RefPSAlign, B.OraclePSAlign):
        B.reset_rand()
        pa = B.make_pa(cls, draft, copy.deepcopy(ev), P)
        log = [pa.ScoreEvents(), [s.score for s in pa.ScorePoints()], [s.score for s in pa.ScoreMutations(muts)]]
        log.append(pa.Mutate(reps=2)); log.append(pa.sequence)
        log.append(pa.Refine()); log.append(pa.sequence)
        log.append([e.ref_align.tolist() for e in pa.events]); log.append([e.ref_like.tolist() for e in pa.events])
        # ViterbiMutate on the events as generated: the reference's own code reads out of bounds (and crashes) when handed
        # alignments with holes or jumps, so that leg keeps to inputs it defines
        B.reset_rand()
        pv = B.make_pa(cls, draft, copy.deepcopy(events), P)
        log.append(pv.Mutate(seqs="viterbi", reps=1)); log.append(pv.sequence)
        logs.append(log)
    assert logs[0] == logs[1]


@pytest.mark.parametrize("seed", [41, 42, 43, 44])
def test_oracle_is_defined_on_alignments_with_holes_and_jumps(seed):
    """The GPU tests hold the HIP path to the oracle on events whose ref_align has holes, jumps or nothing at all — inputs on
    which the reference's own ViterbiMutate reads out of bounds.  The restatement must be well defined there: same answer twice,
    and clean under tools/oracle_asan.sh (this test is part of that run)."""
    rng = np.random.default_rng(seed)
    P = dict(DEFAULT_PARAMS, verbose=0, realign_width=float(rng.choice([3, 8, 40])), scoring_width=float(rng.choice([4, 20])), point_width=3.0)
    draft, events, truth = synth.make_region(int(rng.integers(120, 300)), 4, 700 + seed, B.oracle_swalign, P)
    ev = copy.deepcopy(events)
    for k, e in enumerate(ev):
        n = e.ref_align.size
        a, b = sorted(rng.integers(5, n - 5, 2))
        e.ref_align[a:b] = 0
        c = int(rng.integers(10, n - 50))
        e.ref_align[c:c + 30] = np.minimum(e.ref_align[c:c + 30] + 40, len(draft) - 5) * (e.ref_align[c:c + 30] > 0)
    ev[-1].ref_align[:] = 0
    runs = []
    for _ in range(2):
        B.reset_rand()
        pa = B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(ev), P)
        log = [pa.ScoreEvents(), [s.score for s in pa.ScorePoints()]]
        log.append(pa.Mutate(seqs="viterbi", reps=1)); log.append(pa.sequence)
        log.append(pa.Mutate(reps=1)); log.append(pa.Refine()); log.append(pa.sequence)
        log.append([e.ref_align.tolist() for e in pa.events])
        runs.append(log)
    assert runs[0] == runs[1]
    assert runs[0][0][-1] == 0.0            # the unaligned event scores nothing
