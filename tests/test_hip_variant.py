"""GPU parity at the `poreseq variant` shape (BASELINE config #3, scaled so the oracle finishes in seconds):
ScoreMutations with scoring_width 100 on random point edits, through the variant driver's start offsetting."""
import copy
import io
import os

import numpy as np
import pytest

import backends as B
from poreseq_amd import _capi, synth
from poreseq_amd.consensus import variant_region, consensus_region, split_regions
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS, MutationInfo

pytestmark = pytest.mark.gpu
P0 = dict(DEFAULT_PARAMS, verbose=0)


def test_variant_region_parity_and_format():
    L, E = 1500, 12
    draft, events, truth = synth.make_region(L, E, 71, B.oracle_swalign, P0)
    rng = np.random.default_rng(71)
    region_start = 9000
    res = []
    for cls in (PSAlign, B.OraclePSAlign):
        muts = synth.random_point_mutations(np.random.default_rng(71), draft, 300)
        for m in muts:
            m.start += region_start            # absolute coordinates, as read from a mutation file
        pa = B.make_pa(cls, draft, copy.deepcopy(events), P0)
        out = io.StringIO()
        ms = variant_region(pa, muts, region_start=region_start, out=out)
        res.append((ms, out.getvalue()))
    a, b = res
    assert np.array_equal(np.array([m.score for m in a[0]]), np.array([m.score for m in b[0]]))
    assert a[1] == b[1]
    line = a[1].splitlines()[0].split("\t")
    assert len(line) == 4 and int(line[0]) >= region_start          # start\torig\tmut\tscore, absolute start


def test_all_points_when_mutation_list_is_empty():
    draft, events, truth = synth.make_region(300, 5, 72, B.oracle_swalign, P0)
    a = variant_region(B.make_pa(PSAlign, draft, copy.deepcopy(events), P0), [], region_start=100)
    b = variant_region(B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(events), P0), [], region_start=100)
    assert [m.start for m in a] == [m.start for m in b] and a[0].start == 100
    assert np.array_equal(np.array([m.score for m in a]), np.array([m.score for m in b]))


def test_consensus_driver_matches_oracle_with_end_trim():
    P = dict(P0, end_trim=20.0)
    draft, events, truth = synth.make_region(500, 8, 73, B.oracle_swalign, P)
    out = []
    for cls in (PSAlign, B.OraclePSAlign):
        B.reset_rand()
        pa = B.make_pa(cls, draft, copy.deepcopy(events), P)
        log = []
        seq, acc = consensus_region(pa, P, refseq=truth, log=log)   # accuracy against the synthetic truth
        out.append((seq, acc, [(c, n) for c, n, _ in log]))
    assert out[0] == out[1]
    assert out[0][1] > 97.0 and len(out[0][0]) < len(draft)          # trimmed, accurate
    # fewer than 5 events: the driver returns the input untouched (Mutate.py:50-53)
    pa = B.make_pa(PSAlign, draft, copy.deepcopy(events[:4]), P)
    assert consensus_region(pa, P) == (draft, 100)


def test_unsupported_and_limits_fail_loudly():
    draft, events, truth = synth.make_region(200, 3, 74, B.oracle_swalign, P0, draft_error=0.0)
    api = _capi.load_hip()
    big = [copy.deepcopy(events[k % 3]) for k in range(257)]     # ViterbiMutate sorts at most 256 events per position
    h = api.align_create(draft, big, P0)
    with pytest.raises(_capi.PoreseqError, match="256 events"):
        api.viterbi_mutate(h, 16, 0.05, 0.01, 0.33, 0.75, 0)
    api.align_destroy(h)
    # realign_width 0 makes every alignment a no-op, exactly like the reference's stripe_width == 0
    z = dict(P0, realign_width=0.0)
    assert B.make_pa(PSAlign, draft, copy.deepcopy(events), z).ScoreEvents() == \
        B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(events), z).ScoreEvents() == [0.0, 0.0, 0.0]
    # scoring_width 0 disables the local re-fill (fillColumn returns at once): still identical
    z = dict(P0, scoring_width=0.0, point_width=0.0)
    a = B.make_pa(PSAlign, draft, copy.deepcopy(events), z).ScorePoints()
    b = B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(events), z).ScorePoints()
    assert np.array_equal(np.array([m.score for m in a]), np.array([m.score for m in b]))


def test_concurrent_regions_from_threads_match_oracle():
    """Per-thread runtimes: two regions scored from two host threads at once give the same bits as alone."""
    import threading
    regs = [synth.make_region(400, 6, 81 + k, B.oracle_swalign, P0) for k in range(3)]
    want = [np.array([m.score for m in B.make_pa(B.OraclePSAlign, d, copy.deepcopy(e), P0).ScorePoints()]) for d, e, _ in regs]
    got = [None] * 3

    def work(k):
        for _ in range(3):
            d, e, _t = regs[k]
            got[k] = np.array([m.score for m in B.make_pa(PSAlign, d, copy.deepcopy(e), P0).ScorePoints()])

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(3):
        assert np.array_equal(got[k], want[k])


def test_dense_calls_queue_for_one_small_slab_and_match_alone():
    """Full score matrices live in process-wide slabs that a dense ScoreMutations call takes for its duration (ps_host.cpp).  With ONE
    slab of 0.3 GB (a subprocess: the slab plan is read once), four threads of lock-step Refine calls wait for each other, a batch whose
    matrices exceed the slab is cut in halves, a single region larger than the slab takes the runtime's own pools — and every result equals
    the single-threaded default run's."""
    import hashlib, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import copy, hashlib, sys, threading\n"
        "sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "from poreseq_amd import synth, _capi\n"
        "from poreseq_amd.batch import RegionBatch\n"
        "from poreseq_amd.poreseqcpp import PSAlign, swalign\n"
        "from poreseq_amd.util import DEFAULT_PARAMS\n"
        "P = dict(DEFAULT_PARAMS, verbose=0)\n"
        "regs = [synth.make_region(700 + 150 * k, 6, 6100 + k, swalign, P) for k in range(8)]\n"
        "regs.append(synth.make_region(3000, 8, 6200, swalign, P))          # one region whose matrices alone exceed a 0.3 GB slab\n"
        "def mk(k):\n"
        "    pa = PSAlign(); pa.sequence, pa.events, pa.params = regs[k][0], copy.deepcopy(regs[k][1]), dict(P); return pa\n"
        "out = {}\n"
        "def work(t, ks):\n"
        "    for rep in range(2):\n"
        "        pas = [mk(k) for k in ks]\n"
        "        with RegionBatch(pas) as rb:\n"
        "            nb = rb.Refine()\n"
        "        out[t] = [(nb[i], pa.sequence) for i, pa in enumerate(pas)]\n"
        "nth = int(sys.argv[1])\n"
        "groups = [[0, 1, 2], [3, 4], [5, 6, 7], [8]]\n"
        "if nth == 1:\n"
        "    for t, ks in enumerate(groups): work(t, ks)\n"
        "else:\n"
        "    th = [threading.Thread(target=work, args=(t, ks)) for t, ks in enumerate(groups)]\n"
        "    [x.start() for x in th]; [x.join() for x in th]\n"
        "print('DIGEST', hashlib.sha1(repr(sorted(out.items())).encode()).hexdigest())\n"
        "print('INFO', _capi.load_hip().info())\n"
    ) % (os.path.dirname(here), here)

    def run(nth, extra):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code, str(nth)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.splitlines()
        return [l for l in lines if l.startswith("DIGEST")][0], [l for l in lines if l.startswith("INFO")][0]

    want, _ = run(1, {})
    got, info = run(4, {"PORESEQ_SLABS": "1", "PORESEQ_SLAB_GB": "0.3"})
    assert got == want
    assert "1 of 1 allocated (0.3 GB" in info


def test_run_regions_in_flight_is_deterministic_and_matches_fresh_process_oracle():
    """The region driver with several regions in flight per GPU: every region is the full stochastic consensus
    schedule (ViterbiMutate draws random numbers), results equal the one-at-a-time run and the oracle run of a
    fresh process (srand(1) before each region) — whichever worker thread picks a region up."""
    from poreseq_amd import dist as psdist
    specs = [(350 + 20 * k, 5, 900 + k) for k in range(5)]
    made = {s: synth.make_region(s[0], s[1], s[2], B.oracle_swalign, P0) for s in specs}

    def process(cls):
        def f(spec):
            d, e, _t = made[spec]
            pa = B.make_pa(cls, d, copy.deepcopy(e), P0)
            seq, _ = consensus_region(pa, P0)
            return seq, np.array(pa.ScoreEvents())
        return f

    one = psdist.run_regions(specs, process(PSAlign), max_events=8, in_flight=1)
    many = psdist.run_regions(specs, process(PSAlign), max_events=8, in_flight=3)
    assert [r[0] for r in one] == [r[0] for r in many]
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(one, many))
    ref = psdist.run_regions(specs[:2], process(B.OraclePSAlign), max_events=8, in_flight=1, fresh_rand=B.reset_rand)
    assert [r[0] for r in ref] == [r[0] for r in one[:2]]
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(ref, one[:2]))


@pytest.mark.parametrize("form", ["chain", "one"])
def test_chained_smith_waterman_strips_under_load_match_oracle(form, monkeypatch):
    """10 kb pairs take three chained 4096-column strips per pair (ps_sw.hip: ticket-ordered workgroups, boundary column handed over
    through global memory) or, in the form lock-step batches use, one 16-wave workgroup per pair.  Eight host threads run such pairs
    at once, next to a thread that keeps the chip busy with fills; every alignment must equal the oracle's, bit for bit."""
    import threading
    monkeypatch.setenv("PORESEQ_SW_FORM", form)
    rng = np.random.default_rng(77)
    pairs = []
    for k in range(8):
        a = synth.random_sequence(rng, 9000 + 400 * k)
        pairs.append((a, synth.corrupt(rng, a, 0.04, 0.04, 0.04)))
    want = [B.oracle_swalign(a, b) for a, b in pairs[:3]]
    ref3 = [swalign(a, b) for a, b in pairs[:3]]
    assert ref3 == want                                              # alone: equals the oracle
    alone = [swalign(a, b) for a, b in pairs]
    got = [None] * 8
    stop = threading.Event()
    draft, events, truth = synth.make_region(4000, 10, 83, swalign, P0)

    def load():
        while not stop.is_set():
            B.make_pa(PSAlign, draft, copy.deepcopy(events), P0).ScoreEvents()

    def work(k):
        for _ in range(3):
            got[k] = swalign(*pairs[k])

    bg = threading.Thread(target=load)
    bg.start()
    th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    stop.set()
    bg.join()
    assert got == alone


@pytest.mark.gpu
def test_second_streams_on_priority_levels_finish_and_match():
    """lock-step batches from several host threads, every runtime with a SECOND stream for its Smith-Waterman
    batches on the next stream priority level (PORESEQ_FORCE_STREAM2=prio: the configuration of round 2's run that did not
    finish), chained Smith-Waterman strips spinning on each other beside fills — under a hard time limit, results equal to the
    default single-stream run.  (The library's default stays one stream per runtime: the second one buys nothing, DESIGN.md.)"""
    import subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import hashlib, sys, threading, copy\n"
        "sys.path[:0] = [%r, %r]\n"
        "from poreseq_amd import synth\n"
        "from poreseq_amd.batch import RegionBatch\n"
        "from poreseq_amd.poreseqcpp import PSAlign, swalign\n"
        "from poreseq_amd.util import DEFAULT_PARAMS\n"
        "P = dict(DEFAULT_PARAMS, verbose=0)\n"
        "regs = [synth.make_region(5000, 6, 9100 + k, swalign, P) for k in range(12)]\n"
        "out = [None] * 6\n"
        "def work(t):\n"
        "    pas = []\n"
        "    for d, ev, _ in regs[2 * t:2 * t + 2]:\n"
        "        pa = PSAlign(); pa.sequence, pa.events, pa.params = d, copy.deepcopy(ev), dict(P); pas.append(pa)\n"
        "    with RegionBatch(pas) as rb:\n"
        "        n = rb.Mutate(reps=2)\n"
        "    out[t] = (n, [pa.sequence for pa in pas])\n"
        "th = [threading.Thread(target=work, args=(t,)) for t in range(6)]\n"
        "[t.start() for t in th]; [t.join() for t in th]\n"
        "print('DIGEST', hashlib.sha1(repr(out).encode()).hexdigest())\n"
    ) % (os.path.dirname(here), here)

    def run(extra):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=420)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]

    want = run({})
    assert run({"PORESEQ_FORCE_STREAM2": "prio"}) == want
    assert run({"PORESEQ_FORCE_STREAM2": "1", "PORESEQ_ONE_PRIORITY": "1"}) == want


def test_refine_regions_batch_slots_in_flight_equal_one_batch_at_a_time():
    """poreseq_amd.dist.refine_regions(in_flight=N): lock-step batches streaming through N host-thread slots (bench.py's shape) give
    every region the result of batches refined one after the other — whichever slot takes a batch, whatever else shares the GPU."""
    from poreseq_amd import dist as psdist
    regions = [(1000 * k, 1000 * k + 420 + 30 * (k % 3)) for k in range(9)]
    made = {r: synth.make_region(r[1] - r[0], 6, 700 + r[0] // 1000, B.oracle_swalign, P0) for r in regions}

    def make(a, b):
        d, e, _t = made[(a, b)]
        return B.make_pa(PSAlign, d, copy.deepcopy(e), P0)

    one = psdist.refine_regions(regions, make, params=P0, batch=2)
    many = psdist.refine_regions(regions, make, params=P0, batch=2, in_flight=3)
    assert one == many and len(one) == len(regions)
    # and the oracle, region by region in a fresh process's random stream, says the same for the first two
    for r, (seq, acc) in list(zip(regions, one))[:2]:
        B.reset_rand()
        d, e, _t = made[r]
        assert consensus_region(B.make_pa(B.OraclePSAlign, d, copy.deepcopy(e), P0), P0) == (seq, acc)
