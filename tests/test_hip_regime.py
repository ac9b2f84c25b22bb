"""The bench's operating regime pinned to the REFERENCE at BASELINE config #2's size (VERDICT r4 "next" 2).

bench.py's 200 kb/s come from several host threads inside the library at once, each driving a lock-step batch of 10 kb regions
with the default kernel choice — which, with more than one thread in the library, is: strip sweeps with kept columns for every
`ScoreMutations` launch size (two wavefronts per sweep), Smith-Waterman's packed 8-column fill, full matrices in the process-wide
slabs for Refine.  Each piece is parity-tested on its own; here the combination replays the reference's own runs: four slots
(poreseq_amd.dist.stream_batches), each a lock-step RegionBatch of three 10 kb / 10-event regions of which one is the
`consensus_L10000_E10` fixture (the reference's 1 090 s schedule: per-call nbases, the sequence after every call, final ref_align /
ref_like digests, final ScoreEvents) and, in one batch, another is `mutate_seeds_L10000` (one Mutate(list of seeds) call).
Reference: poreseq/Mutate.py:62-93 through pyx:378-472."""
import copy

import numpy as np
import pytest

import backends as B
import golden_util as G
from poreseq_amd import _capi, synth
from poreseq_amd import dist as psdist
from poreseq_amd.batch import RegionBatch
from poreseq_amd.poreseqcpp import PSAlign, swalign
from test_golden_large import _final_refs


@pytest.mark.gpu
def test_four_slots_of_lock_step_10kb_batches_replay_the_reference_fixtures():
    api = _capi.load_hip()
    zc, zs = G.load("consensus_L10000_E10"), G.load("mutate_seeds_L10000")
    fc, fs = G.regen(zc, swalign), G.regen(zs, swalign)              # (draft, events, truth, params), inputs verified by checksum
    par = fc[3]
    others = [synth.make_region(10000, 10, 7100 + k, swalign, par) for k in range(2)]
    seeds = [str(s) for s in zs["seeds"]]

    def region(src):
        return B.make_pa(PSAlign, src[0], copy.deepcopy(src[1]), dict(par))

    def work(slot):
        api.prof_enable(2); api.prof_reset()                          # this thread's runtime: which kernels ran
        srcs = [fc, fs, others[0]] if slot == 0 else [fc, others[slot % 2], others[(slot + 1) % 2]]
        pas = [region(s) for s in srcs]
        with RegionBatch(pas) as rb:
            if slot == 0:                                             # one Mutate(list of seed strings, reps=2) call on the seed-list fixture
                nb = rb.Mutate(idx=[1], seqs=seeds, reps=2)
                assert nb[1] == int(zs["nbases"]) and pas[1].sequence == str(zs["sequence"])
                rb.sync([1])
                _final_refs(pas[1], zs, "mutate_seeds_L10000 (slot 0)")
                assert np.array_equal(np.array(rb.ScoreEvents(idx=[1])[0]), zs["final_ScoreEvents"])
            for call, nb, seq in zip(zc["calls"], zc["nbases"], zc["sequences"]):
                call = str(call)
                got = rb.Mutate(reps=4) if call == "Mutate:self" else rb.Mutate(seqs="viterbi") if call == "Mutate:viterbi" else rb.Refine()
                assert got[0] == int(nb), (slot, call, got)
                assert pas[0].sequence == str(seq), (slot, call)
            rb.sync()
            _final_refs(pas[0], zc, "consensus_L10000_E10 (slot %d)" % slot)
            assert np.array_equal(np.array(rb.ScoreEvents(idx=[0])[0]), zc["final_ScoreEvents"])
        ran = {k: api.prof_get(k)[1] for k in ("sweep", "sweep_w2", "sweep_w4", "fill", "sw", "sw_pk8", "slab")}
        api.prof_enable(0)
        return ran

    for k in ("set_sweep_min", "set_sweep2_min", "set_sparse_min"):    # the library's own thresholds
        getattr(api, k)(-1)
    api.set_sweep_form(0, 0)
    ran = psdist.stream_batches(range(4), work, in_flight=4, enter=psdist._enter_hip_library)
    for slot, r in enumerate(ran):
        # every ScoreMutations of the schedule but Refine's took the kept-column strip sweeps, two wavefronts each, whatever the launch
        # size (with a lone thread they would start at 160 sweeps and a 3-region call's 60 would take k_fill); Refine's full matrices
        # took a slab; every Smith-Waterman batch ran the packed 8-column fill
        assert r["sweep"] > 0 and r["sweep_w2"] == r["sweep"] and r["sweep_w4"] == 0, (slot, r)
        assert r["slab"] >= 4 and r["fill"] >= r["slab"], (slot, r)
        assert r["sw"] > 0 and r["sw_pk8"] == r["sw"], (slot, r)
    assert "slabs for full score matrices: 0 of" not in api.info()


@pytest.mark.gpu
@pytest.mark.parametrize("nw", [1, 2, 4])
def test_first_call_of_a_fresh_process_is_a_kept_column_sweep(nw):
    """A process whose FIRST device work is a column-sparse ScoreMutations with edits at both ends of the sequence (kept column 0 and
    the last one): the record and code pools are then exactly as large as this call needs, so a store outside them is a memory fault
    instead of a silent write into a pool another test had grown (the kept columns' padding once wrapped an unsigned offset that way).
    Scores and alignments equal the oracle's."""
    import os, subprocess, sys
    code = (
        "import sys, copy; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, backends as B\n"
        "from poreseq_amd import synth, _capi\n"
        "from poreseq_amd.poreseqcpp import PSAlign\n"
        "from poreseq_amd.util import DEFAULT_PARAMS, MutationInfo\n"
        "P = dict(DEFAULT_PARAMS, verbose=0)\n"
        "draft, events, truth = synth.make_region(1500, 5, 9100, B.oracle_swalign, P)\n"
        "def e(st, o, m):\n"
        "    x = MutationInfo(); x.start, x.orig, x.mut = st, o, m; return x\n"
        "n = len(draft)\n"
        "muts = [e(0, '', 'T'), e(0, draft[0], ''), e(1, draft[1], 'G'), e(2, '', 'AC'), e(n - 1, draft[n - 1], 'A'), e(n - 2, '', 'G'), e(n, '', 'A'),\n"
        "        e(n // 2, draft[n // 2], ''), e(n // 3, '', 'T')]\n"
        "api = _capi.load_hip(); api.set_sparse_min(0); api.set_sweep_form(0, %d)\n"
        "api.prof_enable(1)\n"
        "res = []\n"
        "for cls in (PSAlign, B.OraclePSAlign):\n"
        "    pa = B.make_pa(cls, draft, copy.deepcopy(events), P)\n"
        "    got = pa.ScoreMutations(muts)\n"
        "    res.append(([g.score for g in got], [ev.ref_align.copy() for ev in pa.events], [ev.ref_like.copy() for ev in pa.events]))\n"
        "assert api.prof_get('sweep')[1] == 1 and api.prof_get('fill')[1] == 0\n"
        "assert res[0][0] == res[1][0]\n"
        "assert all(np.array_equal(a, b) for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]))\n"
        "print('FRESH-OK')\n"
    ) % (B.ROOT, os.path.join(B.ROOT, "tests"), nw)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FRESH-OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.gpu
def test_windows_that_leave_one_lane_idle_match_the_oracle():
    """A strip sweep's window (strips in band on one step) may take all lanes of the sweep but one (sweep_win_max: the lane above the lowest
    strip in band must hold an out-of-band strip).  Band widths scanned around the point where the window of K = 4 rows per lane reaches 63 of
    64 lanes (one wavefront), 127 of 128 (two) and, at K = 2, 255 of 256 (four): forward and backward DP matrices with step codes (ps_debug_fill) and ScoreEvents equal
    the oracle at every width, and the trace shows that the maximal windows were among them."""
    import os, re, subprocess, sys
    code = (
        "import sys, copy; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, backends as B\n"
        "from poreseq_amd import synth, _capi\n"
        "from poreseq_amd.poreseqcpp import PSAlign\n"
        "from poreseq_amd.util import DEFAULT_PARAMS\n"
        "hip, orc = _capi.load_hip(), B.oracle_api()\n"
        "hip.set_sweep_min(0); hip.set_sweep2_min(0); hip.set_sparse_min(0)\n"
        "for nw, K, widths, L in ((1, 4, range(144, 160), 900), (2, 4, range(296, 312, 2), 1500), (4, 2, range(356, 380, 2), 1800)):\n"
        "    hip.set_sweep_form(K, nw)\n"
        "    for W in widths:\n"
        "        P = dict(DEFAULT_PARAMS, verbose=0, realign_width=float(W))\n"
        "        draft, events, truth = synth.make_region(L, 2, 6000 + W, B.oracle_swalign, P)\n"
        "        mk = lambda cls: B.make_pa(cls, draft, copy.deepcopy(events), P)\n"
        "        assert mk(PSAlign).ScoreEvents() == mk(B.OraclePSAlign).ScoreEvents(), (nw, W)\n"
        "        for d in (0, 1):\n"
        "            out = []\n"
        "            for api in (hip, orc):\n"
        "                h = api.align_create(draft, copy.deepcopy(events), P)\n"
        "                out.append(api.debug_fill(h, 1, d, events[1].mean.size, len(draft) - 4))\n"
        "                api.align_destroy(h)\n"
        "            for k in range(4 if d == 0 else 2):\n"
        "                assert np.array_equal(out[0][k], out[1][k], equal_nan=True), (nw, W, d, k)\n"
        "print('WINDOWS-OK')\n"
    ) % (B.ROOT, os.path.join(B.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500, env=dict(os.environ, PORESEQ_TRACE="1"))
    assert r.returncode == 0 and "WINDOWS-OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    wins = {(int(k), int(n), int(w)) for k, n, w in re.findall(r"K = (\d+) on (\d) wavefronts, widest window (\d+) strips", r.stderr)}
    assert (4, 1, 63) in wins and (4, 2, 127) in wins and (2, 4, 255) in wins, sorted(wins)   # the windows that use every lane but one did occur
    assert any(k > 4 for k, n, w in wins)                                      # and wider ones took the next strip height


@pytest.mark.gpu
@pytest.mark.parametrize("nw,K", [(1, 4), (2, 4), (4, 2), (2, 5), (4, 3), (1, 6)])
def test_sweep_matrices_where_the_band_shifts_every_column(nw, K):
    """Narrow bands (realign_width 33 and 61 on 260 / 400 bases: the band moves down on three columns of four) through the strip sweeps with
    full records: forward and backward main / stay matrices and the step codes decoded from the sweeps' predicate bits equal the oracle's cell
    for cell.  The cell below the previous column's last band row (i = p1 + 1) is the one the reference treats as an implicit start although
    its diagonal neighbour holds a score (cpp/Alignment.cpp:207: p0 < i <= p1) — every shifting column has one, in both directions; strip
    heights with planes of 4, 4 + 1, 2, 2 + 1 and 4 + 2 rows cover every layout of the code fields.  The launch counters prove that sweeps ran."""
    import copy
    import numpy as np
    import backends as B
    from poreseq_amd import synth, _capi
    from poreseq_amd.util import DEFAULT_PARAMS
    hip, orc = _capi.load_hip(), B.oracle_api()
    hip.set_sweep_min(0); hip.set_sweep2_min(0); hip.set_sparse_min(0); hip.set_sweep_form(K, nw)
    try:
        for L, W, seed in ((260, 33, 13), (400, 61, 14)):
            P = dict(DEFAULT_PARAMS, verbose=0, realign_width=float(W))
            draft, events, truth = synth.make_region(L, 3, seed, B.oracle_swalign, P)
            hip.prof_reset(); hip.prof_enable(1)
            for d in (0, 1):
                out = []
                for api in (hip, orc):
                    h = api.align_create(draft, copy.deepcopy(events), P)
                    out.append(api.debug_fill(h, 2, d, events[2].mean.size, len(draft) - 4))
                    api.align_destroy(h)
                for k in range(4 if d == 0 else 2):
                    assert np.array_equal(out[0][k], out[1][k], equal_nan=True), (nw, K, W, d, k)
            hip.prof_enable(0)
            assert hip.prof_get("sweep")[1] == 2 and hip.prof_get("fill")[1] == 0
    finally:
        hip.set_sweep_min(-1); hip.set_sweep2_min(-1); hip.set_sparse_min(-1); hip.set_sweep_form(0, 0)


@pytest.mark.gpu
def test_align_slabs_are_recycled_without_leaking_state():
    """AlignData slabs come from a process-wide cache (ps_host.cpp, align_slab_take: no hipMalloc / hipFree per region): a handle that takes
    over the slab of a destroyed one — same size, smaller, from another thread's stream — computes what a fresh process computes.  Regions of
    alternating sizes are created, scored and destroyed in turn, on two host threads at once; every result equals the oracle's."""
    import copy
    import threading
    import numpy as np
    import backends as B
    from poreseq_amd import synth
    from poreseq_amd.poreseqcpp import PSAlign
    from poreseq_amd.util import DEFAULT_PARAMS
    P = dict(DEFAULT_PARAMS, verbose=0)
    cases = [(300, 5, 31), (280, 5, 32), (520, 4, 33), (300, 5, 34), (240, 6, 35), (510, 4, 36)]
    regions = [synth.make_region(L, E, seed, B.oracle_swalign, P) for L, E, seed in cases]
    want = []
    for d, ev, _ in regions:
        pa = B.make_pa(B.OraclePSAlign, d, copy.deepcopy(ev), P)
        sc = pa.ScoreEvents()
        n = pa.Refine()
        want.append((sc, n, pa.sequence))
    errs = []

    def worker(order):
        try:
            for _ in range(3):
                for k in order:
                    d, ev, _ = regions[k]
                    pa = B.make_pa(PSAlign, d, copy.deepcopy(ev), P)
                    sc = pa.ScoreEvents()          # (every PSAlign call creates and destroys its AlignData: pyx:139-153)
                    n = pa.Refine()
                    assert (sc, n, pa.sequence) == want[k], k
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(o,)) for o in ([0, 1, 2, 3, 4, 5], [5, 3, 1, 4, 2, 0])]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


@pytest.mark.gpu
def test_verbose_diagnostics_read_like_the_reference():
    """`verbose` (cpp/MakeMutations.cpp:28-32, 55-66, 91-95, 112-118; cpp/FindMutations.cpp:34-35, 100-109, 230-231; cpp/Viterbi.cpp:258-259):
    a single PSAlign's Refine / Mutate calls write the reference's progress lines to stderr.  Refine's text — "Point ", "Scoring (w)" with a
    dot per event, "Testing N mutations...", "Kept mutation i at s of a to b with score x" — equals the reference C++'s character for
    character where that library is built (oracle/_ref)."""
    import subprocess, sys
    import backends as B
    code = (
        "import sys, copy; sys.path[:0] = [%r, %r]\n"
        "import backends as B\n"
        "from poreseq_amd import synth\n"
        "from poreseq_amd.poreseqcpp import PSAlign\n"
        "from poreseq_amd.util import DEFAULT_PARAMS\n"
        "P = dict(DEFAULT_PARAMS, verbose=2)\n"
        "draft, events, truth = synth.make_region(300, 5, 41, B.oracle_swalign, dict(P, verbose=0))\n"
        "cls = {'hip': PSAlign, 'ref': B.RefPSAlign}[sys.argv[1]]\n"
        "pa = B.make_pa(cls, draft, copy.deepcopy(events), P)\n"
        "B.reset_rand()\n"
        "sys.stderr.write('== Refine\\n'); sys.stderr.flush(); n = pa.Refine()\n"
        "sys.stderr.write('\\n== Mutate\\n'); sys.stderr.flush(); pa.Mutate(seqs='viterbi')\n"
        "print(n, pa.sequence)\n"
    ) % (B.ROOT, __import__("os").path.join(B.ROOT, "tests"))
    def run(which):
        r = subprocess.run([sys.executable, "-c", code, which], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout, r.stderr
    out, err = run("hip")
    refine, mutate = err.split("== Mutate")[0], err.split("== Mutate")[1]
    assert "Point Scoring (20)....." in refine and "Testing " in refine and " mutations..." in refine and "Kept mutation 0 at " in refine
    assert "Viterbi" in mutate and "Finding mutations" in mutate and "Scoring (" in mutate
    if B.have_ref():
        rout, rerr = run("ref")
        assert rout == out
        assert rerr.split("== Mutate")[0] == refine          # Refine's diagnostics: character for character
