"""Lock-step refinement of several regions (poreseq_amd.batch / consensus.consensus_regions, the ps_batch_* ABI).

The contract: every region comes out exactly as if it had been refined on its own by a fresh process.  CPU: the Python
driver over the oracle's (looping) batch entry points.  GPU: the HIP library's batched launch chains against the oracle
run region by region, and against the library's own single-region path at a size the oracle cannot reach in a test.
"""
import copy
import os

import numpy as np
import pytest

import backends as B
from poreseq_amd import synth
from poreseq_amd.consensus import consensus_region, consensus_regions
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS

P = dict(DEFAULT_PARAMS, verbose=0)


def _regions(sizes, seed0, sw):
    out = []
    for k, (L, E) in enumerate(sizes):
        draft, events, truth = synth.make_region(L, E, seed0 + k, sw, P)
        out.append((draft, events, truth))
    return out


def _one_by_one(cls, regs, params):
    res, logs = [], []
    for draft, events, truth in regs:
        B.reset_rand()                       # each region = a fresh process (Viterbi.cpp:108 never seeds rand())
        pa = B.make_pa(cls, draft, copy.deepcopy(events), params)
        log = []
        seq, acc = consensus_region(pa, params, log=log)
        res.append((seq, acc, [np.array(ev.ref_align) for ev in pa.events], [np.array(ev.ref_like) for ev in pa.events]))
        logs.append(log)
    return res, logs


def _lock_step(cls, regs, params):
    pas = [B.make_pa(cls, d, copy.deepcopy(ev), params) for d, ev, _ in regs]
    logs = [[] for _ in regs]
    out = consensus_regions(pas, params, logs=logs)
    res = [(s, a, [np.array(ev.ref_align) for ev in pa.events], [np.array(ev.ref_like) for ev in pa.events])
           for (s, a), pa in zip(out, pas)]
    return res, logs


def _same(a, b):
    (ra, la), (rb, lb) = a, b
    assert la == lb                                         # per call: name, nbases, sequence after the call
    for x, y in zip(ra, rb):
        assert x[0] == y[0] and x[1] == y[1]
        for u, v in zip(x[2], y[2]):
            assert np.array_equal(u, v)
        for u, v in zip(x[3], y[3]):
            assert np.array_equal(u, v)


def test_lock_step_driver_equals_region_by_region_oracle():
    """ragged batch: different lengths / event counts, one region below the 5-event threshold (returned untouched)"""
    regs = _regions([(260, 5), (180, 6), (220, 3), (300, 5)], 7100, B.oracle_swalign)
    _same(_one_by_one(B.OraclePSAlign, regs, P), _lock_step(B.OraclePSAlign, regs, P))


@pytest.mark.gpu
def test_hip_lock_step_equals_oracle_region_by_region():
    regs = _regions([(420, 6), (300, 5), (350, 7), (260, 5), (500, 6)], 7200, B.oracle_swalign)
    _same(_one_by_one(B.OraclePSAlign, regs, P), _lock_step(PSAlign, regs, P))


@pytest.mark.gpu
@pytest.mark.parametrize("kept_columns", ["default", "always", "never"])
def test_hip_lock_step_equals_hip_single_at_3kb(kept_columns):
    """eight 3 kb regions (bands narrower than the columns, several seed-batch chunks) against the single-region path; with the
    kept-column sweeps (k_sweeps) for every ScoreMutations call whose edit lists allow it — regions of different lengths and edit lists
    share one launch, each with its own table of kept columns — for none, and with the default thresholds"""
    from poreseq_amd import _capi
    api = _capi.load_hip()
    regs = _regions([(3000, 10)] * 3 + [(2500, 8), (1500, 10), (3000, 6), (800, 10), (2000, 9)], 7300, swalign)
    want = _one_by_one(PSAlign, regs, P)
    api.set_sparse_min({"default": -1, "always": 0, "never": 1 << 30}[kept_columns])
    try:
        _same(want, _lock_step(PSAlign, regs, P))
    finally:
        api.set_sparse_min(-1)


@pytest.mark.gpu
def test_hip_batch_score_events_and_memory_chunking(monkeypatch):
    """a tiny matrix budget forces one candidate sequence per launch chain: same results"""
    regs = _regions([(600, 6), (500, 5)], 7400, swalign)
    want = _lock_step(PSAlign, regs, P)
    monkeypatch.setenv("PORESEQ_MAX_BATCH_GB", "0.01")
    _same(want, _lock_step(PSAlign, regs, P))
    from poreseq_amd.batch import RegionBatch
    pas = [B.make_pa(PSAlign, d, copy.deepcopy(ev), P) for d, ev, _ in regs]
    with RegionBatch(pas) as rb:
        got = rb.ScoreEvents()
    assert got == [B.make_pa(PSAlign, d, copy.deepcopy(ev), P).ScoreEvents() for d, ev, _ in regs]


@pytest.mark.gpu
def test_hip_seed_chunks_are_cut_again_when_the_bands_are_wider_than_guessed():
    """FindMutations sizes its candidate batches on a guess of the band footprint; with a guess far too small
    (PORESEQ_DEBUG_GUESS_P) the first chunk must be re-cut, and the results stay the same"""
    import hashlib, json, os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import hashlib, json, sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from test_batch import _regions, _lock_step, P\n"
        "from poreseq_amd.poreseqcpp import PSAlign, swalign\n"
        "regs = _regions([(900, 8), (700, 6)], 7500, swalign)\n"
        "res, logs = _lock_step(PSAlign, regs, P)\n"
        "h = hashlib.sha1(json.dumps(logs, default=str).encode())\n"
        "for s, a, ras, rls in res:\n"
        "    h.update(str(s).encode()); h.update(repr(a).encode())\n"
        "    for u in ras + rls: h.update(u.tobytes())\n"
        "print('DIGEST', h.hexdigest())\n"
    ) % (os.path.dirname(here), here)
    def run(extra):
        env = dict(os.environ, PORESEQ_TRACE="1", **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0], r.stderr
    want, err0 = run({})
    got, err1 = run({"PORESEQ_DEBUG_GUESS_P": "64", "PORESEQ_MAX_BATCH_GB": "0.2", "PORESEQ_NO_SWEEP": "1"})   # k_fill's matrices
    assert "cut again" not in err0 and "over the share" not in err0 and "the rest in chunks" not in err0
    assert "cut again" in err1           # FindMutations' candidate batches
    assert "over the share" in err1      # the lock-step realigns of ScoreAlignments / ScoreMutations
    assert got == want
    # the strip sweep (forward-only batches): a strip height too small for the band is raised, code pools over the share are split
    got3, err3 = run({"PORESEQ_DEBUG_SWEEP_K": "4", "PORESEQ_MAX_BATCH_GB": "0.02", "PORESEQ_SWEEP_MIN": "0"})
    assert "K = 4 on 1 wave" in err3 and ("K = 6 on 1 wave" in err3 or "K = 10 on 1 wave" in err3) and "of step codes" in err3
    assert got3 == want
    assert run({"PORESEQ_SWEEP_MIN": "0"})[0] == want    # both kernels: the same results
    got2, err2 = run({"PORESEQ_MAX_BATCH_GB": "0.002"})
    assert "the rest in chunks" in err2  # Smith-Waterman batches larger than an eighth of the share
    assert got2 == want


def test_package_import_asks_hip_for_more_hardware_queues_unless_told_otherwise():
    """poreseq_amd/__init__.py: GPU_MAX_HW_QUEUES=16 is exported at import when nobody has set it and nothing in the process has
    opened the GPU (lock-step batches in flight want a hardware queue each: DESIGN.md 5b), together with the flag that tells the
    library the value is in force; a value from the user's environment stays and is not flagged."""
    import subprocess, sys
    code = ("import os, sys; sys.path.insert(0, %r); import poreseq_amd; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), os.environ.get('PORESEQ_HWQ_SET_BY_PACKAGE'))" % B.ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "PORESEQ_HWQ_SET_BY_PACKAGE")}
    assert subprocess.check_output([sys.executable, "-c", code], env=env, timeout=300).decode().split()[-2:] == ["16", "1"]
    assert subprocess.check_output([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="4"), timeout=300).decode().split()[-2:] == ["4", "None"]
    # a process that has opened the GPU before the import (here: a stand-in file descriptor on /dev/kfd is not available on a CPU box,
    # so the check itself is exercised): the package leaves the environment alone
    code2 = ("import os, sys; sys.path.insert(0, %r); import poreseq_amd as p; "
             "p._os.readlink = lambda path: '/dev/kfd'; p._os.environ.pop('GPU_MAX_HW_QUEUES', None); p._os.environ.pop('PORESEQ_HWQ_SET_BY_PACKAGE', None); "
             "p._want_hw_queues(); print(os.environ.get('GPU_MAX_HW_QUEUES'), os.environ.get('PORESEQ_HWQ_SET_BY_PACKAGE'))" % B.ROOT)
    assert subprocess.check_output([sys.executable, "-c", code2], env=env, timeout=300).decode().split()[-2:] == ["None", "None"]


def test_library_says_how_its_streams_get_hardware_queues():
    """ps_info (C ABI): a GPU_MAX_HW_QUEUES that appears in the environment after start-up WITHOUT the package's flag is not trusted
    (HIP may have started before it was set: seven streams would share four queues of one priority level); the start-up environment
    and the package's own export are."""
    import subprocess, sys
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "%s\n"
            "from poreseq_amd import _capi\nprint(_capi.load_hip().info())" )
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "PORESEQ_HWQ_SET_BY_PACKAGE", "PORESEQ_ONE_PRIORITY", "PORESEQ_PRIORITY_LEVELS")}
    run = lambda pre, e: subprocess.check_output([sys.executable, "-c", code % (B.ROOT, pre)], env=e, timeout=300).decode()
    assert "exported by the poreseq_amd package" in run("", env)
    assert "start-up environment" in run("", dict(env, GPU_MAX_HW_QUEUES="12"))
    late = run("os.environ['GPU_MAX_HW_QUEUES'] = '12'   # set by a host program after start-up, before the import: the package does not flag it", env)
    assert "not trusted" in late and "dealt over the priority levels" in late
    assert "HIP's default" in run("", dict(env, GPU_MAX_HW_QUEUES="4"))
