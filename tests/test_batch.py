"""Lock-step refinement of several regions (poreseq_amd.batch / consensus.consensus_regions, the ps_batch_* ABI).

The contract: every region comes out exactly as if it had been refined on its own by a fresh process.  CPU: the Python
driver over the oracle's (looping) batch entry points.  GPU: the HIP library's batched launch chains against the oracle
run region by region, and against the library's own single-region path at a size the oracle cannot reach in a test.
"""
import copy

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
def test_hip_lock_step_equals_hip_single_at_3kb():
    """eight 3 kb regions (bands narrower than the columns, several seed-batch chunks) against the single-region path"""
    regs = _regions([(3000, 10)] * 3 + [(2500, 8), (1500, 10), (3000, 6), (800, 10), (2000, 9)], 7300, swalign)
    _same(_one_by_one(PSAlign, regs, P), _lock_step(PSAlign, regs, P))


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
