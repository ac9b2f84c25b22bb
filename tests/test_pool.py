"""Op-level scheduling of regions (poreseq_amd.pool.consensus_pool): every region must come out exactly as if it had been
refined on its own by a fresh process, whatever company it kept in the batched native calls.  CPU: on the oracle's (looping)
ps_batch_* entry points, which checks the scheduler — the programs, the hand-over of results, the bookkeeping; the GPU library
behind the same entry points is covered by tests/test_batch.py."""
import copy

import numpy as np
import pytest

import backends as B
from poreseq_amd import synth
from poreseq_amd.consensus import consensus_region
from poreseq_amd.pool import consensus_pool, policy_fill, policy_greedy
from poreseq_amd.util import DEFAULT_PARAMS

P = dict(DEFAULT_PARAMS, verbose=0)


def _regions(sizes, seed0):
    return [synth.make_region(L, E, seed0 + k, B.oracle_swalign, P) for k, (L, E) in enumerate(sizes)]


def _alone(regs):
    res, logs = [], []
    for d, ev, _ in regs:
        B.reset_rand()                                   # a fresh process per region
        pa = B.make_pa(B.OraclePSAlign, d, copy.deepcopy(ev), P)
        log = []
        res.append((consensus_region(pa, P, log=log), [np.array(e.ref_align) for e in pa.events], [np.array(e.ref_like) for e in pa.events]))
        logs.append(log)
    return res, logs


@pytest.mark.parametrize("workers,batch_size,policy", [(1, 16, policy_greedy), (4, 3, policy_greedy), (3, 4, policy_fill)])
def test_pool_equals_regions_refined_alone(workers, batch_size, policy):
    regs = _regions([(260, 6), (180, 5), (300, 8), (150, 4), (220, 7), (240, 6), (200, 5)], 8100)
    want, wlogs = _alone(regs)
    pas = [B.make_pa(B.OraclePSAlign, d, copy.deepcopy(ev), P) for d, ev, _ in regs]
    logs = [[] for _ in regs]
    got = consensus_pool(pas, P, logs=logs, workers=workers, batch_size=batch_size, policy=policy, serialize_native=True)   # (the oracle is not re-entrant)
    assert logs == wlogs                                 # per call: name, nbases, sequence after the call
    for (w, wra, wrl), g, pa in zip(want, got, pas):
        assert tuple(w) == tuple(g)
        for u, v in zip(wra, [np.array(e.ref_align) for e in pa.events]):
            assert np.array_equal(u, v)
        for u, v in zip(wrl, [np.array(e.ref_like) for e in pa.events]):
            assert np.array_equal(u, v)
