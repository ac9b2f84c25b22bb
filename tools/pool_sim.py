#!/usr/bin/env python3
"""Scheduling policies for many regions on one GPU, compared WITHOUT a GPU (CPU only, minutes):
    python tools/pool_sim.py [regions] [workers] [batch_size] [bases]
Runs the real schedule of every region on the CPU oracle (so rounds, convergence and list sizes are real, at a small region size)
and plays the native calls on a virtual clock: a call of kind k over n regions takes  lat[k] + per[k] * n  (launch chains are
latency-bound: measured on MI355X at 10 kb, section 5b of DESIGN.md), up to `workers` calls run side by side and each is
stretched by  1 + share * (calls in flight - 1)  (what a launch loses to the others: k_fill launches take 23 ms in the bench,
13 ms alone, with ~5 kernels in flight).  Compared: fixed lock-step batches (the measured default), the greedy pool and the
fill-or-wait pool of poreseq_amd/pool.py.  A model, not a measurement: it ranks policies by how full they keep the batches and
how many chains they keep in flight; the numbers to trust come from bench.py --scheduler pool on the GPU."""
import copy, heapq, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import backends as B
from poreseq_amd import synth
from poreseq_amd.pool import RegionPool, _Engine, _KINDS, policy_greedy, policy_fill
from poreseq_amd.util import DEFAULT_PARAMS

R = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W = int(sys.argv[2]) if len(sys.argv) > 2 else 6
BS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L = int(sys.argv[4]) if len(sys.argv) > 4 else 250
P = dict(DEFAULT_PARAMS, verbose=0)
# seconds per call at 10 kb / 10 events on MI355X, one batch alone: fixed part (the chain of launches) + per region
LAT = {"viterbi": 0.050, "find": 0.060, "score": 0.030, "make": 0.015}
PER = {"viterbi": 0.0010, "find": 0.0040, "score": 0.0012, "make": 0.0010}
SHARE = 0.15

regs = [synth.make_region(L if k % 3 else (2 * L) // 3, 8 if k % 4 else 6, 9300 + k, B.oracle_swalign, P) for k in range(R)]
mk = lambda: [B.make_pa(B.OraclePSAlign, d, copy.deepcopy(ev), P) for d, ev, _ in regs]


def simulate(policy, cohorts=None):
    """virtual-time run; cohorts: list of region-index sets that may only be batched among themselves (lock-step batches)"""
    pas = mk()
    eng = _Engine(RegionPool(pas), P, 4, None, None)
    cohort_of = {}
    if cohorts:
        for c, idx in enumerate(cohorts):
            for i in idx:
                cohort_of[i] = c
    t, running, events = 0.0, 0, []          # events: (finish time, seq, kind, group)
    waiting = [r for r in eng.regions if r.want is not None]
    calls, served, busy = 0, 0, 0.0
    seq = 0
    busy_cohorts = set()
    while waiting or events:
        progressed = True
        while progressed and running < W:
            progressed = False
            if cohorts:
                # lock-step: a cohort issues the call its regions wait for once ALL of its unfinished members are there
                for c in range(len(cohorts)):
                    if c in busy_cohorts:
                        continue
                    mine = [r for r in waiting if cohort_of[r.i] == c]
                    if not mine:
                        continue
                    alive = [r for r in eng.regions if cohort_of[r.i] == c and r.want is not None]
                    if len(mine) < len(alive):
                        continue                      # members still inside a call
                    # RegionBatch: one PSAlign-level call at a time for the whole batch — members that have finished it wait
                    # for the others' rounds; within the call every member still in it is at the same point
                    first = min(r.phase for r in mine)
                    group = [r for r in mine if r.phase == first]
                    kinds = set(r.want for r in group)
                    assert len(kinds) == 1, kinds
                    pick = (kinds.pop(), group)
                    busy_cohorts.add(c)
                    break
                else:
                    pick = None
            else:
                by_kind = {k: [r for r in waiting if r.want == k] for k in _KINDS}
                pick = policy(by_kind, running, W, BS)
                if pick is None and running == 0 and waiting:
                    pick = policy_greedy(by_kind, 0, W, BS)
            if pick is None:
                break
            kind, group = pick
            for r in group:
                waiting.remove(r)
            running += 1
            dur = (LAT[kind] + PER[kind] * len(group)) * (1.0 + SHARE * (running - 1))
            seq += 1
            heapq.heappush(events, (t + dur, seq, kind, group))
            calls += 1; served += len(group); busy += dur
            progressed = True
        if not events:
            break
        t, _, kind, group = heapq.heappop(events)
        running -= 1
        if cohorts:
            busy_cohorts.discard(cohort_of[group[0].i])
        eng.issue(kind, group)                # the real (oracle) call: results decide what each region asks for next
        waiting.extend(r for r in group if r.want is not None)
    eng.finish()
    return t, calls, served / max(calls, 1), busy / max(t, 1e-9)


print("%d regions of %d / %d bases on the oracle; %d workers, batches of %d; call = lat + per x regions, stretched %.0f %% per call in flight" % (R, L, 2 * L // 3, W, BS, 100 * SHARE))
lock_cohorts = [set(range(k, R, W)) for k in range(W)]
for name, pol, coh in (("lock-step batches", None, lock_cohorts), ("pool, greedy", policy_greedy, None), ("pool, fill or wait", policy_fill, None)):
    B.reset_rand()
    t, calls, fill, conc = simulate(pol, coh)
    print("%-20s virtual time %7.2f s   %4d calls, %5.2f regions per call, %4.2f calls in flight on average" % (name, t, calls, fill, conc))
