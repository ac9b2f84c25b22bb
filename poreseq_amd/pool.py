"""Op-level scheduling of many independent regions on one GPU (experimental; `consensus_regions` / `RegionBatch` is the measured path).

`RegionBatch` refines a fixed set of regions in lock-step: every PSAlign call of the schedule is issued once for the regions of
the batch that still take part in it, so late rounds — few regions still changing — are small launches at full latency, and a
batch moves at the pace of its slowest member.  Here every region is its own little program (a generator that follows the
schedule of `consensus_region` call by call and yields the native operations it needs: ViterbiMutate, FindMutations,
ScoreMutations, MakeMutations), and a handful of worker threads keep picking the kind of operation most regions are waiting for
and issue it for up to `batch_size` of them through the same `ps_batch_*` entry points.  Batches stay full until the very end,
and no region waits for another one's rounds.

Results are those of `consensus_region` run on each region alone, bit for bit: a region's sequence of native calls and their
arguments are exactly the same, only the company it keeps in a launch changes (tests/test_pool.py, CPU, on the oracle).
"""
import threading

from . import poreseqcpp

_KINDS = ("viterbi", "find", "score", "make")   # the native, batchable operations (ties are broken in this order)


class _Region:
    __slots__ = ("i", "pa", "h", "rng", "prog", "want", "arg", "result", "phase")   # phase: PSAlign-level calls completed


class RegionPool:
    """The regions `pas` with their native AlignData resident (load()), refined by run().  One-shot: run() closes the pool."""

    def __init__(self, pas, api=None):
        self.pas = list(pas)
        self.native = self.pas[0]._native if self.pas else poreseqcpp._api   # (tests rebind PSAlign._native to the oracle)
        self.api = api if api is not None else self.native()
        self.regions = None

    def load(self):
        """marshal the events and copy them to the GPU (e.g. before a timed section)"""
        if self.regions is None:
            self.regions = []
            for i, pa in enumerate(self.pas):
                pa.params.setdefault('verbose', 0)
                if len(pa.events) < 5:                      # Mutate.py:50-53
                    continue
                r = _Region()
                r.i, r.pa, r.want, r.arg, r.result, r.phase = i, pa, None, None, None, 0
                r.h = self.api.align_create(pa.sequence, pa.events, pa.params)
                r.rng = self.api.rng_create(1)              # rand() of a fresh process per region (Viterbi.cpp:108)
                self.regions.append(r)
        return self

    def run(self, params=None, reps=4, refseqs=None, logs=None, workers=4, batch_size=16, serialize_native=False, stats=None,
            policy=None):
        """`stats`, when a dict, receives {kind: [native calls, regions served]}; `policy`: policy_greedy (default) / policy_fill"""
        return _run_pool(self, params, reps, refseqs, logs, workers, batch_size, serialize_native, stats, policy or policy_greedy)


def consensus_pool(pas, params=None, reps=4, refseqs=None, logs=None, workers=4, batch_size=16, api=None, serialize_native=False,
                   stats=None, policy=None):
    """The consensus schedule of `consensus_region` for the independent regions `pas`, scheduled operation by operation.
    Returns [(sequence, accuracy)] in the order of `pas`.  `logs`, when given, is a list of lists receiving
    (call, nbases, sequence) per region after every PSAlign-level call, as `consensus_regions` records them.
    `serialize_native`: let one worker at a time into the native library (the CPU oracle switches glibc's process-wide rand()
    state per region and is not re-entrant; the HIP library is: every host thread has its own runtime and every region its own
    generator)."""
    return RegionPool(pas, api).load().run(params, reps, refseqs, logs, workers, batch_size, serialize_native, stats, policy)


def policy_greedy(by_kind, running, workers, batch_size):
    """whatever most regions wait for, at once (keeps every worker busy; batches may be small)"""
    kind = max(_KINDS, key=lambda k: len(by_kind[k]))
    return (kind, by_kind[kind][:batch_size]) if by_kind[kind] else None


def policy_fill(by_kind, running, workers, batch_size):
    """a full batch if there is one; a partial one only while fewer than half of the workers are busy (otherwise wait for the
    operations in flight to bring more regions to the same point)"""
    kind = max(_KINDS, key=lambda k: len(by_kind[k]))
    if not by_kind[kind]:
        return None
    if len(by_kind[kind]) >= batch_size or running < max(1, workers // 2):
        return kind, by_kind[kind][:batch_size]
    return None


class _Engine:
    """The regions' programs and the native operations they ask for; drivers (the threaded one below, the virtual-time one of
    tools/pool_sim.py) decide which regions go into which call."""

    def __init__(self, pool, params, reps, refseqs, logs):
        self.pool, self.params, self.logs = pool, params, logs
        self.api, self.native = pool.api, pool.native
        pas = pool.pas
        pool.load()
        self.regions, pool.regions = pool.regions, None
        self.refseqs = [pa.sequence for pa in pas] if refseqs is None else list(refseqs)
        self.out = [None] * len(pas)
        live = set(r.i for r in self.regions)
        for i in range(len(pas)):
            if i not in live:
                self.out[i] = (self.refseqs[i], 100)        # fewer than 5 events: handed back as loaded
        for r in self.regions:
            r.prog = self._program(r, reps)
            r.result = None
            self.advance(r)

    # -- a region's program: the schedule of consensus_region, call by call ---------------------------------------------------
    def _note(self, r, call, nb):
        r.phase += 1
        if self.logs is not None:
            self.logs[r.i].append((call, nb, r.pa.sequence))

    def _new_call(self, r, point_width=False):
        p = r.pa.params
        w = p['point_width'] if (point_width and 'point_width' in p) else p.get('scoring_width', 150)
        self.api.check(self.api.lib.ps_align_new_call(r.h, int(w)))

    def _rounds(self, r, propose, nrounds):
        """nrounds x {propose -> ScoreMutations -> MakeMutations}; ends when a round changes nothing (pyx:417-431)"""
        api = self.api
        tot = 0
        for _ in range(nrounds):
            hm = yield from propose()
            try:
                scored = yield ("score", hm)
            finally:
                api.muts_destroy(hm)
            try:
                nb = yield ("make", scored)
            finally:
                api.muts_destroy(scored)
            if nb == 0:
                break
            tot += nb
        r.pa.sequence = api.align_sequence(r.h)
        return tot

    def _mutate(self, r, kind, nrounds):
        api = self.api
        self._new_call(r)
        if kind == 'self':
            cand = [x.sequence for x in r.pa.events[::2]]
        else:
            cand = yield ("viterbi", None)
        hseq = api.seqs_create(cand)

        def propose():
            hm = yield ("find", hseq)
            return hm
        try:
            tot = yield from self._rounds(r, propose, nrounds)
        finally:
            api.seqs_destroy(hseq)
        return tot

    def _refine(self, r):
        self._new_call(r, point_width=True)

        def propose():
            return self.api.find_point_mutations(r.h)
            yield   # (a generator: FindPointMutations is host-only and needs no batching)
        return (yield from self._rounds(r, propose, 1))

    def _program(self, r, reps):
        self._note(r, "Mutate:self", (yield from self._mutate(r, 'self', reps)))
        for _ in range(reps):
            self._note(r, "Mutate:viterbi", (yield from self._mutate(r, 'viterbi', 4)))
            nb = yield from self._refine(r)
            self._note(r, "Refine", nb)
            if nb == 0:
                break

    # -- what a driver needs ------------------------------------------------------------------------------------------------
    def advance(self, r):
        """run the region's program up to its next native operation (host-side work only)"""
        try:
            r.want, r.arg = r.prog.send(r.result)
        except StopIteration:
            r.want, r.arg = None, None
        r.result = None

    def issue(self, kind, group):
        """one batched native call for the regions `group`, all waiting for `kind`; then every program moves on"""
        api = self.api
        hs = [r.h for r in group]
        if kind == "viterbi":
            res = api.batch_viterbi_mutate(hs, [r.rng for r in group], 16, 0.05, 0.01, 0.33, 0.75)
        elif kind == "find":
            res = api.batch_find_mutations(hs, [r.arg for r in group])
        elif kind == "score":
            res = api.batch_score_mutations(hs, [r.arg for r in group])
        else:
            res = api.batch_make_mutations(hs, [r.arg for r in group])
        for r, x in zip(group, res):
            r.result = x
            self.advance(r)

    def finish(self, ok=True):
        api = self.api
        try:
            if ok:
                for r in self.regions:
                    pa = r.pa
                    pa.sequence = api.align_sequence(r.h)
                    api.align_update_events(r.h, pa.events)
                    p = pa.params if self.params is None else self.params
                    if 'end_trim' in p and len(pa.sequence) > 2 * p['end_trim']:
                        pa.sequence = pa.sequence[int(p['end_trim']):-int(p['end_trim'])]
                    self.out[r.i] = (pa.sequence, poreseqcpp.swalign(pa.sequence, self.refseqs[r.i], self.native)[0])
        finally:
            for r in self.regions:
                api.align_destroy(r.h)
                api.rng_destroy(r.rng)
        return self.out


def _run_pool(pool, params, reps, refseqs, logs, workers, batch_size, serialize_native, stats=None, policy=policy_greedy):
    eng = _Engine(pool, params, reps, refseqs, logs)
    lock = threading.Condition()
    native_lock = threading.Lock()
    state = {"running": 0, "error": None}
    waiting = [r for r in eng.regions if r.want is not None]

    def worker():
        while True:
            with lock:
                while True:
                    if state["error"] is not None:
                        return
                    by_kind = {k: [r for r in waiting if r.want == k] for k in _KINDS}
                    pick = policy(by_kind, state["running"], workers, batch_size)
                    if pick is None and state["running"] == 0 and waiting:
                        pick = policy_greedy(by_kind, 0, workers, batch_size)   # nothing in flight to wait for
                    if pick is not None:
                        kind, group = pick
                        for r in group:
                            waiting.remove(r)
                        state["running"] += 1
                        if stats is not None:
                            c = stats.setdefault(kind, [0, 0])
                            c[0] += 1; c[1] += len(group)
                        break
                    if state["running"] == 0:
                        lock.notify_all()
                        return                      # nothing waiting, nothing in flight: all programs have ended
                    lock.wait()
            try:
                if serialize_native:
                    with native_lock:
                        eng.issue(kind, group)
                else:
                    eng.issue(kind, group)
            except Exception as e:   # pragma: no cover
                with lock:
                    state["error"] = e
                    state["running"] -= 1
                    lock.notify_all()
                return
            with lock:
                state["running"] -= 1
                waiting.extend(r for r in group if r.want is not None)
                lock.notify_all()

    threads = [threading.Thread(target=worker) for _ in range(max(1, workers) - 1)]
    for t in threads:
        t.start()
    worker()
    for t in threads:
        t.join()
    out = eng.finish(ok=state["error"] is None)
    if state["error"] is not None:
        raise state["error"]
    return out
