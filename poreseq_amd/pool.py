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
    __slots__ = ("i", "pa", "h", "rng", "prog", "want", "arg", "result")


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
                r.i, r.pa, r.want, r.arg, r.result = i, pa, None, None, None
                r.h = self.api.align_create(pa.sequence, pa.events, pa.params)
                r.rng = self.api.rng_create(1)              # rand() of a fresh process per region (Viterbi.cpp:108)
                self.regions.append(r)
        return self

    def run(self, params=None, reps=4, refseqs=None, logs=None, workers=4, batch_size=16, serialize_native=False):
        return _run_pool(self, params, reps, refseqs, logs, workers, batch_size, serialize_native)


def consensus_pool(pas, params=None, reps=4, refseqs=None, logs=None, workers=4, batch_size=16, api=None, serialize_native=False):
    """The consensus schedule of `consensus_region` for the independent regions `pas`, scheduled operation by operation.
    Returns [(sequence, accuracy)] in the order of `pas`.  `logs`, when given, is a list of lists receiving
    (call, nbases, sequence) per region after every PSAlign-level call, as `consensus_regions` records them.
    `serialize_native`: let one worker at a time into the native library (the CPU oracle switches glibc's process-wide rand()
    state per region and is not re-entrant; the HIP library is: every host thread has its own runtime and every region its own
    generator)."""
    return RegionPool(pas, api).load().run(params, reps, refseqs, logs, workers, batch_size, serialize_native)


def _run_pool(pool, params, reps, refseqs, logs, workers, batch_size, serialize_native):
    pas, api, native = pool.pas, pool.api, pool.native
    pool.load()
    regions, pool.regions = pool.regions, None
    n = len(pas)
    refseqs = [pa.sequence for pa in pas] if refseqs is None else list(refseqs)
    out = [None] * n
    live = set(r.i for r in regions)
    for i in range(n):
        if i not in live:
            out[i] = (refseqs[i], 100)                      # fewer than 5 events: handed back as loaded

    def note(r, call, nb):
        if logs is not None:
            logs[r.i].append((call, nb, r.pa.sequence))

    def new_call(r, point_width=False):
        p = r.pa.params
        w = p['point_width'] if (point_width and 'point_width' in p) else p.get('scoring_width', 150)
        api.check(api.lib.ps_align_new_call(r.h, int(w)))

    def rounds(r, propose, nrounds):
        """nrounds x {propose -> ScoreMutations -> MakeMutations}; ends when a round changes nothing (pyx:417-431)"""
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

    def mutate(r, kind, nrounds):
        new_call(r)
        if kind == 'self':
            cand = [x.sequence for x in r.pa.events[::2]]
        else:
            cand = yield ("viterbi", None)
        hseq = api.seqs_create(cand)

        def propose():
            hm = yield ("find", hseq)
            return hm
        try:
            tot = yield from rounds(r, propose, nrounds)
        finally:
            api.seqs_destroy(hseq)
        return tot

    def refine(r):
        new_call(r, point_width=True)

        def propose():
            return api.find_point_mutations(r.h)
            yield   # (a generator: FindPointMutations is host-only and needs no batching)
        return (yield from rounds(r, propose, 1))

    def program(r):
        note(r, "Mutate:self", (yield from mutate(r, 'self', reps)))
        for _ in range(reps):
            note(r, "Mutate:viterbi", (yield from mutate(r, 'viterbi', 4)))
            nb = yield from refine(r)
            note(r, "Refine", nb)
            if nb == 0:
                break

    lock = threading.Condition()
    native_lock = threading.Lock()
    state = {"running": 0, "error": None}

    def advance(r):
        """run the region's program up to its next native operation (called with the lock NOT held: host-side work only)"""
        try:
            r.want, r.arg = r.prog.send(r.result)
        except StopIteration:
            r.want, r.arg = None, None
        r.result = None

    for r in regions:
        r.prog = program(r)
        r.result = None
        advance(r)

    def issue(kind, group):
        hs = [r.h for r in group]
        if kind == "viterbi":
            return api.batch_viterbi_mutate(hs, [r.rng for r in group], 16, 0.05, 0.01, 0.33, 0.75)
        if kind == "find":
            return api.batch_find_mutations(hs, [r.arg for r in group])
        if kind == "score":
            return api.batch_score_mutations(hs, [r.arg for r in group])
        return api.batch_make_mutations(hs, [r.arg for r in group])

    waiting = [r for r in regions if r.want is not None]

    def worker():
        while True:
            with lock:
                while True:
                    if state["error"] is not None:
                        return
                    by_kind = {k: [r for r in waiting if r.want == k] for k in _KINDS}
                    kind = max(_KINDS, key=lambda k: len(by_kind[k]))
                    if by_kind[kind]:
                        group = by_kind[kind][:batch_size]
                        for r in group:
                            waiting.remove(r)
                        state["running"] += 1
                        break
                    if state["running"] == 0:
                        lock.notify_all()
                        return                      # nothing waiting, nothing in flight: all programs have ended
                    lock.wait()
            try:
                if serialize_native:
                    with native_lock:
                        res = issue(kind, group)
                else:
                    res = issue(kind, group)
                for r, x in zip(group, res):
                    r.result = x
                    advance(r)
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
    try:
        if state["error"] is not None:
            raise state["error"]
        for r in regions:
            pa = r.pa
            pa.sequence = api.align_sequence(r.h)
            api.align_update_events(r.h, pa.events)
            p = pa.params if params is None else params
            if 'end_trim' in p and len(pa.sequence) > 2 * p['end_trim']:
                pa.sequence = pa.sequence[int(p['end_trim']):-int(p['end_trim'])]
            out[r.i] = (pa.sequence, poreseqcpp.swalign(pa.sequence, refseqs[r.i], native)[0])
    finally:
        for r in regions:
            api.align_destroy(r.h)
            api.rng_destroy(r.rng)
    return out
