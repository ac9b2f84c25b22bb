"""Lock-step refinement of several independent regions on one GPU.

The reference refines one region per process (cmdline.py:182-195); its regions are independent work-items.  One
region's calls keep a few percent of an MI355X busy, so the MI355X way to run many regions is not many processes or
threads but ONE host thread that issues every phase for all regions at once: `RegionBatch` mirrors the `PSAlign`
methods the consensus schedule uses (`Mutate`, `Refine`, `ScoreEvents`) over a list of `PSAlign` objects and drives
the `ps_batch_*` entry points of include/poreseq_hip.h, where each phase (Smith-Waterman batch, banded fills, edit
scoring, Viterbi) is one launch chain over all regions' events.

Results are those of running the same `PSAlign` calls region by region, bit for bit: every region owns a generator
(`ps_rng`, seeded like a fresh process) for ViterbiMutate's stochastic back-traces, and regions that have converged
simply drop out of the later rounds of a call, as their own `break` would.
"""
from . import poreseqcpp


class RegionBatch:
    """A set of PSAlign objects (independent regions) refined in lock-step.  All methods work in place on the members."""

    def __init__(self, pas, api=None, resident=True):
        """resident=True keeps one native AlignData per region for the life of the batch: the events are marshalled and
        copied to the GPU once, every later call only announces itself (`ps_align_new_call` resets the scoring width and
        the seed-likelihood cache, the two things a fresh AlignData would differ in), and sequence / ref_align / ref_like
        are written back to the Python objects when the batch is closed (or by `sync()`).  resident=False rebuilds the
        AlignData for every call, as PythonToAlignData does (pyx:139-153); results are identical."""
        self.pas = list(pas)
        self.api = api if api is not None else (self.pas[0]._native() if self.pas else poreseqcpp._api())
        self.rngs = [self.api.rng_create(1) for _ in self.pas]   # rand() of a fresh process per region (Viterbi.cpp:108)
        self.resident = bool(resident)
        self._h = {}

    def load(self, idx=None):
        """Create the resident AlignData of the regions `idx` now (e.g. before a timed section)."""
        for i in (range(len(self.pas)) if idx is None else idx):
            if self.resident and i not in self._h:
                pa = self.pas[i]
                self._h[i] = self.api.align_create(pa.sequence, pa.events, pa.params)
        return self

    def sync(self, idx=None):
        """Write sequence / ref_align / ref_like of resident regions back to their PSAlign objects."""
        for i in (list(self._h) if idx is None else idx):
            h = self._h.get(i)
            if h is not None:
                pa = self.pas[i]
                pa.sequence = self.api.align_sequence(h)
                self.api.align_update_events(h, pa.events)

    def close(self):
        self.sync()
        for h in self._h.values():
            self.api.align_destroy(h)
        self._h = {}
        for r in self.rngs:
            self.api.rng_destroy(r)
        self.rngs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -- plumbing ---------------------------------------------------------------------------------------------
    def _open(self, idx, point_width=False):
        """Native AlignData for the regions `idx`, as PythonToAlignData builds one per PSAlign call (pyx:139-153)."""
        hs = []
        for i in idx:
            pa = self.pas[i]
            if self.resident:
                self.load([i])
                h = self._h[i]
                w = pa.params['point_width'] if (point_width and 'point_width' in pa.params) else pa.params.get('scoring_width', 150)
                self.api.check(self.api.lib.ps_align_new_call(h, int(w)))
            else:
                h = self.api.align_create(pa.sequence, pa.events, pa.params)
                if point_width and 'point_width' in pa.params:
                    self.api.check(self.api.lib.ps_align_set_scoring_width(h, int(pa.params['point_width'])))
            hs.append(h)
        return hs

    def _close(self, idx, hs, write_back=True):
        for i, h in zip(idx, hs):
            if self.resident:
                if write_back:
                    self.pas[i].sequence = self.api.align_sequence(h)   # cheap; refs follow at sync() / close()
                continue
            if write_back:
                pa = self.pas[i]
                pa.sequence = self.api.align_sequence(h)
                self.api.align_update_events(h, pa.events)
            self.api.align_destroy(h)

    def _rounds(self, idx, hs, propose, reps):
        """reps x {propose -> ScoreMutations -> MakeMutations}; a region leaves when a round changes nothing (pyx:417-431)."""
        tot = {i: 0 for i in idx}
        live = list(range(len(idx)))
        for _ in range(reps):
            if not live:
                break
            lh = [hs[k] for k in live]
            hm = propose(live, lh)
            try:
                scored = self.api.batch_score_mutations(lh, hm)
            finally:
                for m in hm:
                    self.api.muts_destroy(m)
            try:
                nb = self.api.batch_make_mutations(lh, scored)
            finally:
                for m in scored:
                    self.api.muts_destroy(m)
            nxt = []
            for k, n in zip(live, nb):
                if n == 0:
                    continue
                tot[idx[k]] += n
                nxt.append(k)
            live = nxt
        return tot

    # -- the PSAlign calls of the consensus schedule, for the regions `idx` (default: all) ---------------------------
    def ScoreEvents(self, idx=None):
        idx = list(range(len(self.pas))) if idx is None else list(idx)
        hs = self._open(idx)
        try:
            sc = self.api.batch_score_alignments(hs, [len(self.pas[i].events) for i in idx])
        finally:
            self._close(idx, hs, write_back=False)
        return [s.tolist() for s in sc]

    def Mutate(self, idx=None, seqs='self', reps=4):
        """PSAlign.Mutate (pyx:378-435) for the regions `idx`; returns {region index: total mutated bases}."""
        idx = list(range(len(self.pas))) if idx is None else list(idx)
        if not idx:
            return {}
        hs = self._open(idx)
        try:
            if isinstance(seqs, str) and seqs == 'self':
                cand = [[x.sequence for x in self.pas[i].events[::2]] for i in idx]
            elif isinstance(seqs, str) and seqs == 'viterbi':
                cand = self.api.batch_viterbi_mutate(hs, [self.rngs[i] for i in idx], 16, 0.05, 0.01, 0.33, 0.75)
            else:
                cand = [list(seqs) for _ in idx]
            hseq = [self.api.seqs_create(c) for c in cand]
            try:
                tot = self._rounds(idx, hs, lambda live, lh: self.api.batch_find_mutations(lh, [hseq[k] for k in live]), reps)
            finally:
                for s in hseq:
                    self.api.seqs_destroy(s)
        except Exception:
            self._close(idx, hs, write_back=False)
            raise
        self._close(idx, hs)
        return tot

    def Refine(self, idx=None):
        """PSAlign.Refine (pyx:437-472) for the regions `idx`; returns {region index: mutated bases}."""
        idx = list(range(len(self.pas))) if idx is None else list(idx)
        if not idx:
            return {}
        hs = self._open(idx, point_width=True)
        try:
            tot = self._rounds(idx, hs, lambda live, lh: [self.api.find_point_mutations(h) for h in lh], 1)
        except Exception:
            self._close(idx, hs, write_back=False)
            raise
        self._close(idx, hs)
        return tot
