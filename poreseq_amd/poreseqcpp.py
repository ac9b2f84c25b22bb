"""Drop-in for the reference's compiled module `poreseq.poreseqcpp` (poreseq/_poreseqcpp.pyx).

Same names, argument meaning, return types and in-place behaviour: `PSAlign`, `swalign`,
`seqtostates`.  Where the reference's Cython code calls its C++ core, this module calls
the C ABI of include/poreseq_hip.h, served by hand-written HIP kernels
(poreseq_amd/csrc).  There is no CPU path: importing works anywhere, but every compute
call raises if libporeseq_hip.so is missing or no MI355X-class device is usable.
"""
import copy

import numpy as np

from . import _capi
from .util import MutationInfo, MutationScore


def _api():
    return _capi.load_hip()


def swalign(seq1, seq2, _api=_api):
    """Smith-Waterman align two sequences (pyx:155-174).

    Returns (accuracy in %, [(i1, i2), ...]) with 1-based indices and 0 for a gap.
    """
    _score, acc, i1, i2 = _api().swfull(seq1, seq2)
    return (acc, list(zip(i1.tolist(), i2.tolist())))


def seqtostates(seq, _api=_api):
    """5-mer states [0, 1023] of a nucleotide string (pyx:176-187)."""
    return _api().seq_to_states(seq).tolist()


class PSAlign:
    """All data of reads aligned to a reference (pyx:189-472).

    Attributes:
        sequence (str): the sequence the events are currently aligned to
        events (list): event objects (see poreseq_amd.events.PSEvent for the duck type)
        params (dict): 'verbose', 'lik_offset', 'realign_width', 'scoring_width', 'point_width'
    All methods work in place; use .Copy() first for non-destructive behaviour.
    """

    _native = staticmethod(_api)  # tests rebind this to the oracle / reference shim

    def __init__(self):
        self.sequence = ""
        self.events = []
        self.params = {}

    # -- plumbing ---------------------------------------------------------------
    class _Data:
        """Scoped native AlignData built from `self`, as PythonToAlignData does per call (pyx:139-153)."""

        def __init__(self, pa, point_width=False):
            self.api = pa._native()
            self.h = self.api.align_create(pa.sequence, pa.events, pa.params)
            if point_width and 'point_width' in pa.params:
                self.api.check(self.api.lib.ps_align_set_scoring_width(self.h, int(pa.params['point_width'])))

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self.api.align_destroy(self.h)
            return False

    def _scores_to_py(self, api, hm):
        start, orig, mut, score = api.muts_export(hm)
        out = []
        for i in range(len(start)):
            s = MutationScore()
            s.start = int(start[i])
            s.orig = orig[i]
            s.mut = mut[i]
            s.score = float(score[i])
            out.append(s)
        return out

    # -- reference API ----------------------------------------------------------
    def Copy(self):
        return copy.deepcopy(self)

    def Coverage(self):
        """Events aligned over each base of self.sequence (pyx:225-239)."""
        cov = np.zeros(len(self.sequence))
        for ev in self.events:
            nzs = ev.ref_align[ev.ref_align > 0]
            lo = int(nzs[0])
            hi = int(np.minimum(nzs[-1], len(cov) - 1))
            cov[lo:hi] += 1
        return cov

    def RealignTo(self, newseq):
        """Re-map every event onto `newseq` through swalign (pyx:241-261)."""
        align = swalign(self.sequence, newseq, self._native)
        if align[0] < 0.6:  # sic: compares a percentage, as the reference does (pyx:256)
            raise Exception('Error rate too large for realignment!')
        pairs = np.array(align[1])
        for ev in self.events:
            ev.mapaligns(pairs)
        self.sequence = newseq

    def ScoreEvents(self):
        """Total likelihood score of each event (pyx:263-276).  Does not write ref_align back."""
        with PSAlign._Data(self) as d:
            return d.api.score_alignments(d.h, len(self.events)).tolist()

    def ScorePoints(self):
        """Score every single-base deletion / substitution / insertion (pyx:278-308)."""
        with PSAlign._Data(self, point_width=True) as d:
            hm = d.api.find_point_mutations(d.h)
            try:
                hs = d.api.score_mutations(d.h, hm)
            finally:
                d.api.muts_destroy(hm)
            try:
                return self._scores_to_py(d.api, hs)
            finally:
                d.api.muts_destroy(hs)

    def ScoreMutations(self, muts):
        """Score the given MutationInfo list, same order (pyx:310-345)."""
        with PSAlign._Data(self) as d:
            hm = d.api.muts_create(muts)
            try:
                hs = d.api.score_mutations(d.h, hm)
            finally:
                d.api.muts_destroy(hm)
            try:
                return self._scores_to_py(d.api, hs)
            finally:
                d.api.muts_destroy(hs)

    def ScoreMutationDeltas(self, muts):
        """ndarray [events][edits]: what each event adds to each edit's score (MakeMutations.cpp:51); `ScoreMutations` returns
        -1e-6 plus their sum in event order.  Not part of the reference's surface: the building block of event-sharded scoring
        (poreseq_amd.dist.score_mutations_event_sharded)."""
        with PSAlign._Data(self) as d:
            hm = d.api.muts_create(muts)
            try:
                return d.api.score_mutation_deltas(d.h, hm, len(self.events), len(muts))
            finally:
                d.api.muts_destroy(hm)

    def ApplyMuts(self, pymuts):
        """Greedy MakeMutations over already-scored mutations (pyx:347-375)."""
        with PSAlign._Data(self, point_width=True) as d:
            hm = d.api.muts_create(pymuts, with_scores=True)
            try:
                d.api.make_mutations(d.h, hm)
            finally:
                d.api.muts_destroy(hm)
            self.sequence = d.api.align_sequence(d.h)
            d.api.align_update_events(d.h, self.events)

    def Mutate(self, seqs='self', reps=4):
        """Seed-sequence driven consensus improvement (pyx:378-435).

        seqs: 'self' (every other event's own sequence), 'viterbi' (16 stochastic Viterbi
        seeds) or a list of strings.  Returns the total number of mutated bases.
        """
        with PSAlign._Data(self) as d:
            sequences = []
            if isinstance(seqs, str) and seqs == 'self':
                seqs = [x.sequence for x in self.events[::2]]
            elif isinstance(seqs, str) and seqs == 'viterbi':
                seqs = None
                sequences = d.api.viterbi_mutate(d.h, 16, 0.05, 0.01, 0.33, 0.75, self.params['verbose'])
            if seqs:
                sequences = list(seqs)
            totbases = 0
            for _ in range(reps):
                hm = d.api.find_mutations(d.h, sequences)
                try:
                    hs = d.api.score_mutations(d.h, hm)
                finally:
                    d.api.muts_destroy(hm)
                try:
                    nbases = d.api.make_mutations(d.h, hs)
                finally:
                    d.api.muts_destroy(hs)
                if nbases == 0:
                    break
                totbases += nbases
            self.sequence = d.api.align_sequence(d.h)
            d.api.align_update_events(d.h, self.events)
            return totbases

    def Refine(self):
        """Brute-force all single-base edits and apply the improving ones (pyx:437-472)."""
        with PSAlign._Data(self, point_width=True) as d:
            hm = d.api.find_point_mutations(d.h)
            try:
                hs = d.api.score_mutations(d.h, hm)
            finally:
                d.api.muts_destroy(hm)
            try:
                nbases = d.api.make_mutations(d.h, hs)
            finally:
                d.api.muts_destroy(hs)
            self.sequence = d.api.align_sequence(d.h)
            d.api.align_update_events(d.h, self.events)
            return nbases
