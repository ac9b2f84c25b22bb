"""Event / model types consumed at the boundary and the array half of the real-data front end (8(f4)).

The native layer reads exactly what the reference's `PythonToEvents` reads
(pyx:99-129): float64 1-D `mean, stdv, ref_align, ref_like`, `sequence`, and
`model.{level_mean, level_stdv, sd_mean, sd_stdv, complement, prob_*}`.  The file formats are read through h5py / pysam where those
are installed (`PSEvent.from_fast5`, `poreseq_amd.loaddata.events_from_bam`: poreseq/EventData.py:113-128, LoadData.py:81-90 — thin
readers with guarded imports; neither library is in the build image, the tests run them on stand-in modules) and everything
the reference computes from the parsed tables is here: `PSEvent.from_basecall` (model scaling, drift, the k-mer walk that seeds
ref_align: EventData.py:130-175), `flip` (:182-224), `mapaligns` (:226-256), `setparams` (:288-312); the read selection of
`EventsFromBAM` on parsed records is `poreseq_amd.loaddata`.  Vectors: tests/golden/frontend.npz, made by running the reference's own
class against stand-ins for the h5py datasets (tests/golden/make_golden_frontend.py).
"""
import copy as _copy

import numpy as np

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def reverse_complement(seq):
    """str(Bio.Seq.Seq(seq).reverse_complement()) for the alphabet the path knows (other characters stay as they are)"""
    return "".join(_COMP.get(c, c) for c in reversed(seq))


def _text(x):
    return x.decode() if isinstance(x, bytes) else str(x)


def _kmer_offsets(sequence, kmers):
    """Offset of every k-mer of a 2D alignment table in the 2D sequence: each k-mer is searched from where the one before it was found
    (str.find; a k-mer that does not occur gives -1, and the search for the next one then starts at -1 — Python's own meaning of a
    negative start, one character from the end — as the reference's walk does, EventData.py:131-138)."""
    out = np.zeros(len(kmers), dtype=np.int64)
    at = 0
    for n, k in enumerate(kmers):
        at = sequence.find(_text(k), at)
        out[n] = at
    return out


def _reverse_complement_states():
    """perm[x] = the 5-mer state of x's reverse complement.  A state is five 2-bit digits, first base most significant, A C G T = 0 1 2 3:
    complementing a base is 3 - digit, reversing the 5-mer reverses the digits."""
    x = np.arange(1024)
    digits = [(x >> (2 * k)) & 3 for k in range(5)]           # digits[0]: the LAST base
    perm = np.zeros(1024, dtype=np.int64)
    for k, d in enumerate(digits):
        perm |= (3 - d) << (2 * (4 - k))                       # the last base's complement becomes the first base
    return perm


_RC_STATES = _reverse_complement_states()
_PER_LEVEL = ("mean", "stdv", "length", "start", "ref_align", "ref_like")     # arrays with one entry per level
_PER_STATE = ("level_mean", "level_stdv", "sd_mean", "sd_stdv")               # model tables with one entry per 5-mer


class PSModel:
    """poreseq/EventData.py:46-78 (same attribute names and fall-back probabilities)."""

    def __init__(self):
        self.level_mean = np.zeros(1024)
        self.level_stdv = np.ones(1024)
        self.sd_mean = np.ones(1024)
        self.sd_stdv = np.ones(1024)
        self.prob_skip = 0.1
        self.prob_stay = 0.1
        self.prob_extend = self.prob_stay
        self.prob_insert = 0.01
        self.name = ''
        self.complement = False


class PSEvent:
    """Array-backed stand-in for poreseq/EventData.py:80-312 (no fast5 access)."""

    def __init__(self, mean, stdv, ref_align=None, ref_like=None, sequence="", model=None):
        self.mean = np.array(mean, dtype=np.float64)
        self.stdv = np.array(stdv, dtype=np.float64)
        n = self.mean.size
        self.ref_align = np.zeros(n) if ref_align is None else np.array(ref_align, dtype=np.float64)
        self.ref_like = np.zeros(n) if ref_like is None else np.array(ref_like, dtype=np.float64)
        self.sequence = sequence
        self.model = model if model is not None else PSModel()
        self.flipped = False

    @classmethod
    def from_basecall(cls, events, model, attrs, sequence, alignment, kmers, complement=False):
        """The event a fast5 file's 2D basecall describes, from its parsed tables (EventData.py:100-175 without the h5py reads).

        events     the strand's `Events` table: fields (or dict keys) mean, stdv, length, start
        model      its `Model` table: level_mean, level_stdv, sd_mean, sd_stdv (1024 entries, 5-mer order AAAAA .. TTTTT)
        attrs      the strand's `basecall_1d_*` summary attributes: shift, scale, scale_sd, drift, var, var_sd, model_file
        sequence   the 2D basecall (second line of the Fastq dataset)
        alignment  level index of this strand for every row of the 2D `Alignment` table (<= 0: k-mer not aligned to a level)
        kmers      the table's k-mer column
        A complement strand is flipped (levels reversed, model states mapped to their reverse complements) and keeps its sequence,
        so that template and complement point the same way (EventData.py:173-175)."""
        sequence = str(sequence)
        level_of_row = np.asarray(alignment)                     # per row of the 2D alignment table: this strand's level (<= 0: none)
        offset_of_row = _kmer_offsets(sequence, kmers).astype(level_of_row.dtype)
        field = lambda table, name: np.asarray(table[name], dtype=np.float64)
        start = field(events, "start")
        ev = cls(field(events, "mean") - attrs["drift"] * (start - start[0]), events["stdv"], sequence=sequence)   # drift off the levels
        ev.length = field(events, "length").copy()
        ev.start = start.copy()
        # the self-alignment seeds ref_align: the level of every aligned row points at its k-mer's offset in the 2D sequence
        # (EventData.py:159-163; rows that share a level: the last one wins, as in a fancy-indexed assignment)
        aligned = level_of_row > 0
        ev.ref_align[level_of_row[aligned]] = offset_of_row[aligned]
        # the strand's pore model, scaled to this read (EventData.py:141-157)
        m = PSModel()
        m.level_mean = field(model, "level_mean") * attrs["scale"] + attrs["shift"]
        m.level_stdv = field(model, "level_stdv") * attrs["var"]
        m.sd_mean = field(model, "sd_mean") * attrs["scale_sd"]
        m.sd_stdv = field(model, "sd_stdv") / np.sqrt(attrs["var_sd"])
        m.name = _text(attrs.get("model_file", "") if hasattr(attrs, "get") else attrs["model_file"])
        m.complement = bool(complement)
        ev.model = m
        if m.complement:
            ev.flip(False)                                       # complement strands are stored the other way round; the sequence stays
        return ev

    @classmethod
    def from_fast5(cls, filename, typ):
        """The reference's `PSEvent(filename, typ)` (EventData.py:100-175): the template ('t') or complement ('c') strand of an
        R7 2D-basecalled fast5 file.  Only the five dataset reads are here (EventData.py:113-128: `Events`, `Model`, the
        `basecall_1d_*` summary attributes, the 2D `Fastq` and `Alignment`); everything computed from them is `from_basecall`.
        Needs h5py, which this package does not depend on: without it the call fails with an ImportError that says so."""
        try:
            import h5py
        except ImportError as e:
            raise ImportError("PSEvent.from_fast5 reads fast5 files through h5py, which is not installed; "
                              "PSEvent.from_basecall takes the parsed tables") from e
        loc = "complement" if str(typ)[0] == "c" else "template"
        base = "/Analyses/Basecall_2D_000/"
        f = h5py.File(filename, "r")
        try:
            fastq = f[base + "BaseCalled_2D/Fastq"][()]
            fastq = fastq.decode() if isinstance(fastq, bytes) else str(fastq)
            aldata = f[base + "BaseCalled_2D/Alignment"]
            return cls.from_basecall(f[base + "BaseCalled_" + loc + "/Events"], f[base + "BaseCalled_" + loc + "/Model"],
                                     f[base + "Summary/basecall_1d_" + loc].attrs, fastq.split("\n")[1],
                                     aldata[loc], aldata["kmer"], complement=(loc == "complement"))
        finally:
            close = getattr(f, "close", None)
            if close is not None:
                close()

    def copy(self):
        return _copy.deepcopy(self)

    def flip(self, flip_sequence=True):
        """Turn the event round in place (EventData.py:182-224): the per-level arrays run backwards and every model table is read
        through the reverse-complement permutation of the 5-mer states (`_RC_STATES`).  With flip_sequence the sequence is
        reverse-complemented too and an aligned level's ref_align i becomes len(sequence) - i (0 stays "not aligned")."""
        for name in _PER_LEVEL:
            if hasattr(self, name):
                setattr(self, name, getattr(self, name)[::-1])
        for name in _PER_STATE:
            setattr(self.model, name, np.asarray(getattr(self.model, name))[_RC_STATES])
        if flip_sequence:
            self.sequence = reverse_complement(self.sequence)
            ra = np.array(self.ref_align, dtype=np.float64)
            self.ref_align = np.where(ra > 0, len(self.sequence) - ra, ra)
        self.makecontiguous()
        self.flipped = not self.flipped

    def makecontiguous(self):
        for obj in (self, self.model):
            for k, v in vars(obj).items():
                if isinstance(v, np.ndarray):
                    setattr(obj, k, np.ascontiguousarray(v, dtype=np.float64))

    def mapaligns(self, pairs):
        """Carry ref_align over to another sequence through aligned (index here, index there) pairs (EventData.py:226-256): of
        pairs that share their first index the first one counts; an aligned level's index is interpolated linearly between the
        pairs and rounded (numpy: half to even); levels that are not aligned, or whose index lies outside the paired range, get 0."""
        pairs = np.asarray(pairs)
        src = pairs[:, 0]
        order = np.argsort(src, kind="stable")
        first = np.ones(len(order), dtype=bool)
        first[1:] = src[order][1:] != src[order][:-1]
        xs, ys = src[order][first], pairs[order, 1][first]       # strictly increasing sources, each with its first partner
        old = self.ref_align
        new = np.zeros_like(old)
        on = old > 0
        new[on] = np.round(np.interp(old[on], xs, ys, left=0, right=0))
        self.ref_align = new
        self.makecontiguous()

    def setparams(self, params):
        """'skip_t' -> model.prob_skip for template models, '_c' for complement (EventData.py:288-312)."""
        for k in params:
            name = 'prob_' + k[:-2]
            if not hasattr(self.model, name):
                continue
            if (k[-2:] == '_t' and not self.model.complement) or (k[-2:] == '_c' and self.model.complement):
                setattr(self.model, name, params[k])
