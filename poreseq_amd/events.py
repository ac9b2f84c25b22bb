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
        alinds = np.asarray(alignment)
        # where each k-mer of the 2D alignment sits in the 2D sequence: a forward walk with str.find (EventData.py:131-138)
        seqinds = 0 * alinds
        curind = 0
        for i in range(len(alinds)):
            k = kmers[i]
            curind = sequence.find(k.decode() if isinstance(k, bytes) else str(k), curind)
            seqinds[i] = curind
        mean = np.asarray(events["mean"], dtype=np.float64)
        start = np.asarray(events["start"], dtype=np.float64)
        ev = cls(mean - attrs["drift"] * (start - start[0]), events["stdv"], sequence=sequence)
        ev.length = np.array(events["length"], dtype=np.float64)
        ev.start = np.array(start, dtype=np.float64)
        # seed ref_align with the self-alignment: every aligned k-mer's level points at the k-mer's position (EventData.py:159-163)
        lvl = alinds > 0
        ev.ref_align[alinds[lvl]] = seqinds[lvl]
        m = PSModel()
        m.level_mean = np.asarray(model["level_mean"], dtype=np.float64) * attrs["scale"] + attrs["shift"]
        m.level_stdv = np.asarray(model["level_stdv"], dtype=np.float64) * attrs["var"]
        m.sd_mean = np.asarray(model["sd_mean"], dtype=np.float64) * attrs["scale_sd"]
        m.sd_stdv = np.asarray(model["sd_stdv"], dtype=np.float64) / np.sqrt(attrs["var_sd"])
        name = attrs.get("model_file", "") if hasattr(attrs, "get") else attrs["model_file"]
        m.name = name.decode() if isinstance(name, bytes) else str(name)
        m.complement = bool(complement)
        ev.model = m
        if m.complement:
            ev.flip(False)
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
        """Reverse the event in place and map every model state to its reverse complement (EventData.py:182-224): state x's
        complement is 1023 - x (two bits per base, A/T and C/G are bitwise complements), its reverse swaps the five 2-bit digits.
        With flip_sequence the sequence is reverse-complemented too and every aligned ref_align index i becomes len - i."""
        for k in ("mean", "stdv", "length", "start", "ref_align", "ref_like"):
            if hasattr(self, k):
                setattr(self, k, getattr(self, k)[::-1])
        flips = 1023 - np.arange(1024)
        flips = (((flips & 0b11) << 8) | ((flips >> 8) & 0b11) | ((flips & 0b1100) << 4) | ((flips >> 4) & 0b1100) | (flips & 0b110000))
        for k in ("level_mean", "level_stdv", "sd_mean", "sd_stdv"):
            setattr(self.model, k, np.asarray(getattr(self.model, k))[flips])
        if flip_sequence:
            self.sequence = reverse_complement(self.sequence)
            self.ref_align = np.array(self.ref_align, dtype=np.float64)
            ra0 = self.ref_align > 0
            self.ref_align[ra0] = len(self.sequence) - self.ref_align[ra0]
        self.makecontiguous()
        self.flipped = not self.flipped

    def makecontiguous(self):
        for obj in (self, self.model):
            for k, v in vars(obj).items():
                if isinstance(v, np.ndarray):
                    setattr(obj, k, np.ascontiguousarray(v, dtype=np.float64))

    def mapaligns(self, pairs):
        """Re-map ref_align through aligned index pairs (EventData.py:226-256): unique in x,
        linear interpolation, round, 0 outside the paired range."""
        pairs = np.asarray(pairs)
        refal = self.ref_align
        keep = refal > 0
        self.ref_align = 0 * self.ref_align
        _, uinds = np.unique(pairs[:, 0], return_index=True)
        pairs = pairs[uinds, :]
        self.ref_align[keep] = np.round(np.interp(refal[keep], pairs[:, 0], pairs[:, 1], 0, 0))
        self.makecontiguous()

    def setparams(self, params):
        """'skip_t' -> model.prob_skip for template models, '_c' for complement (EventData.py:288-312)."""
        for k in params:
            name = 'prob_' + k[:-2]
            if not hasattr(self.model, name):
                continue
            if (k[-2:] == '_t' and not self.model.complement) or (k[-2:] == '_c' and self.model.complement):
                setattr(self.model, name, params[k])
