"""Array half of the real-data front end against vectors produced by the reference's own PSEvent class and
EventsFromBAM function run on stand-ins for the h5py datasets and pysam records (tests/golden/frontend.npz, made by
tests/golden/make_golden_frontend.py): model scaling, drift, the k-mer walk, the complement / reverse-strand flip, the read
selection and the alignment transfer.  Everything here is index and table arithmetic in float64: required bit-exact."""
import os

import numpy as np
import pytest

import golden_util as G
from poreseq_amd import loaddata
from poreseq_amd.events import PSEvent, reverse_complement

Z = np.load(os.path.join(G.GOLDEN, "frontend.npz"), allow_pickle=False)
FIELDS = ("mean", "stdv", "length", "start", "ref_align", "ref_like")
MFIELDS = ("level_mean", "level_stdv", "sd_mean", "sd_stdv")


def load(tag, loc):
    """the strand's event from the stored tables, as a fast5 reader would call it"""
    name = "template" if loc == "t" else "complement"
    attrs = {k: Z["%s_in_%s_attr_%s" % (tag, name, k)][()] for k in ("shift", "scale", "scale_sd", "drift", "var", "var_sd", "model_file")}
    al = Z["%s_in_alignment" % tag]
    return PSEvent.from_basecall(Z["%s_in_%s_events" % (tag, name)], Z["%s_in_%s_model" % (tag, name)], attrs, str(Z["%s_in_sequence" % tag]),
                                 al[name], al["kmer"], complement=(loc == "c"))


def same(ev, tag):
    for k in FIELDS:
        assert np.array_equal(getattr(ev, k), Z["%s_%s" % (tag, k)]), (tag, k)
        assert getattr(ev, k).flags["C_CONTIGUOUS"] and getattr(ev, k).dtype == np.float64
    for k in MFIELDS:
        assert np.array_equal(getattr(ev.model, k), Z["%s_model_%s" % (tag, k)]), (tag, k)
    assert ev.sequence == str(Z["%s_sequence" % tag])
    assert ev.flipped == bool(Z["%s_flipped" % tag]) and ev.model.complement == bool(Z["%s_complement" % tag])
    assert ev.model.name == str(Z["%s_model_name" % tag])


@pytest.mark.parametrize("tag", [str(n).split(".")[0] for n in Z["event_names"]])
@pytest.mark.parametrize("loc", ["t", "c"])
def test_event_from_basecall_tables_and_flip_equal_the_reference_class(tag, loc):
    ev = load(tag, loc)
    same(ev, "%s_%s" % (tag, loc))
    assert ev.flipped == (loc == "c")                  # a complement strand arrives flipped, with its sequence untouched
    ev2 = ev.copy()
    ev2.flip()
    same(ev2, "%s_%s_flip" % (tag, loc))
    ev2.flip()                                         # an involution: levels, model permutation, sequence, aligned indices
    for k in FIELDS:
        assert np.array_equal(getattr(ev2, k), getattr(ev, k))
    for k in MFIELDS:
        assert np.array_equal(getattr(ev2.model, k), getattr(ev.model, k))
    assert ev2.sequence == ev.sequence and ev2.flipped == ev.flipped


def test_flip_permutation_is_the_reverse_complement_of_every_5mer():
    from poreseq_amd.poreseqcpp import seqtostates
    ev = PSEvent(np.zeros(3), np.ones(3), sequence="ACGTA")
    ev.model.level_mean = np.arange(1024, dtype=np.float64)
    ev.flip(False)
    rng = np.random.default_rng(3)
    for _ in range(200):
        kmer = "".join("ACGT"[k] for k in rng.integers(0, 4, 5))
        s = seqtostates(kmer)[0]
        assert ev.model.level_mean[s] == seqtostates(reverse_complement(kmer))[0]


class Rec:
    def __init__(self, k):
        self.query_name = str(Z["bam_rec_name"][k])
        self.is_reverse = bool(Z["bam_rec_is_reverse"][k])
        self.s, self.e = int(Z["bam_rec_ref_start"][k]), int(Z["bam_rec_ref_end"][k])
        clip = int(Z["bam_rec_hard_clip"][k])
        self.p = [(None if a < 0 else int(a), None if b < 0 else int(b)) for a, b in Z["bam_rec%d_pairs" % k]]
        self.cigar = ([(5, clip)] if clip else []) + [(0, len(self.p))]

    def get_overlap(self, start, end):
        return max(0, min(end, self.e) - max(start, self.s))

    def get_aligned_pairs(self):
        return list(self.p)


def test_events_from_bam_records_equal_the_reference_function():
    recs = [Rec(k) for k in range(len(Z["bam_rec_name"]))]
    start, end = (int(x) for x in Z["bam_region"])
    mo, mc, mn = (int(x) for x in Z["bam_params"])
    params = {"min_overlap": mo, "max_coverage": mc, "min_coverage": mn}

    def loader(name, loc):
        tag = name.split(".")[0]
        if "%s_in_sequence" % tag not in Z.files:
            raise IOError("no such fast5 file")        # a read whose file is missing is skipped, both strands
        return load(tag, loc)

    sel = loaddata.select_records(recs, start, end, params)
    assert [r.query_name for r in sel] == ["missing.fast5", "readC.fast5", "readB.fast5"]    # overlaps 221, 195, 192 (then readA 169 / 118, readD 30): most first, three at most
    events = loaddata.events_from_bam_records(recs, loader, start, end, params)
    assert len(events) == int(Z["bam_n_events"])
    for k, ev in enumerate(events):
        same(ev, "bam_ev%d" % k)
    assert bool(Z["bam_insufficient_raises"])
    with pytest.raises(Exception, match="Insufficient coverage!"):
        loaddata.events_from_bam_records(recs, loader, start, end, dict(params, min_coverage=9))
    with pytest.raises(Exception, match="No aligned reads found!"):
        loaddata.events_from_bam_records(recs, lambda n, l: (_ for _ in ()).throw(IOError()), start, end, params)


# ---- the file half (EventData.py:113-128, LoadData.py:81-90) on stand-ins for the h5py / pysam modules -------------------------------
class _Dataset:
    """what the readers use of an h5py Dataset: field / index access, [()] for scalars, .attrs"""
    def __init__(self, value, attrs=None):
        self.value, self.attrs = value, attrs or {}

    def __getitem__(self, k):
        return self.value if isinstance(k, tuple) and k == () else self.value[k]


def _fake_fast5(tag):
    base = "/Analyses/Basecall_2D_000/"
    store = {}
    for name in ("template", "complement"):
        attrs = {k: Z["%s_in_%s_attr_%s" % (tag, name, k)][()] for k in ("shift", "scale", "scale_sd", "drift", "var", "var_sd", "model_file")}
        store[base + "BaseCalled_%s/Events" % name] = _Dataset(Z["%s_in_%s_events" % (tag, name)])
        store[base + "BaseCalled_%s/Model" % name] = _Dataset(Z["%s_in_%s_model" % (tag, name)])
        store[base + "Summary/basecall_1d_%s" % name] = _Dataset(None, attrs)
    seq = str(Z["%s_in_sequence" % tag])
    store[base + "BaseCalled_2D/Fastq"] = _Dataset(("@read\n%s\n+\n%s\n" % (seq, "!" * len(seq))).encode())   # h5py >= 3 hands out bytes
    store[base + "BaseCalled_2D/Alignment"] = _Dataset(Z["%s_in_alignment" % tag])
    return store


def _h5py_standin():
    import types

    def File(filename, mode):
        tag = os.path.basename(filename).split(".")[0]
        if "%s_in_sequence" % tag not in Z.files:
            raise IOError("unable to open file: %s" % filename)
        return _fake_fast5(tag)
    return types.SimpleNamespace(File=File)


@pytest.mark.parametrize("loc", ["t", "c"])
def test_from_fast5_reads_the_reference_s_five_datasets(monkeypatch, loc):
    import sys
    monkeypatch.setitem(sys.modules, "h5py", _h5py_standin())
    for tag in (str(n).split(".")[0] for n in Z["event_names"]):
        same(PSEvent.from_fast5("/data/run7/%s.fast5" % tag, loc), "%s_%s" % (tag, loc))
    with pytest.raises(IOError):
        PSEvent.from_fast5("/data/run7/missing.fast5", loc)


def test_events_from_bam_opens_the_files_like_the_reference(monkeypatch, tmp_path):
    import sys, types
    from poreseq_amd.util import RegionInfo
    recs = [Rec(k) for k in range(len(Z["bam_rec_name"]))]
    start, end = (int(x) for x in Z["bam_region"])
    seen = {}

    class AlignmentFile:
        nreferences, references = 1, ("chrT",)

        def __init__(self, fn, mode):
            seen["open"] = (fn, mode)

        def fetch(self, reference=None, start=None, end=None):
            seen["fetch"] = (reference, start, end)
            return iter(recs)
    monkeypatch.setitem(sys.modules, "pysam", types.SimpleNamespace(AlignmentFile=AlignmentFile))
    monkeypatch.setitem(sys.modules, "h5py", _h5py_standin())
    mo, mc, mn = (int(x) for x in Z["bam_params"])
    params = {"min_overlap": mo, "max_coverage": mc, "min_coverage": mn, "skip_t": 0.07, "stay_c": 0.12}
    reg = RegionInfo("%d:%d" % (start, end))
    events = loaddata.events_from_bam("/data/run7", "aln.bam", reg, params)
    assert seen == {"open": ("aln.bam", "rb"), "fetch": ("chrT", start, end)} and reg.name == "chrT"    # the name is written back (LoadData.py:84-87)
    assert len(events) == int(Z["bam_n_events"])
    for k, ev in enumerate(events):
        same(ev, "bam_ev%d" % k)
    AlignmentFile.nreferences, AlignmentFile.references = 2, ("chrT", "chrU")
    with pytest.raises(Exception, match="Multiple references in BAM"):
        loaddata.events_from_bam("/data/run7", "aln.bam", RegionInfo("%d:%d" % (start, end)), params)
    # LoadAlignedEvents: reference slice, events, setparams on every event (LoadData.py:10-51)
    AlignmentFile.nreferences, AlignmentFile.references = 1, ("chrT",)
    ref = "".join("ACGT"[k] for k in np.random.default_rng(5).integers(0, 4, 700))
    fa = tmp_path / "ref.fa"
    fa.write_text(">chrT some description\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")

    class PA:
        pass
    pa = loaddata.load_aligned_events(str(fa), "aln.bam", "/data/run7", RegionInfo("chrT:%d:%d" % (start, end)), params, psalign=PA)
    assert pa.sequence == ref[start:end] and pa.params is params and len(pa.events) == int(Z["bam_n_events"])
    for ev in pa.events:
        assert (ev.model.prob_stay == 0.12) == ev.model.complement and (ev.model.prob_skip == 0.07) == (not ev.model.complement)
    whole = loaddata.load_aligned_events(str(fa), "aln.bam", "/data/run7", RegionInfo(None), dict(params, min_coverage=0), psalign=PA)
    assert whole.sequence == ref
    (tmp_path / "two.fa").write_text(">a\nACGT\n>b\nGGCC\n")
    assert loaddata.load_reference(str(tmp_path / "two.fa"), "b") == "GGCC"
    with pytest.raises(Exception, match="Multiple references in fasta"):
        loaddata.load_reference(str(tmp_path / "two.fa"))


def test_file_readers_say_which_library_is_missing(monkeypatch):
    import sys
    from poreseq_amd.util import RegionInfo
    monkeypatch.setitem(sys.modules, "h5py", None)      # (None in sys.modules: the import fails, installed or not)
    monkeypatch.setitem(sys.modules, "pysam", None)
    with pytest.raises(ImportError, match="h5py"):
        PSEvent.from_fast5("x.fast5", "t")
    with pytest.raises(ImportError, match="pysam"):
        loaddata.events_from_bam(".", "x.bam", RegionInfo("0:10"), {})


def test_missing_h5py_reaches_the_caller_not_no_reads_found(monkeypatch, capsys):
    """with pysam present and h5py absent the promised ImportError must surface (before: swallowed per strand, the user
    saw 'No aligned reads found!'); other per-strand failures are skipped and — under `verbose` — shown, as LoadData.py:146-147 does."""
    import sys, types
    from poreseq_amd.util import RegionInfo

    class AlignmentFile:
        nreferences, references = 1, ("chrT",)

        def __init__(self, fn, mode):
            pass

        def fetch(self, reference=None, start=None, end=None):
            return iter([])
    monkeypatch.setitem(sys.modules, "pysam", types.SimpleNamespace(AlignmentFile=AlignmentFile))
    monkeypatch.setitem(sys.modules, "h5py", None)
    with pytest.raises(ImportError, match="h5py"):
        loaddata.events_from_bam(".", "x.bam", RegionInfo("0:10"), {})
    recs = [Rec(k) for k in range(len(Z["bam_rec_name"]))]
    start, end = (int(x) for x in Z["bam_region"])
    mo, mc, mn = (int(x) for x in Z["bam_params"])
    params = {"min_overlap": mo, "max_coverage": mc, "min_coverage": mn, "verbose": 1}

    def no_reader(name, loc):
        raise ImportError("no h5py here")
    with pytest.raises(ImportError, match="no h5py here"):
        loaddata.events_from_bam_records(recs, no_reader, start, end, params)

    def broken(name, loc):
        raise KeyError("Basecall_2D_000")
    with pytest.raises(Exception, match="No aligned reads found"):
        loaddata.events_from_bam_records(recs, broken, start, end, params)
    assert "Skipping" in capsys.readouterr().err
