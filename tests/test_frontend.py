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
