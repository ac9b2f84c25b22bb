"""SURVEY.md 8(f4) carried into the HIP path: events BUILT BY THE FRONT END — `PSEvent.from_basecall` on parsed fast5 tables (template and
flipped complement strands: all 1024 model rows permuted, levels reversed), `events_from_bam_records` (read selection, hard-clip /
region offsets, reverse-strand `flip`, `mapaligns`) — go through `PSAlign.ScoreEvents` / `ScorePoints` / `Refine` on the HIP library
and must equal the CPU oracle bit for bit (reference: poreseq/EventData.py:100-256, LoadData.py:67-153, then pyx:189-472).

Two sources of tables: the reference-generated vectors of tests/golden/frontend.npz (random levels: the DP runs on whatever it is
given) and reads simulated from 5-mer models (synth), written back into the file layout a fast5 file holds — complement strands
un-flipped, reverse-strand reads reverse-complemented — so that the alignments are real ones."""
import copy
import os

import numpy as np
import pytest

import backends as B
import golden_util as G
from poreseq_amd import loaddata, synth
from poreseq_amd.events import PSEvent, reverse_complement
from poreseq_amd.poreseqcpp import PSAlign
from poreseq_amd.util import DEFAULT_PARAMS

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(G.GOLDEN, "frontend.npz"), allow_pickle=False)
P0 = dict(DEFAULT_PARAMS, verbose=0)


def scores(ms):
    return np.array([m.score for m in ms])


def both(seq, events, params, refine=True):
    """ScoreEvents, ScorePoints (and Refine) of the HIP library and of the oracle on copies of the same events"""
    out = []
    for cls in (PSAlign, B.OraclePSAlign):
        pa = B.make_pa(cls, seq, copy.deepcopy(events), params)
        res = [pa.ScoreEvents(), scores(pa.ScorePoints())]
        if refine:
            res += [pa.Refine(), pa.sequence, [e.ref_align.copy() for e in pa.events], [e.ref_like.copy() for e in pa.events]]
        out.append(res)
    hip, orc = out
    assert hip[0] == orc[0]
    assert np.array_equal(hip[1], orc[1])
    if refine:
        assert hip[2] == orc[2] and hip[3] == orc[3]
        for x, y in zip(hip[4] + hip[5], orc[4] + orc[5]):
            assert np.array_equal(x, y)
    return hip


def load_fixture(tag, loc):
    name = "template" if loc == "t" else "complement"
    attrs = {k: Z["%s_in_%s_attr_%s" % (tag, name, k)][()] for k in ("shift", "scale", "scale_sd", "drift", "var", "var_sd", "model_file")}
    al = Z["%s_in_alignment" % tag]
    return PSEvent.from_basecall(Z["%s_in_%s_events" % (tag, name)], Z["%s_in_%s_model" % (tag, name)], attrs, str(Z["%s_in_sequence" % tag]),
                                 al[name], al["kmer"], complement=(loc == "c"))


class FixtureRec:
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


def test_reference_vector_events_through_the_hip_path():
    """the four events EventsFromBAM builds from the fixture's records (two strands of a forward and of a reverse-strand read: one
    template, one flipped complement, both flipped once more) + every strand of every fixture read on its own 2D sequence"""
    recs = [FixtureRec(k) for k in range(len(Z["bam_rec_name"]))]
    start, end = (int(x) for x in Z["bam_region"])
    mo, mc, mn = (int(x) for x in Z["bam_params"])

    def loader(name, loc):
        tag = name.split(".")[0]
        if "%s_in_sequence" % tag not in Z.files:
            raise IOError("no such fast5 file")
        return load_fixture(tag, loc)
    events = loaddata.events_from_bam_records(recs, loader, start, end, {"min_overlap": mo, "max_coverage": mc, "min_coverage": mn})
    assert [ev.model.complement for ev in events] == [False, True, False, True] and any(ev.flipped != ev.model.complement for ev in events)
    for ev in events:
        ev.setparams(P0)
    region = synth.random_sequence(np.random.default_rng(11), end - start)
    both(region, events, dict(P0, realign_width=60.0, scoring_width=15.0), refine=False)
    # every strand against the read's own 2D sequence, where from_basecall's seed alignment points
    for tag in (str(n).split(".")[0] for n in Z["event_names"]):
        evs = [load_fixture(tag, "t"), load_fixture(tag, "c")]
        for ev in evs:
            ev.setparams(P0)
        both(str(Z["%s_in_sequence" % tag]), evs, dict(P0, realign_width=80.0), refine=False)


# ---- simulated reads written into the file layout and read back through the front end ----------------------------------------------
def tables_of(ev, complement, rng):
    """The parsed tables of a fast5 strand whose `from_basecall` event is (up to rounding) `ev`, an event in its READ's orientation
    with ref_align = 1-based positions in the read's 2D sequence: levels and model go back to the file's orientation (a complement
    strand is stored un-flipped, EventData.py:173-175), the scaling attributes are divided out, the 2D alignment table lists for every
    5-mer of the sequence the level aligned to it."""
    f = ev.copy()
    if complement:
        f.flip(False)
    n = f.mean.size
    att = {"shift": float(rng.normal(2, 1)), "scale": float(rng.normal(1.05, 0.03)), "scale_sd": float(rng.normal(0.95, 0.03)),
           "drift": float(rng.normal(0.002, 0.001)), "var": float(rng.normal(1.1, 0.05)), "var_sd": float(rng.normal(1.3, 0.05)),
           "model_file": "sim_%s.model" % ("complement" if complement else "template")}
    length = np.abs(rng.normal(0.02, 0.01, n)) + 0.002
    start = 100.0 + np.cumsum(length)
    table = {"mean": f.mean + att["drift"] * (start - start[0]), "stdv": f.stdv.copy(), "length": length, "start": start}
    model = {"level_mean": (f.model.level_mean - att["shift"]) / att["scale"], "level_stdv": f.model.level_stdv / att["var"],
             "sd_mean": f.model.sd_mean / att["scale_sd"], "sd_stdv": f.model.sd_stdv * np.sqrt(att["var_sd"])}
    seq = ev.sequence
    nk = len(seq) - 4
    col = np.full(nk, -1, dtype=np.int64)
    for u in range(1, n):                                   # (level 0 can never be listed: `alinds > 0`, EventData.py:161)
        p = int(f.ref_align[u])
        if 1 <= p <= nk:
            col[p - 1] = u
    kmers = np.array([seq[i:i + 5] for i in range(nk)])
    return table, model, att, col, kmers


@pytest.mark.parametrize("L,E", [(500, 6), (1200, 5)])
def test_simulated_reads_through_from_basecall_flip_and_bam_records(L, E):
    rng = np.random.default_rng(100 + L)
    truth_draft, sim, truth = synth.make_region(L, E, 8100 + L, B.oracle_swalign, P0, draft_error=0.0)
    lead = synth.random_sequence(rng, 37)
    draft = synth.corrupt(rng, truth, 0.03, 0.03, 0.03)                    # the region of the reference the reads are aligned to
    pairs = B.oracle_swalign(truth, draft)[1]                               # 1-based, 0 = gap
    files, recs = {}, []

    class Rec:
        pass
    for r in range(E // 2):                                                 # read r = events 2r (template) and 2r + 1 (complement)
        reverse = r % 2 == 1
        strands = []
        for e in (2 * r, 2 * r + 1):
            ev = sim[e].copy()
            ev.sequence = truth
            if reverse:
                ev.flip()                                                   # the read as its own file sees it: the other strand of the reference
            strands.append(tables_of(ev, e % 2 == 1, rng))
        name = "sim%d.fast5" % r
        files[name] = (strands, reverse_complement(truth) if reverse else truth)
        rec = Rec()
        rec.query_name, rec.is_reverse = name, reverse
        clip = 3 if r == 0 else 0
        # pysam reports a reverse-strand read's pairs in reference orientation (the read reverse-complemented): indices into `truth`
        ap = [((a - 1 - clip) if a > 0 and a - 1 >= clip else None, (b - 1 + len(lead)) if b > 0 else None) for a, b in pairs]
        rec.cigar = ([(5, clip)] if clip else []) + [(0, len(ap))]
        rec.get_aligned_pairs = lambda ap=ap: list(ap)
        rec.get_overlap = lambda s, e: e - s
        recs.append(rec)

    def loader(name, loc):
        (t, c), seq = files[name]
        table, model, att, col, kmers = t if loc == "t" else c
        return PSEvent.from_basecall(table, model, att, seq, col, kmers, complement=(loc == "c"))
    events = loaddata.events_from_bam_records(recs, loader, len(lead), len(lead) + len(draft), {"max_coverage": 30})
    assert len(events) == 2 * (E // 2) and [ev.model.complement for ev in events] == [False, True] * (E // 2)
    for ev in events:
        ev.setparams(P0)
        assert np.count_nonzero(ev.ref_align > 0) > 0.5 * ev.mean.size       # the alignments arrived on the region
    hip = both(draft, events, P0)
    assert min(hip[0]) > 0.7 * len(draft)                                     # real alignments (~1 per base against a 91 % draft; 2.3 against the truth)
    assert hip[2] > 0                                                         # Refine repairs bases of the 91 % draft
