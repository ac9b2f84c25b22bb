#!/usr/bin/env python3
"""Golden vectors for the array half of the real-data front end, produced by the REFERENCE'S OWN CODE run in
memory (build container only):

  event    the body of poreseq/EventData.py PSEvent (class node compiled as it stands: __init__ lines 100-175, flip 182-224,
           mapaligns 226-256) on stand-ins for the h5py datasets a fast5 file would hold — model scaling, drift correction, the
           k-mer walk that seeds ref_align, the complement flip permutation of the 1024 model states
  bam      poreseq/LoadData.py:67-153 (EventsFromBAM) on stand-ins for pysam's AlignmentFile and its records — overlap filter,
           descending-overlap order, one alignment per read name up to max_coverage, hard-clip and region offsets of the aligned
           pairs, reverse-strand flip, mapaligns

The reference's modules cannot be imported (h5py, pysam and Biopython are absent; LoadData.py holds a Python-2 print statement):
the class / function text is read from /root/reference at generation time, compiled in memory and run against the stand-ins —
nothing of it is written anywhere.  Only inputs and outputs are stored (tests/golden/frontend.npz).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_frontend.py
"""
import ast
import copy
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


class Seq:   # the one Biopython call the class makes: str(Seq(s).reverse_complement())
    def __init__(self, s):
        self.s = s

    def reverse_complement(self):
        return "".join(COMP[c] for c in reversed(self.s))


class Attrs(dict):
    pass


class Dataset:
    """an h5py dataset stand-in: field access on a structured array, [()] for a scalar string, .attrs"""
    def __init__(self, value, attrs=None):
        self.value = value
        self.attrs = Attrs(attrs or {})

    def __getitem__(self, k):
        if isinstance(k, tuple) and k == ():
            return self.value
        return self.value[k]

    def __len__(self):
        return len(self.value)


def fake_fast5(rng, n_levels, seq, loc_names=("template", "complement")):
    """the datasets PSEvent.__init__ reads, with random but plausible contents; returns {path: Dataset} and the raw inputs"""
    store, raw = {}, {}
    base = "/Analyses/Basecall_2D_000/"
    nal = len(seq) - 4
    kmers = np.array([seq[i:i + 5] for i in range(nal)])
    al = np.zeros(nal, dtype=[("template", "i8"), ("complement", "i8"), ("kmer", "U5")])
    al["kmer"] = kmers
    for loc in loc_names:
        n = n_levels[loc]
        ev = np.zeros(n, dtype=[("mean", "f8"), ("stdv", "f8"), ("length", "f8"), ("start", "f8")])
        ev["mean"] = rng.normal(60, 8, n)
        ev["stdv"] = np.abs(rng.normal(1.2, 0.3, n)) + 0.2
        ev["length"] = np.abs(rng.normal(0.02, 0.01, n)) + 0.002
        ev["start"] = 100.0 + np.cumsum(ev["length"])
        md = np.zeros(1024, dtype=[("level_mean", "f8"), ("level_stdv", "f8"), ("sd_mean", "f8"), ("sd_stdv", "f8")])
        md["level_mean"] = rng.normal(55, 10, 1024)
        md["level_stdv"] = np.abs(rng.normal(1.0, 0.2, 1024)) + 0.3
        md["sd_mean"] = np.abs(rng.normal(1.1, 0.2, 1024)) + 0.3
        md["sd_stdv"] = np.abs(rng.normal(0.4, 0.1, 1024)) + 0.1
        att = {"shift": float(rng.normal(2, 1)), "scale": float(rng.normal(1.05, 0.03)), "scale_sd": float(rng.normal(0.95, 0.03)),
               "drift": float(rng.normal(0.002, 0.001)), "var": float(rng.normal(1.1, 0.05)), "var_sd": float(rng.normal(1.3, 0.05)),
               "model_file": "model_%s.model" % loc}
        # alignment of the 2D sequence's k-mers to this strand's levels: increasing level indices with holes (-1 = not aligned)
        idx = np.sort(rng.choice(np.arange(1, n), size=min(nal, n - 1), replace=False))
        col = np.full(nal, -1, dtype=np.int64)
        col[:idx.size] = idx
        col[rng.random(nal) < 0.15] = -1
        al[loc] = col
        store[base + "BaseCalled_%s/Events" % loc] = Dataset(ev)
        store[base + "BaseCalled_%s/Model" % loc] = Dataset(md)
        store[base + "Summary/basecall_1d_%s" % loc] = Dataset(None, att)
        raw[loc] = {"events": ev, "model": md, "attrs": att}
    store[base + "BaseCalled_2D/Fastq"] = Dataset("@read\n%s\n+\n%s\n" % (seq, "!" * len(seq)))
    store[base + "BaseCalled_2D/Alignment"] = Dataset(al)
    raw["alignment"] = al
    raw["sequence"] = seq
    return store, raw


def reference_psevent(files):
    """the reference's PSEvent class (and MakeContiguous), compiled from its own text against stand-ins for h5py / Biopython"""
    path = os.path.join(REF, "poreseq", "EventData.py")
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name in ("PSEvent", "PSModel")) or
            (isinstance(n, ast.FunctionDef) and n.name == "MakeContiguous")]
    h5py = types.SimpleNamespace(File=lambda fn, mode: files[fn])
    env = {"h5py": h5py, "np": np, "copy": copy, "Seq": Seq}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), env)
    return env["PSEvent"]


def event_out(ev, tag, out):
    for k in ("mean", "stdv", "length", "start", "ref_align", "ref_like"):
        out["%s_%s" % (tag, k)] = np.array(getattr(ev, k), dtype=np.float64)
    for k in ("level_mean", "level_stdv", "sd_mean", "sd_stdv"):
        out["%s_model_%s" % (tag, k)] = np.array(getattr(ev.model, k), dtype=np.float64)
    out["%s_sequence" % tag] = np.array(ev.sequence)
    out["%s_flipped" % tag] = np.array(bool(ev.flipped))
    out["%s_complement" % tag] = np.array(bool(ev.model.complement))
    out["%s_model_name" % tag] = np.array(str(ev.model.name))


class BamRecord:
    """what EventsFromBAM reads of a pysam.AlignedSegment"""
    def __init__(self, name, ref_start, pairs, is_reverse, hard_clip):
        self.query_name = name
        self.is_reverse = is_reverse
        self.pairs = pairs
        self.ref_start = ref_start
        self.ref_end = max(p[1] for p in pairs if p[1] is not None) + 1
        self.cigar = ([(5, hard_clip)] if hard_clip else []) + [(0, len(pairs))]

    def get_overlap(self, start, end):
        return max(0, min(end, self.ref_end) - max(start, self.ref_start))

    def get_aligned_pairs(self):
        return list(self.pairs)


def main():
    rng = np.random.default_rng(20240)
    out = {}
    from poreseq_amd import synth
    # ---- event: three fake reads, template + complement each
    files, raws = {}, {}
    names = ["readA.fast5", "readB.fast5", "readC.fast5", "readD.fast5"]
    for name in names:
        seq = synth.random_sequence(rng, int(rng.integers(180, 260)))
        store, raw = fake_fast5(rng, {"template": int(rng.integers(300, 400)), "complement": int(rng.integers(300, 400))}, seq)
        files[name] = store
        raws[name] = raw
    PSEvent = reference_psevent(files)
    out["event_names"] = np.array(names[:3])
    for name in names:
        raw = raws[name]
        tag = name.split(".")[0]
        out["%s_in_sequence" % tag] = np.array(raw["sequence"])
        out["%s_in_alignment" % tag] = raw["alignment"]
        for loc in ("template", "complement"):
            out["%s_in_%s_events" % (tag, loc)] = raw[loc]["events"]
            out["%s_in_%s_model" % (tag, loc)] = raw[loc]["model"]
            for k, v in raw[loc]["attrs"].items():
                out["%s_in_%s_attr_%s" % (tag, loc, k)] = np.array(v)
    for name in names[:3]:
        tag = name.split(".")[0]
        for loc in ("t", "c"):
            ev = PSEvent(name, loc)
            event_out(ev, "%s_%s" % (tag, loc), out)
            ev2 = ev.copy()
            ev2.flip()                                   # the reverse-strand flip: sequence and aligned indices too
            event_out(ev2, "%s_%s_flip" % (tag, loc), out)

    # ---- bam: EventsFromBAM on stand-in records over a 600-base reference region [150, 550)
    path = os.path.join(REF, "poreseq", "LoadData.py")
    lines = open(path).read().splitlines()
    fn_src = "\n".join(lines[66:153])                    # def EventsFromBAM ... return events
    fn_src = fn_src.replace("print str(e.message)", "pass")   # (the file's one Python-2 print statement, in an except branch)
    recs = []
    spec = [("readA.fast5", 100, 240, False, 0), ("readB.fast5", 300, 230, True, 7), ("readC.fast5", 140, 200, False, 3),
            ("readA.fast5", 420, 120, False, 0),      # a second alignment of readA (both beyond max_coverage here)
            ("readD.fast5", 520, 200, False, 0),      # overlap 30: below min_overlap
            ("missing.fast5", 200, 220, False, 0)]    # most overlap of all, but no such file: both strands raise and are skipped
    for name, rstart, nal, rev, clip in spec:
        qlen = len(raws[name]["sequence"]) if name in raws else 260
        nal = min(nal, qlen - clip - 5)
        q, r, pairs = 0, rstart, []
        while q < nal:
            u = rng.random()
            if u < 0.05:
                pairs.append((q, None)); q += 1          # insertion in the read
            elif u < 0.10:
                pairs.append((None, r)); r += 1          # deletion
            else:
                pairs.append((q, r)); q += 1; r += 1
        recs.append(BamRecord(name, rstart, pairs, rev, clip))
    order = [3, 0, 4, 2, 5, 1]                           # the file's own (coordinate-independent) order of records: the sort must not rely on it
    recs = [recs[k] for k in order]

    class AlignmentFile:
        def __init__(self, fn, mode):
            self.nreferences = 1
            self.references = ["chr"]

        def fetch(self, reference=None, start=None, end=None):
            return [x for x in recs if x.get_overlap(start, end) > 0]

    class RegionInfo:
        def __init__(self):
            self.name, self.start, self.end = None, 150, 550

    env = {"pysam": types.SimpleNamespace(AlignmentFile=AlignmentFile), "os": os, "np": np, "PSEvent": PSEvent}
    exec(compile(fn_src, "LoadData.py:67-153", "exec"), env)
    params = {"min_overlap": 100, "max_coverage": 3, "min_coverage": 2}
    # (event "files" are looked up by os.path.join(eventdir, query_name): an empty eventdir keeps the names)
    events = env["EventsFromBAM"]("", "x.bam", RegionInfo(), params)
    out["bam_n_events"] = np.array(len(events))
    for k, ev in enumerate(events):
        event_out(ev, "bam_ev%d" % k, out)
    out["bam_rec_name"] = np.array([r.query_name for r in recs])
    out["bam_rec_ref_start"] = np.array([r.ref_start for r in recs])
    out["bam_rec_ref_end"] = np.array([r.ref_end for r in recs])
    out["bam_rec_is_reverse"] = np.array([r.is_reverse for r in recs])
    out["bam_rec_hard_clip"] = np.array([r.cigar[0][1] if r.cigar[0][0] == 5 else 0 for r in recs])
    for k, r in enumerate(recs):
        out["bam_rec%d_pairs" % k] = np.array([[-1 if a is None else a, -1 if b is None else b] for a, b in r.pairs], dtype=np.int64)
    out["bam_region"] = np.array([150, 550])
    out["bam_params"] = np.array([params["min_overlap"], params["max_coverage"], params["min_coverage"]])
    # insufficient coverage raises
    try:
        env["EventsFromBAM"]("", "x.bam", RegionInfo(), dict(params, min_coverage=9))
        out["bam_insufficient_raises"] = np.array(False)
    except Exception as e:
        out["bam_insufficient_raises"] = np.array(str(e) == "Insufficient coverage!")
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), **out)
    print("wrote frontend.npz:", len(out), "arrays;", int(out["bam_n_events"]), "events from the BAM stand-in")


if __name__ == "__main__":
    main()
