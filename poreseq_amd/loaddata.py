"""The reference's BAM front end (poreseq/LoadData.py:10-153).

`events_from_bam` / `load_aligned_events` open the files the way the reference does (pysam for the BAM file, h5py for the fast5
files: guarded imports, neither is in the build image).  The work is in `events_from_bam_records`, which takes what
`AlignmentFile.fetch` returns — any objects with `query_name`, `is_reverse`, `cigar`, `get_overlap(start, end)` and `get_aligned_pairs()` (pysam's own
AlignedSegment has them) — and a callable that loads one strand of one read (`poreseq_amd.events.PSEvent.from_basecall` on the
tables of its fast5 file).  What it does with them is the reference's: overlap filter, descending-overlap order, one alignment per
read name up to max_coverage, hard-clip and region offsets of the aligned pairs, reverse-strand flip, `mapaligns`.
Vectors: tests/golden/frontend.npz (the reference's function run on stand-in records, tests/golden/make_golden_frontend.py).
"""
import os

import numpy as np


def select_records(records, start, end, params):
    """LoadData.py:92-121: records overlapping [start, end) by at least params['min_overlap'], most overlap first (Python's stable
    sort keeps the file order among equals, as the reference's does), the first alignment of every read name, at most
    params['max_coverage'] of them; fewer than params['min_coverage'] alignments raise."""
    recs = list(records)
    if "min_overlap" in params:
        recs = [x for x in recs if x.get_overlap(start, end) >= params["min_overlap"]]
    recs.sort(key=lambda x: x.get_overlap(start, end), reverse=True)
    if "min_coverage" in params and len(recs) < params["min_coverage"]:
        raise Exception("Insufficient coverage!")
    names, keep = [], []
    for r in recs:
        if r.query_name not in names:
            names.append(r.query_name)
            keep.append(r)
        if "max_coverage" in params and len(keep) >= params["max_coverage"]:
            break
    return keep


def region_pairs(record, start):
    """LoadData.py:129-137: the (read index, reference index) pairs of one record — gaps dropped, read indices shifted by a leading
    hard clip, reference indices made relative to the region's start."""
    aps = np.array([x for x in record.get_aligned_pairs() if x[0] is not None and x[1] is not None])
    cig0 = record.cigar[0]
    if cig0[0] == 5:
        aps[:, 0] += cig0[1]
    if start > 0:
        aps[:, 1] -= start
    return aps


def events_from_bam_records(records, load_event, start, end, params):
    """EventsFromBAM (LoadData.py:67-153) from parsed records: `load_event(query_name, 't' | 'c')` returns the strand's PSEvent or
    raises (a read without that strand is skipped, as the reference's try / except does); reverse-strand reads are flipped; every
    event's ref_align is mapped from its own 2D sequence onto the region through the record's aligned pairs."""
    events = []
    for rec in select_records(records, start, end, params):
        aps = region_pairs(rec, start)
        for loc in ("t", "c"):
            try:
                ev = load_event(rec.query_name, loc)
                if rec.is_reverse:
                    ev.flip()
                ev.mapaligns(aps)
                events.append(ev)
            except ImportError:
                raise                      # a missing reader module is not "a read without that strand"
            except Exception as e:         # LoadData.py:146-147: the strand is skipped, the error shown
                if params.get("verbose", 0):
                    import sys
                    sys.stderr.write("Skipping %s strand %s: %s\n" % (rec.query_name, loc, e))
    if not events:
        raise Exception("No aligned reads found!")
    return events


def events_from_bam(eventdir, bamfile, reginfo, params):
    """EventsFromBAM (LoadData.py:67-153): the events of the reads a BAM file aligns to `reginfo` (name / start / end; a missing name
    is taken from a single-reference BAM file and written back, as the reference does), each strand loaded from
    `eventdir/<query name>` by `PSEvent.from_fast5`.  Needs pysam (and h5py): without it the call fails with an ImportError that says so."""
    try:
        import pysam
    except ImportError as e:
        raise ImportError("events_from_bam reads BAM files through pysam, which is not installed; "
                          "events_from_bam_records takes the parsed records") from e
    try:
        import h5py  # noqa: F401 — PSEvent.from_fast5 reads every strand through it: fail here, not once per strand
    except ImportError as e:
        raise ImportError("events_from_bam loads every strand with PSEvent.from_fast5, which needs h5py (not installed)") from e
    from .events import PSEvent
    bam = pysam.AlignmentFile(bamfile, "rb")
    if reginfo.name is None:
        if bam.nreferences > 1:
            raise Exception("Multiple references in BAM, one must be specified!")
        reginfo.name = bam.references[0]
    records = list(bam.fetch(reference=reginfo.name, start=reginfo.start, end=reginfo.end))
    return events_from_bam_records(records, lambda name, loc: PSEvent.from_fast5(os.path.join(eventdir, name), loc),
                                   reginfo.start, reginfo.end, params)


def load_reference(fastafile, refname=None):
    """LoadReference (LoadData.py:53-65) without Biopython: the named record of a FASTA file (the only one when no name is given)."""
    refs, name = {}, None
    with open(fastafile) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                name = line[1:].split()[0] if len(line) > 1 else ""
                refs[name] = []
            elif line and name is not None:
                refs[name].append(line)
    if refname is None:
        if len(refs) != 1:
            raise Exception("Multiple references in fasta, must specify one")
        refname = next(iter(refs))
    return "".join(refs[refname])


def load_aligned_events(fastafile, bamfile, eventdir, reginfo, params, psalign=None):
    """LoadAlignedEvents (LoadData.py:10-51): the PSAlign of one region — reference slice, the events aligned to it, the parameters
    (`setparams` on every event).  `psalign`: the class to build (default: this package's drop-in PSAlign)."""
    refseq = load_reference(fastafile, reginfo.name)
    if reginfo.start is None and reginfo.end is None:
        reginfo.start, reginfo.end = 0, len(refseq)
    events = events_from_bam(eventdir, bamfile, reginfo, params)
    if len(params) > 0:
        for ev in events:
            ev.setparams(params)
    if psalign is None:
        from .poreseqcpp import PSAlign as psalign
    pa = psalign()
    pa.sequence = refseq[reginfo.start:reginfo.end]
    pa.events = events
    pa.params = params
    return pa
