"""Read selection of the reference's BAM front end on PARSED records (poreseq/LoadData.py:67-153).

pysam is not in this image, so nothing here opens a BAM file: `events_from_bam_records` takes what `AlignmentFile.fetch` would have
returned — any objects with `query_name`, `is_reverse`, `cigar`, `get_overlap(start, end)` and `get_aligned_pairs()` (pysam's own
AlignedSegment has them) — and a callable that loads one strand of one read (`poreseq_amd.events.PSEvent.from_basecall` on the
tables of its fast5 file).  What it does with them is the reference's: overlap filter, descending-overlap order, one alignment per
read name up to max_coverage, hard-clip and region offsets of the aligned pairs, reverse-strand flip, `mapaligns`.
Vectors: tests/golden/frontend.npz (the reference's function run on stand-in records, tests/golden/make_golden_frontend.py).
"""
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
            except Exception:
                pass
    if not events:
        raise Exception("No aligned reads found!")
    return events
