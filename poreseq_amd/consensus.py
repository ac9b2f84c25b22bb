"""Host drivers that reproduce the *call schedule* of the reference's workflow layer on top of
`poreseqcpp.PSAlign`:

  consensus_region  <- poreseq/Mutate.py:8-101   (Mutate('self') then {Mutate('viterbi'), Refine()})
  variant_region    <- poreseq/Variant.py:66-95  (ScoreMutations / ScorePoints with start offsetting)
  split_regions     <- poreseq/split_fasta.py:94-101 (max_length pieces with 1 kb overlap)

fast5 / BAM loading is out of scope: callers hand over a PSAlign whose events are already
loaded (synthetic here), exactly what `LoadAlignedEvents` would have returned.
"""
import sys

import numpy as np

from . import poreseqcpp


def _report(verbose, text):
    if verbose > 0:
        sys.stderr.write(text + "\n")


def consensus_region(pa, params=None, reps=4, verbose=0, refseq=None, log=None):
    """Run the consensus schedule in place on `pa`; returns (sequence, accuracy_vs_refseq).

    The call sequence is the reference's (Mutate.py:39-101) and has to be: a region with fewer than 5 events is handed
    back untouched (Mutate.py:50-53); otherwise Mutate('self', reps), then up to `reps` rounds of Mutate('viterbi')
    followed by Refine(), ending after the first Refine that changes nothing; `end_trim` bases come off both ends;
    the accuracy is the swalign identity against the sequence the region was loaded with.
    `log`, when given, receives (call, nbases, sequence) after every PSAlign call; `verbose` > 0 prints progress.
    """
    params = pa.params if params is None else params
    pa.params.setdefault('verbose', 0)
    identity = lambda a, b: poreseqcpp.swalign(a, b, pa._native)
    refseq = pa.sequence if refseq is None else refseq
    if len(pa.events) < 5:
        _report(verbose, "fewer than 5 events: region returned as loaded")
        return (refseq, 100)
    _report(verbose, "refining %d bases with %d events" % (len(refseq), len(pa.events)))

    def call(name, fn):
        n = fn()
        if log is not None:
            log.append((name, n, pa.sequence))
        return n

    call("Mutate:self", lambda: pa.Mutate(reps=reps))
    if verbose > 0:
        _report(verbose, "identity after seeding from the reads: %.1f%%" % identity(pa.sequence, refseq)[0])
    for _ in range(reps):
        call("Mutate:viterbi", lambda: pa.Mutate(seqs='viterbi'))
        changed = call("Refine", pa.Refine)
        if verbose > 0:
            _report(verbose, "identity: %.1f%%" % identity(pa.sequence, refseq)[0])
        if changed == 0:
            break
    trim = int(params['end_trim']) if 'end_trim' in params else 0
    if 'end_trim' in params and len(pa.sequence) > 2 * params['end_trim']:
        pa.sequence = pa.sequence[trim:-trim]
    acc, pairs = identity(pa.sequence, refseq)
    if verbose > 0:
        gaps = np.sum(np.array(pairs) == 0, 0)
        _report(verbose, "final identity %.1f%%, %d insertions, %d deletions, coverage %.1fX"
                % (acc, gaps[0], gaps[1], np.mean(pa.Coverage())))
    return (pa.sequence, acc)


def consensus_regions(pas, params=None, reps=4, refseqs=None, logs=None, batch=None, resident=True):
    """The consensus schedule of `consensus_region` for several independent regions in lock-step (poreseq_amd.batch):
    every PSAlign call of the schedule is issued once for all regions that still take part in it, so each phase is one
    launch chain on the GPU.  Returns [(sequence, accuracy)] in the order of `pas`; each entry equals what
    `consensus_region(pa)` returns for that region run on its own from a fresh process.
    `logs`, when given, is a list of lists receiving (call, nbases, sequence) per region after every call.
    `batch`: an already loaded RegionBatch over `pas` (events resident on the GPU, see RegionBatch.load); it is closed here.
    """
    from .batch import RegionBatch
    n = len(pas)
    refseqs = [pa.sequence for pa in pas] if refseqs is None else list(refseqs)
    out = [None] * n
    todo = []
    for i, pa in enumerate(pas):
        if 'verbose' not in pa.params:
            pa.params['verbose'] = 0
        if len(pa.events) < 5:                      # Mutate.py:50-53
            out[i] = (refseqs[i], 100)
        else:
            todo.append(i)
    if not todo and batch is not None:
        batch.close()                               # (every region had fewer than five events: nothing ran, the handles still go)
    if todo:
        def note(i, call, nb):
            if logs is not None:
                logs[i].append((call, nb, pas[i].sequence))
        with (batch if batch is not None else RegionBatch(pas, resident=resident)) as rb:
            tot = rb.Mutate(todo, reps=reps)
            for i in todo:
                note(i, "Mutate:self", tot[i])
            live = list(todo)
            for _ in range(reps):
                if not live:
                    break
                tot = rb.Mutate(live, seqs='viterbi')
                for i in live:
                    note(i, "Mutate:viterbi", tot[i])
                nb = rb.Refine(live)
                for i in live:
                    note(i, "Refine", nb[i])
                live = [i for i in live if nb[i] != 0]
        api = pas[todo[0]]._native
        for i in todo:
            pa = pas[i]
            p = pa.params if params is None else params
            if 'end_trim' in p and len(pa.sequence) > 2 * p['end_trim']:
                pa.sequence = pa.sequence[int(p['end_trim']):-int(p['end_trim'])]
            out[i] = (pa.sequence, poreseqcpp.swalign(pa.sequence, refseqs[i], api)[0])
    return out


def variant_region(pa, muts, region_start=0, params=None, out=None):
    """Score `muts` (or every point edit when the list is empty) as Variant.py:66-95 does: starts are
    region-relative inside the call and absolute again in the returned / printed MutationScores."""
    for m in muts:
        m.start -= region_start
    mutscores = pa.ScoreMutations(muts) if len(muts) > 0 else pa.ScorePoints()
    for ms in mutscores:
        ms.start += region_start
        if out is not None:
            out.write(str(ms) + '\n')
    return mutscores


def train(make_pa, params, refseq, iters=1, reps=10, save=None, paramlists=None, in_flight=1, lock_step=False):
    """Transition-parameter search of `poreseq train` (cmdline.py:246-267): every iteration runs the consensus schedule
    (reps = 10) once per candidate parameter set (VaryParams: 16 of them) on the same region and keeps the most accurate.

    make_pa(params) returns a freshly loaded PSAlign of the training region whose event models carry the transition
    probabilities of `params` (LoadAlignedEvents + setparams in the reference).  `paramlists` (optional, one list per
    iteration) replaces VaryParams, e.g. for reproducible tests.  Returns (best params, best accuracy per iteration).

    The candidates are independent replicas of one region:
      lock_step=True   all of them go through the schedule as ONE lock-step batch (poreseq_amd.batch): a single launch
                       chain per phase, the GPU-native mode.  Each replica draws its stochastic Viterbi seeds from the
                       generator of a fresh process, i.e. the result is that of 16 separate `poreseq consensus` runs.
      in_flight=k      k replicas at a time on host threads (same random-stream semantics as lock_step).
      default          one after another in this thread: replica k continues the rand() stream where replica k-1
                       stopped, which is what the reference's single process does — bit-parity mode.
    Which candidate wins can differ between the first two and the last by the luck of the seeds; keep the default
    when comparing against the reference.
    """
    from .util import VaryParams, SaveParams
    best_accs = []
    for it in range(iters):
        cands = paramlists[it] if paramlists is not None else VaryParams(params)

        def one(p):
            return consensus_region(make_pa(p), p, reps=reps, refseq=refseq)[1]

        if lock_step:
            pas = [make_pa(p) for p in cands]
            accs = [acc for _, acc in consensus_regions(pas, None, reps=reps, refseqs=[refseq] * len(pas))]
        elif in_flight > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=int(in_flight)) as pool:
                accs = list(pool.map(one, cands))
        else:
            accs = [one(p) for p in cands]
        params = cands[int(np.argmax(accs))]
        if save:
            SaveParams(save, params)
        best_accs.append(max(accs))
    return params, best_accs


def split_regions(length, region_length=10000):
    """[(start, end)] region work-items: region_length pieces stepping by region_length - 1000
    (split_fasta.py:94-101; a 35 000-base sequence gives 0:10000, 9000:19000, 18000:28000, 27000:35000)."""
    region_length = int(region_length)
    out = []
    dl = region_length - 1000
    istart, iend = 0, min(region_length, length)
    while istart < iend:
        out.append((istart, iend))
        iend = min(iend + dl, length)
        istart = min(istart + dl, length)
    return out


def merge_seqs(seq1, seq2, overlap=1000, swalign=None):
    """Stitch two region sequences whose ends share about `overlap` bases: the tail of `seq1` is aligned with the head
    of `seq2` and the join is made at the middle aligned pair.

    Index arithmetic as merge_fasta.py:8-39, quirks included, so that stitched assemblies are identical: a `seq2` shorter
    than the overlap contributes all but its last base to the alignment; the identity threshold compares a percentage
    with 0.70; the cut into `seq1` is counted from its END (a middle pair on the tail's last base therefore cuts at 0,
    i.e. drops `seq1` — the reference's behaviour for degenerate overlaps).
    """
    swalign = poreseqcpp.swalign if swalign is None else swalign
    tail_from = -overlap if len(seq1) >= overlap else 0
    head_to = overlap if len(seq2) >= overlap else len(seq2) - 1
    acc, pairs = swalign(seq1[tail_from:], seq2[:head_to])
    if acc < 0.70:
        raise Exception('Insufficient accuracy for overlap')
    both = [(a, b) for a, b in pairs if a > 0 and b > 0]
    a_mid, b_mid = both[int(len(both) / 2)]
    return seq1[:tail_from + a_mid] + seq2[b_mid:]


def polish(sequence, make_region_pa, params=None, region_length=10000, overlap=1000, batch=16, reps=4, refine=None, swalign=None):
    """Assembly polish = the reference's split -> consensus per region -> merge pipeline (split_fasta.py:50-133,
    `poreseq consensus` per region file, merge_fasta.py:41-80) as one call.

    sequence         the draft to polish (only its length and the region coordinates are used here)
    make_region_pa   callable (start, end) -> PSAlign loaded with the draft slice and the events overlapping it
                     (what LoadAlignedEvents returns for region 'start:end')
    batch            regions refined in lock-step per GPU (poreseq_amd.batch); `refine` replaces the default
                     `poreseq_amd.dist.refine_regions` (regions sharded over the ranks of the process group, longest first)
    swalign          the Smith-Waterman the stitching uses (default: this package's, on the GPU; tests of the driver pass a checker's)
    Returns (polished sequence, [(start, end, region consensus, accuracy)]).  With an `end_trim` in `params` the region
    sequences lose that many bases per end before stitching, exactly as the region FASTA files of the reference do.
    """
    from . import dist as psdist
    regs = split_regions(len(sequence), region_length)
    refine = psdist.refine_regions if refine is None else refine
    done = refine(regs, make_region_pa, params=params, batch=batch, reps=reps)
    merged = done[0][0]
    for seq, _ in done[1:]:
        merged = merge_seqs(merged, seq, overlap, swalign)
    return merged, [(a, b, s, acc) for (a, b), (s, acc) in zip(regs, done)]
