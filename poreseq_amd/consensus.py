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


def consensus_region(pa, params=None, reps=4, verbose=0, refseq=None, log=None):
    """Run the consensus schedule in place on `pa`; returns (sequence, accuracy_vs_refseq).

    Mirrors Mutate.py:39-101: fewer than 5 events -> return the input untouched (Mutate.py:50-53);
    Mutate('self', reps); up to `reps` x {Mutate('viterbi') (inner reps 4), Refine()} stopping when
    Refine changes nothing; end_trim; accuracy by swalign against the loaded reference.
    `log`, when given, receives (call, nbases, sequence) after every PSAlign call.
    """
    params = pa.params if params is None else params
    if 'verbose' not in pa.params:
        pa.params['verbose'] = 0
    sw = lambda a, b: poreseqcpp.swalign(a, b, pa._native)
    refseq = pa.sequence if refseq is None else refseq
    if len(pa.events) < 5:
        if verbose > 0:
            sys.stderr.write("Coverage is 1 or 2, not mutating...\n")
        return (refseq, 100)
    if verbose > 0:
        sys.stderr.write("Mutating {} bases using {} events\n".format(len(refseq), len(pa.events)))
    nb = pa.Mutate(reps=reps)
    if log is not None:
        log.append(("Mutate:self", nb, pa.sequence))
    if verbose > 0:
        sys.stderr.write("Accuracy: " + str(round(sw(pa.sequence, refseq)[0], 1)) + "%\n")
    for _ in range(reps):
        nb = pa.Mutate(seqs='viterbi')
        if log is not None:
            log.append(("Mutate:viterbi", nb, pa.sequence))
        nbases = pa.Refine()
        if log is not None:
            log.append(("Refine", nbases, pa.sequence))
        if verbose > 0:
            sys.stderr.write("Accuracy: " + str(round(sw(pa.sequence, refseq)[0], 1)) + "%\n")
        if nbases == 0:
            break
    if 'end_trim' in params and len(pa.sequence) > 2 * params['end_trim']:
        pa.sequence = pa.sequence[int(params['end_trim']):-int(params['end_trim'])]
    acc, inds = sw(pa.sequence, refseq)
    if verbose > 0:
        errs = np.sum(np.array(inds) == 0, 0)
        sys.stderr.write("Final accuracy: " + str(round(acc, 1)) + "%\n")
        sys.stderr.write("Insertions: {}, Deletions: {}\n".format(errs[0], errs[1]))
        sys.stderr.write("Final coverage: " + str(round(np.mean(pa.Coverage()), 1)) + "X\n")
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


def train(make_pa, params, refseq, iters=1, reps=10, save=None, paramlists=None, in_flight=1):
    """Transition-parameter search of `poreseq train` (cmdline.py:246-267): each iteration runs the consensus
    schedule (reps = 10) once per perturbed parameter set and keeps the most accurate one.

    make_pa(params) must return a freshly loaded PSAlign for the training region with the event models'
    transition probabilities taken from `params` (what LoadAlignedEvents + setparams do in the reference);
    the 16 replicas are independent region work-items.  `paramlists` (optional) replaces VaryParams, e.g.
    for reproducible tests.  Returns (best params, best accuracy per iteration).

    in_flight > 1 runs that many replicas concurrently on the GPU (one host thread each).  The reference runs them
    one after another in one process, so the stochastic Viterbi seeds of replica k continue the rand() stream
    of replica k-1; concurrent replicas each continue their own thread's stream instead (as 16 separate
    `poreseq consensus` processes would), which can change which replica wins by chance — keep 1 for parity.
    """
    from .util import VaryParams, SaveParams
    best_accs = []
    for it in range(iters):
        paramlist = paramlists[it] if paramlists is not None else VaryParams(params)
        def one(p):
            return consensus_region(make_pa(p), p, reps=reps, refseq=refseq)[1]
        if in_flight > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=int(in_flight)) as pool:
                accs = list(pool.map(one, paramlist))
        else:
            accs = [one(p) for p in paramlist]
        params = paramlist[int(np.argmax(accs))]
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
    """Join two region consensus sequences that share `overlap` bases, at the middle of their
    Smith-Waterman alignment (merge_fasta.py:8-39, same thresholds and index arithmetic)."""
    swalign = poreseqcpp.swalign if swalign is None else swalign
    i0, i1 = -overlap, overlap
    if len(seq1) < overlap:
        i0 = 0
    if len(seq2) < overlap:
        i1 = len(seq2) - 1
    acc, inds = swalign(seq1[i0:], seq2[:i1])
    if acc < 0.70:  # sic: a percentage compared with 0.70 (merge_fasta.py:32)
        raise Exception('Insufficient accuracy for overlap')
    inds = [x for x in inds if x[0] > 0 and x[1] > 0]
    imid = inds[int(len(inds) / 2)]
    i0 += imid[0]
    i1 = imid[1]
    return seq1[:i0] + seq2[i1:]
