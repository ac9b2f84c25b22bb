#!/usr/bin/env python3
"""Generate the BASELINE-sized golden vectors from the REAL reference (build container only).

Same reference build as make_golden.py (the reference's own Cython `PSAlign`, compiled into a temporary
directory outside the repo).  The inputs of these cases are megabytes of random floats, so the fixtures keep
only what pins them — the generator arguments and a SHA-256 of every generated input array — plus the
reference's outputs; tests regenerate the inputs with `poreseq_amd.synth` (seeded numpy streams) and refuse to
run on a checksum mismatch.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_large.py [case ...]

Cases (SURVEY.md section 8d / VERDICT round 1 item 1):
  score10k_E10      10 kb, 10 events: ScoreEvents, ScoreMutations of 2000 point + 20 multi-base edits at
                    scoring_width 100 and at scoring_width 20 (the point_width of Refine)
  score10k_E30      10 kb, 30 events (one region of config #3): the same calls
  consensus_L1000   full Mutate.py schedule, 1 kb, 10 events (the north_star comparison point)
  consensus_L1500   full schedule, 1.5 kb, 10 events, default widths (bands narrower than the columns)
  consensus_L3000   full schedule, 3 kb, 10 events
  consensus_L10000_E10   full schedule at BASELINE config #2 size (10 kb, 10 events; ~15-25 min of reference time).  The final
                    ref_align vectors are stored as int32, the final ref_like vectors as SHA-256 digests (760 KB of random
                    mantissas otherwise); every intermediate sequence and count is stored
  mutate_seeds_L10000    one `Mutate(seqs=[4 seed strings], reps=2)` call at 10 kb / 10 events (FindMutations' 5-strip
                    Smith-Waterman chains and candidate batches with a caller's seed list), then ScoreEvents
"""
import copy
import ctypes
import hashlib
import os
import sys
import tempfile
import time

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from poreseq_amd import synth  # noqa: E402
from poreseq_amd.util import DEFAULT_PARAMS, MutationInfo  # noqa: E402
import make_golden as MG  # noqa: E402


def input_digest(draft, events, truth):
    """SHA-256 over every generated input, in a fixed order (shared with tests/golden_util.py)."""
    h = hashlib.sha256()
    h.update(draft.encode("ascii")); h.update(b"|"); h.update(truth.encode("ascii"))
    for ev in events:
        m = ev.model
        for a in (ev.mean, ev.stdv, ev.ref_align, ev.ref_like, m.level_mean, m.level_stdv, m.sd_mean, m.sd_stdv):
            h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        h.update(np.array([m.prob_skip, m.prob_stay, m.prob_extend, m.prob_insert], dtype=np.float64).tobytes())
        h.update(ev.sequence.encode("ascii"))
    return h.hexdigest()


def edit_list(draft, seed, npoint=2000):
    """2000 random point edits + 20 multi-base edits spread over the region (deterministic)."""
    rng = np.random.default_rng(seed)
    muts = synth.random_point_mutations(rng, draft, npoint)
    L = len(draft)
    for k in range(20):
        st = int(rng.integers(5, L - 80))
        no, nm = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        if no == 0 and nm == 0:
            nm = 3
        mi = MutationInfo()
        mi.start, mi.orig, mi.mut = st, draft[st:st + no], synth.random_sequence(rng, nm)
        muts.append(mi)
    return muts


def main():
    want = sys.argv[1:] or ["score10k_E10", "score10k_E30", "consensus_L1000", "consensus_L1500", "consensus_L3000"]
    tmp = tempfile.mkdtemp(prefix="poreseq_ref_")
    ref = MG.build_reference_module(tmp)
    libc = ctypes.CDLL(None)
    P = dict(DEFAULT_PARAMS)
    P["verbose"] = 0

    def mk(draft, events, params):
        pa = ref.PSAlign()
        pa.sequence = draft
        pa.events = copy.deepcopy(events)
        pa.params = dict(params)
        return pa

    def header(L, E, seed, par, draft, events, truth):
        return {"L": np.array(L), "E": np.array(E), "seed": np.array(seed),
                "params_keys": np.array(sorted(par)), "params_vals": np.array([par[k] for k in sorted(par)]),
                "input_sha256": np.array(input_digest(draft, events, truth)), "draft_len": np.array(len(draft))}

    for name, L, E, seed in [("score10k_E10", 10000, 10, 2102), ("score10k_E30", 10000, 30, 2103)]:
        if name not in want:
            continue
        t0 = time.time()
        draft, events, truth = synth.make_region(L, E, seed, ref.swalign, P)
        out = header(L, E, seed, P, draft, events, truth)
        out["ScoreEvents"] = np.array(mk(draft, events, P).ScoreEvents())
        muts = edit_list(draft, seed)
        out["muts_start"] = np.array([m.start for m in muts], dtype=np.int32)
        out["muts_orig"] = np.array([m.orig for m in muts])
        out["muts_mut"] = np.array([m.mut for m in muts])
        for tag, sw in (("sw100", 100.0), ("sw20", 20.0)):
            par = dict(P, scoring_width=sw)
            sc = mk(draft, events, par).ScoreMutations(muts)
            out["ScoreMutations_" + tag] = np.array([s.score for s in sc], dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, "%.1f s" % (time.time() - t0), flush=True)

    for name, L, E, seed in [("consensus_L1000", 1000, 10, 2201), ("consensus_L1500", 1500, 10, 2202),
                             ("consensus_L3000", 3000, 10, 2203), ("consensus_L10000_E10", 10000, 10, 2204)]:
        if name not in want:
            continue
        t0 = time.time()
        libc.srand(1)  # rand() is never seeded by the reference; this equals a fresh process
        draft, events, truth = synth.make_region(L, E, seed, ref.swalign, P)
        out = header(L, E, seed, P, draft, events, truth)
        pa = mk(draft, events, P)
        calls, nb, seqs = [], [], []
        calls.append("Mutate:self"); nb.append(pa.Mutate(reps=4)); seqs.append(pa.sequence)
        for _ in range(4):
            calls.append("Mutate:viterbi"); nb.append(pa.Mutate(seqs="viterbi")); seqs.append(pa.sequence)
            calls.append("Refine"); n = pa.Refine(); nb.append(n); seqs.append(pa.sequence)
            if n == 0:
                break
        out["calls"] = np.array(calls)
        out["nbases"] = np.array(nb)
        out["sequences"] = np.array(seqs)
        big = L >= 10000
        for e, ev in enumerate(pa.events):
            if big:   # compact form: int32 alignment (values are integers), digest of the likelihoods
                assert np.array_equal(ev.ref_align, np.rint(ev.ref_align))
                out["final_ev%d_ref_align_i32" % e] = ev.ref_align.astype(np.int32)
                out["final_ev%d_ref_like_sha256" % e] = np.array(hashlib.sha256(np.ascontiguousarray(ev.ref_like, dtype=np.float64).tobytes()).hexdigest())
            else:
                out["final_ev%d_ref_align" % e] = ev.ref_align
                out["final_ev%d_ref_like" % e] = ev.ref_like
        out["final_ScoreEvents"] = np.array(pa.ScoreEvents())
        out["final_accuracy"] = np.array(ref.swalign(pa.sequence, truth)[0])
        out["reference_seconds"] = np.array(time.time() - t0)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, list(zip(calls, nb)), float(out["final_accuracy"]), "%.1f s" % (time.time() - t0), flush=True)

    if "mutate_seeds_L10000" in want:
        name, L, E, seed = "mutate_seeds_L10000", 10000, 10, 2205
        t0 = time.time()
        libc.srand(1)
        draft, events, truth = synth.make_region(L, E, seed, ref.swalign, P)
        out = header(L, E, seed, P, draft, events, truth)
        rng = np.random.default_rng(seed + 7)
        seeds = [synth.corrupt(rng, truth, 0.02, 0.02, 0.02) for _ in range(4)]
        pa = mk(draft, events, P)
        out["seeds"] = np.array(seeds)
        out["nbases"] = np.array(pa.Mutate(seqs=list(seeds), reps=2))
        out["sequence"] = np.array(pa.sequence)
        for e, ev in enumerate(pa.events):
            out["final_ev%d_ref_align_i32" % e] = ev.ref_align.astype(np.int32)
            out["final_ev%d_ref_like_sha256" % e] = np.array(hashlib.sha256(np.ascontiguousarray(ev.ref_like, dtype=np.float64).tobytes()).hexdigest())
        out["final_ScoreEvents"] = np.array(pa.ScoreEvents())
        out["reference_seconds"] = np.array(time.time() - t0)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, int(out["nbases"]), "%.1f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
