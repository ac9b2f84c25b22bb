#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (run in the build container only).

Builds the reference's own Cython module (poreseq/_poreseqcpp.pyx + cpp/*.cpp) into a
temporary directory outside the repo, imports it, runs its `PSAlign` / `swalign` /
`seqtostates` on seeded synthetic inputs and stores inputs + outputs.  Only the data files
are committed; no reference source, object or bytecode enters the repo
(PYTHONDONTWRITEBYTECODE keeps the reference tree clean too).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import copy
import ctypes
import glob
import os
import subprocess
import sys
import sysconfig
import tempfile

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PORESEQ_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from poreseq_amd import synth  # noqa: E402
from poreseq_amd.util import DEFAULT_PARAMS  # noqa: E402


def build_reference_module(tmp):
    """SURVEY.md 8c recipe: cython --cplus on a renamed temp copy of the pyx, g++ -std=c++0x -O3."""
    pyx = os.path.join(tmp, "poreseqcpp.pyx")
    with open(os.path.join(REF, "poreseq", "_poreseqcpp.pyx")) as f:
        src = f.read()
    with open(pyx, "w") as f:
        f.write(src)
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-2", "-X", "c_string_type=unicode",
                           "-X", "c_string_encoding=ascii", "-I", REF, pyx, "-o", os.path.join(tmp, "poreseqcpp.cpp")])
    inc = ["-I" + REF, "-I" + os.path.join(REF, "cpp"), "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include()]
    flags = ["-std=c++0x", "-O3", "-fPIC", "-w"]
    objs = []
    for s in ["FindMutations", "Alignment", "swlib", "EventUtil", "Viterbi"]:
        o = os.path.join(tmp, s + ".o")
        subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(REF, "cpp", s + ".cpp"), "-o", o])
        objs.append(o)
    o = os.path.join(tmp, "MakeMutations.o")  # pointer `> 0` is a hard error on g++ >= 11 (see oracle/Makefile)
    sed = subprocess.Popen(["sed", "s/if (likes > 0)/if (likes != 0)/", os.path.join(REF, "cpp", "MakeMutations.cpp")],
                           stdout=subprocess.PIPE)
    subprocess.check_call(["g++"] + flags + inc + ["-x", "c++", "-c", "-", "-o", o], stdin=sed.stdout)
    sed.wait()
    objs.append(o)
    o = os.path.join(tmp, "poreseqcpp.o")
    subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(tmp, "poreseqcpp.cpp"), "-o", o])
    objs.append(o)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    subprocess.check_call(["g++", "-shared", "-o", os.path.join(tmp, "poreseqcpp" + ext)] + objs)
    sys.path.insert(0, os.path.join(REF, "poreseq"))  # the pyx does `from Util import ...`
    sys.path.insert(0, tmp)
    import poreseqcpp
    return poreseqcpp


def pack_events(events):
    d = {}
    for e, ev in enumerate(events):
        m = ev.model
        d["ev%d_mean" % e] = ev.mean
        d["ev%d_stdv" % e] = ev.stdv
        d["ev%d_ref_align" % e] = ev.ref_align
        d["ev%d_ref_like" % e] = ev.ref_like
        d["ev%d_model" % e] = np.stack([m.level_mean, m.level_stdv, m.sd_mean, m.sd_stdv])
        d["ev%d_trans" % e] = np.array([m.prob_skip, m.prob_stay, m.prob_extend, m.prob_insert])
        d["ev%d_complement" % e] = np.array(m.complement)
        d["ev%d_sequence" % e] = np.array(ev.sequence)
    d["n_events"] = np.array(len(events))
    return d


def pack_scores(prefix, scores):
    return {prefix + "_start": np.array([s.start for s in scores], dtype=np.int32),
            prefix + "_orig": np.array([s.orig for s in scores]),
            prefix + "_mut": np.array([s.mut for s in scores]),
            prefix + "_score": np.array([s.score for s in scores], dtype=np.float64)}


def main():
    tmp = tempfile.mkdtemp(prefix="poreseq_ref_")
    ref = build_reference_module(tmp)
    libc = ctypes.CDLL(None)
    P = dict(DEFAULT_PARAMS)
    P["verbose"] = 0

    def mk(draft, events, params):
        pa = ref.PSAlign()
        pa.sequence = draft
        pa.events = copy.deepcopy(events)
        pa.params = dict(params)
        return pa

    # ---- case A: config #1 shape, scaled (ScoreEvents / ScorePoints / ScoreMutations) -------------
    for name, L, E, seed, par in [("score_L300_E5", 300, 5, 1101, dict(P)),
                                  ("score_L240_E4_narrow", 240, 4, 1102, dict(P, realign_width=40.0, scoring_width=15.0, point_width=6.0))]:
        draft, events, truth = synth.make_region(L, E, seed, ref.swalign, par)
        out = {"sequence": np.array(draft), "truth": np.array(truth),
               "params_keys": np.array(sorted(par)), "params_vals": np.array([par[k] for k in sorted(par)])}
        out.update(pack_events(events))
        out["ScoreEvents"] = np.array(mk(draft, events, par).ScoreEvents())
        out.update(pack_scores("ScorePoints", mk(draft, events, par).ScorePoints()))
        rng = np.random.default_rng(seed)
        muts = synth.random_point_mutations(rng, draft, 40)
        # a few multi-base edits, an edit at the very end and one past the state list
        for st, o, m in [(10, draft[10:13], "ACGTA"), (50, draft[50:56], ""), (L - 3, draft[L - 3:L - 2], "G"),
                         (len(draft), "", "A"), (0, "", "TT"), (2, draft[2:3], "")]:
            mi = ref.MutationInfo() if hasattr(ref, "MutationInfo") else None
            from poreseq_amd.util import MutationInfo
            mi = MutationInfo()
            mi.start, mi.orig, mi.mut = st, o, m
            muts.append(mi)
        out["muts_start"] = np.array([m.start for m in muts], dtype=np.int32)
        out["muts_orig"] = np.array([m.orig for m in muts])
        out["muts_mut"] = np.array([m.mut for m in muts])
        out.update(pack_scores("ScoreMutations", mk(draft, events, par).ScoreMutations(muts)))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name)

    # ---- case B: full consensus schedule (Mutate.py:70-85) from process start --------------------
    libc.srand(1)  # rand() is never seeded by the reference; this equals a fresh process
    L, E, seed = 400, 6, 1201
    draft, events, truth = synth.make_region(L, E, seed, ref.swalign, P)
    out = {"sequence": np.array(draft), "truth": np.array(truth),
           "params_keys": np.array(sorted(P)), "params_vals": np.array([P[k] for k in sorted(P)])}
    out.update(pack_events(events))
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
    for e, ev in enumerate(pa.events):
        out["final_ev%d_ref_align" % e] = ev.ref_align
        out["final_ev%d_ref_like" % e] = ev.ref_like
    out["final_accuracy"] = np.array(ref.swalign(pa.sequence, truth)[0])
    np.savez_compressed(os.path.join(HERE, "consensus_L400_E6.npz"), **out)
    print("wrote consensus", list(zip(calls, nb)), out["final_accuracy"])

    # ---- case C: swalign / seqtostates / viterbi (deterministic, nkeep = 0 via shim not exposed) -
    rng = np.random.default_rng(77)
    s1 = synth.random_sequence(rng, 257)
    s2 = synth.corrupt(rng, s1, 0.06, 0.06, 0.06)
    acc, pairs = ref.swalign(s1, s2)
    odd = s1[:40] + "-" + s1[40:90] + "N" + s1[90:120]
    np.savez_compressed(os.path.join(HERE, "sw_states.npz"), s1=np.array(s1), s2=np.array(s2), accuracy=np.array(acc),
                        pairs=np.array(pairs, dtype=np.int32), odd=np.array(odd),
                        odd_states=np.array(ref.seqtostates(odd), dtype=np.int32),
                        s1_states=np.array(ref.seqtostates(s1), dtype=np.int32))
    print("wrote sw_states")
    for f in glob.glob(os.path.join(REF, "poreseq", "__pycache__")):
        print("WARNING: reference tree has", f)


if __name__ == "__main__":
    main()
