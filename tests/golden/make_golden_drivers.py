#!/usr/bin/env python3
"""Golden vectors for the workflow helpers, produced by the REFERENCE'S OWN FUNCTION BODIES (build container only):

  split    the region loop of poreseq/split_fasta.py:94-101, executed as it stands on a set of sequence lengths
  merge    poreseq/merge_fasta.py:8-39 (merge_seqs), executed as it stands with the reference's compiled swalign
  vary     poreseq/Params.py:31-60 (VaryParams) after random.seed(7)

The reference's modules cannot be imported (Python-2 imports, Biopython); the function text is read from /root/reference at
generation time, compiled in memory and run — nothing of it is written anywhere.  Only inputs and outputs are stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_drivers.py
"""
import ast
import json
import os
import random
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from poreseq_amd import synth  # noqa: E402
from poreseq_amd.util import DEFAULT_PARAMS  # noqa: E402

REF = MG.REF


def function_from(path, name, env):
    """compile one top-level function of a reference file in memory"""
    src = open(path).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    code = compile(ast.Module(body=[fn], type_ignores=[]), path, "exec")
    exec(code, env)
    return env[name]


def main():
    tmp = tempfile.mkdtemp(prefix="poreseq_ref_")
    ref = MG.build_reference_module(tmp)
    out = {}

    # ---- split: lines 94-101 of split_fasta.py, run for one reference id at a time
    lines = open(os.path.join(REF, "poreseq", "split_fasta.py")).read().splitlines()[93:101]
    body = "\n".join(l[8:] for l in lines)          # the loop body is indented by two levels
    cases = []
    for length, rl in [(35000, 10000), (48500, 10000), (4600000, 10000), (9999, 10000), (10000, 10000), (10001, 10000),
                       (1000, 10000), (500, 10000), (19000, 10000), (12345, 5000), (1, 10000)]:
        env = {"refseq": "A" * length if length < 100000 else type("S", (), {"__len__": lambda self, n=length: n})(),
               "region_length": rl, "regions": [], "refid": "r"}
        exec(compile(body, "split_fasta.py:94-101", "exec"), env)
        regs = [[int(x) for x in s.split(":")[1:]] for s in env["regions"]]
        cases.append({"length": length, "region_length": rl, "regions": regs if len(regs) <= 64 else None,
                      "n_regions": len(regs), "first": regs[:3], "last": regs[-3:]})
    out["split"] = cases

    # ---- merge: merge_seqs as it stands, with the reference's swalign
    merge_seqs = function_from(os.path.join(REF, "poreseq", "merge_fasta.py"), "merge_seqs", {"swalign": ref.swalign})
    rng = np.random.default_rng(31)
    mcases = []
    genome = synth.random_sequence(rng, 6000)
    for (a0, a1, b0, b1, err, ov) in [(0, 3000, 2000, 5000, 0.01, 1000), (0, 2500, 1500, 4000, 0.03, 1000), (100, 1500, 1000, 2400, 0.02, 500),
                                      (0, 800, 300, 1600, 0.01, 1000), (0, 1200, 700, 1300, 0.02, 1000), (0, 3000, 2000, 5000, 0.0, 1000)]:
        s1 = synth.corrupt(rng, genome[a0:a1], err, err, err)
        s2 = synth.corrupt(rng, genome[b0:b1], err, err, err)
        mcases.append({"seq1": s1, "seq2": s2, "overlap": ov, "merged": merge_seqs(s1, s2, ov)})
    out["merge"] = mcases

    # ---- vary: VaryParams after random.seed(7)
    VaryParams = function_from(os.path.join(REF, "poreseq", "Params.py"), "VaryParams", {"random": random})
    params = dict(DEFAULT_PARAMS)
    random.seed(7)
    pl = VaryParams(params)
    out["vary"] = {"seed": 7, "params_keys": list(params.keys()), "params_vals": [params[k] for k in params],
                   "lists": [[p[k] for k in params] for p in pl]}

    with open(os.path.join(HERE, "drivers.json"), "w") as f:
        json.dump(out, f)
    print("wrote drivers.json:", len(out["split"]), "split cases,", len(mcases), "merge cases, 16 parameter sets")


if __name__ == "__main__":
    main()
