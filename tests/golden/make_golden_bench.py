#!/usr/bin/env python3
"""Digests of bench.py's OWN regions after the reference's full consensus schedule (build container only).

bench.py refines, per timed step, one of (at most) three sets of synthetic 10 kb / 10-event regions; region 0 of set s on rank 0 is
`synth.make_region(10000, 10, 1002 + s, ...)`.  This script runs the REAL reference C++ (oracle/_ref/libps_ref.so, compiled from
/root/reference/cpp by oracle/Makefile) through the full Mutate.py schedule on those regions — from a fresh rand() stream, as a fresh
`poreseq consensus` process would — and stores the SHA-256 of the final (trimmed) sequence, its length and accuracy, with the
checksum of the generated inputs.  bench.py compares the GPU's result for that region of EVERY timed step (`parity_in_run`).
~18 min of one core per 10 kb region; the regions run in parallel processes.

    python tests/golden/make_golden_bench.py        -> tests/golden/bench_regions.json
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [(10000, 10, 1002), (10000, 10, 1003), (10000, 10, 1004), (1000, 10, 5000)]     # (bases, events, seed): bench.py's region 0 of each set; its 1 kb region


def one(case):
    L, E, seed = case
    import backends as B
    import golden_util as G
    from poreseq_amd import synth
    from poreseq_amd.consensus import consensus_region
    from poreseq_amd.util import DEFAULT_PARAMS
    assert B.have_ref(), "build oracle/_ref first (make -C oracle)"
    params = dict(DEFAULT_PARAMS, verbose=0)
    d, ev, tr = synth.make_region(L, E, seed, B.ref_swalign, params)
    digest_in = G.input_digest(d, ev, tr)
    pa = B.make_pa(B.RefPSAlign, d, ev, params)
    B.reset_rand()
    log = []
    t = time.time()
    seq, acc = consensus_region(pa, params, log=log)
    return {"length": L, "events": E, "seed": seed, "input_sha256": digest_in,
            "sequence_sha256": hashlib.sha256(seq.encode("ascii")).hexdigest(), "sequence_len": len(seq), "accuracy": acc,
            "calls": [c for c, _, _ in log], "nbases": [int(n) for _, n, _ in log], "reference_seconds": time.time() - t}


if __name__ == "__main__":
    with mp.get_context("spawn").Pool(len(CASES)) as pool:
        res = pool.map(one, CASES, chunksize=1)
    out = {"source": "oracle/_ref/libps_ref.so (the reference's cpp/*.cpp compiled where they lie) through poreseq_amd.consensus.consensus_region; "
                     "tests/golden/make_golden_bench.py", "regions": res}
    with open(os.path.join(HERE, "bench_regions.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
