"""Workflow helpers against vectors produced by the reference's own function bodies (tests/golden/drivers.json, made by
tests/golden/make_golden_drivers.py): region splitting (split_fasta.py:94-101), overlap stitching (merge_fasta.py:8-39),
VaryParams (Params.py:31-60); the polish driver and the region sharding on the CPU oracle."""
import copy
import json
import os
import random

import numpy as np
import pytest

import backends as B
import golden_util as G
from poreseq_amd import consensus, synth
from poreseq_amd import dist as psdist
from poreseq_amd.util import DEFAULT_PARAMS, VaryParams

Z = json.load(open(os.path.join(G.GOLDEN, "drivers.json")))
P = dict(DEFAULT_PARAMS, verbose=0)


def test_split_regions_equals_reference_loop():
    for c in Z["split"]:
        got = consensus.split_regions(c["length"], c["region_length"])
        assert len(got) == c["n_regions"], c
        assert [list(r) for r in got[:3]] == c["first"] and [list(r) for r in got[-3:]] == c["last"], c
        if c["regions"] is not None:
            assert [list(r) for r in got] == c["regions"]
    assert len(consensus.split_regions(4600000, 10000)) == 512      # config #5: 512 regions (the last one 1 kb short)


def test_merge_seqs_equals_reference_function():
    for c in Z["merge"]:
        assert consensus.merge_seqs(c["seq1"], c["seq2"], c["overlap"], swalign=B.oracle_swalign) == c["merged"]


@pytest.mark.gpu
def test_hip_merge_seqs_equals_reference_function():
    """the same vectors (merge_fasta.py:8-39 run in memory) with the HIP Smith-Waterman behind `swalign`"""
    from poreseq_amd.poreseqcpp import swalign
    for c in Z["merge"]:
        assert consensus.merge_seqs(c["seq1"], c["seq2"], c["overlap"], swalign=swalign) == c["merged"]
        assert consensus.merge_seqs(c["seq1"], c["seq2"], c["overlap"]) == c["merged"]        # the driver's default backend is the HIP library


def test_vary_params_equals_reference_function():
    v = Z["vary"]
    params = dict(zip(v["params_keys"], v["params_vals"]))
    random.seed(v["seed"])
    got = VaryParams(params)
    assert [[p[k] for k in v["params_keys"]] for p in got] == v["lists"]      # same draws, same keys, same order


def test_shard_longest_first_is_balanced_and_complete():
    lens = [10000] * 9 + [3500, 10000, 700]
    parts = [psdist.shard(list(range(len(lens))), r, 4, weights=lens) for r in range(4)]
    assert sorted(i for p in parts for i, _ in p) == list(range(len(lens)))          # every region exactly once
    loads = [sum(lens[i] for i, _ in p) for p in parts]
    assert max(loads) - min(loads) <= 10000 and max(loads) <= 30000                   # at most one region of imbalance
    assert psdist.shard(list("abcde"), 1, 2) == [(1, "b"), (3, "d")]                  # unweighted: round-robin, as before


def _genome_regions(Lg, E, seed, sw, rl=1200, ov=1000):
    """a small 'assembly': truth, overlapping regions and a loader that builds each region's PSAlign on demand"""
    rng = np.random.default_rng(seed)
    truth = synth.random_sequence(rng, Lg)

    def loader(cls):
        def make(a, b):
            draft, events, t = synth.make_region(b - a, E, seed + a, sw, P, truth=truth[a:b])
            par = dict(P)
            par.pop("end_trim")            # keep the full region so that neighbours still share `overlap` bases
            return B.make_pa(cls, draft, events, par)
        return make
    return truth, loader


def test_polish_driver_on_the_oracle_equals_hand_rolled_pipeline(monkeypatch):
    """split -> consensus per region -> merge, through `polish`, against the same steps written out (the two short tail
    regions are also re-run one by one from a fresh random stream; the stitching is re-done by hand)"""
    truth, loader = _genome_regions(1500, 5, 700, B.oracle_swalign)
    monkeypatch.setattr(consensus.poreseqcpp, "swalign", lambda a, b, api=None: B.oracle_swalign(a, b))
    whole, parts = consensus.polish(truth, loader(B.OraclePSAlign), params=None, region_length=1400, overlap=1000, batch=4, reps=1)
    regs = consensus.split_regions(len(truth), 1400)
    assert [(a, b) for a, b, _, _ in parts] == regs == [(0, 1400), (400, 1500), (800, 1500), (1200, 1500)]
    for k in (2, 3):
        B.reset_rand()
        pa = loader(B.OraclePSAlign)(*regs[k])
        assert consensus.consensus_region(pa, None, reps=1)[0] == parts[k][2]
    want = parts[0][2]
    for _, _, nxt, _ in parts[1:]:
        want = consensus.merge_seqs(want, nxt, 1000, swalign=B.oracle_swalign)
    assert whole == want
    assert B.oracle_swalign(whole, truth)[0] > 93.0            # (one round only: the draft is ~89 % accurate)
