"""GPU test of the distributed plumbing: bench.py launched through torch.distributed.run (as the driver does
for N > 1) with one rank, and the RCCL gather path of poreseq_amd.dist forced on with a one-rank group."""
import json
import os
import subprocess
import sys

import pytest

import backends as B

pytestmark = pytest.mark.gpu


def test_bench_under_torchrun_single_rank():
    env = dict(os.environ, PORESEQ_FORCE_PG="1")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                   "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(B.ROOT, "bench.py"),
                                   "--gpus", "1", "--steps", "1", "--warmup", "0", "--length", "1000", "--regions-per-gpu", "6",
                                   "--batches-in-flight", "2", "--no-cpu"],
                                  env=env, timeout=900, stderr=subprocess.STDOUT)
    line = [l for l in out.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["unit"] == "kb/s" and r["scaling"] == "weak"
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["achieved"] > 0 and r["roofline"]["kernel"] in ("k_fill", "k_sweep", "k_sweeps", "k_sweep_w", "k_sweeps_w")
    assert r["north_star_1kb"]["lock_step_kb_s"] > r["north_star_1kb"]["single_region_kb_s"] > 0
    assert r["config"]["batches_in_flight"] == 2 and r["single_region_s"] > 0
    assert r["accuracy"]["consensus_percent"] > 97.0


def test_bench_refuses_gpus_without_matching_world():
    p = subprocess.run([sys.executable, os.path.join(B.ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, timeout=300)
    assert p.returncode == 2 and b"WORLD_SIZE" in p.stderr


def test_rccl_gather_of_region_results():
    code = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
os.environ["PORESEQ_FORCE_PG"] = "1"
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ["MASTER_PORT"] = "29519"
from poreseq_amd import dist as psdist, synth
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.util import DEFAULT_PARAMS
rank, local, world = psdist.init()
import torch.distributed as dist
assert dist.is_initialized() and dist.get_backend() == "nccl"
P = dict(DEFAULT_PARAMS, verbose=0)
def process(reg):
    L, E, seed = reg
    draft, events, truth = synth.make_region(L, E, seed, swalign, P)
    pa = PSAlign(); pa.sequence, pa.events, pa.params = draft, events, dict(P)
    sc = pa.ScoreEvents(); pa.Refine()
    return pa.sequence, np.array(sc)
res = psdist.run_regions([(300, 5, 1), (320, 5, 2), (280, 4, 3)], process, max_events=8)
print(json.dumps({"n": len(res), "lens": [len(r[0]) for r in res], "tmax": psdist.max_over_ranks(2.5), "s0": res[0][1][:5].tolist()}))
''' % B.ROOT
    out = subprocess.check_output([sys.executable, "-c", code], timeout=900, stderr=subprocess.STDOUT)
    r = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])
    assert r["n"] == 3 and all(l > 250 for l in r["lens"]) and r["tmax"] == 2.5 and all(s > 0 for s in r["s0"])


def test_bench_two_ranks_share_the_gpu_over_gloo():
    """bench.py's N > 1 path end to end — barrier, max over ranks, whole-job value, core pinning, process-group tear-down — with two
    ranks on the one GPU of the test box (gloo instead of RCCL, which refuses two ranks per device; no data-path collective is involved)"""
    env = dict(os.environ, PORESEQ_DIST_BACKEND="gloo")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29519", os.path.join(B.ROOT, "bench.py"),
                                   "--gpus", "2", "--steps", "1", "--warmup", "0", "--length", "1000", "--regions-per-gpu", "6",
                                   "--batches-in-flight", "2", "--no-cpu", "--no-extras"],
                                  env=env, timeout=900, stderr=subprocess.STDOUT)
    line = [l for l in out.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert abs(r["value"] - 2 * 6 * 1.0 / (r["ms_per_step"] / 1e3)) < 1e-6 * r["value"]      # whole-job rate over the slowest rank's time
    assert r["resident"]["value"] >= r["value"]


@pytest.mark.gpu
def test_bench_eight_ranks_share_the_gpu_over_gloo():
    """the driver's N = 8 command line end to end at tiny size — rendezvous, one runtime set per rank, core slices, per-step
    barrier + max over ranks, whole-job value, tear-down — with eight ranks on the one GPU of the test box (gloo).  Every rank
    plans its slabs, shares and pool ceiling for an eighth of the device (dist.init -> PORESEQ_DEVICE_FRACTION, ps_info reports it)"""
    env = dict(os.environ, PORESEQ_DIST_BACKEND="gloo")
    for k in ("PORESEQ_SLAB_GB", "PORESEQ_SLABS", "PORESEQ_DEVICE_FRACTION"):
        env.pop(k, None)
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                                   "--master-addr", "127.0.0.1", "--master-port", "29523", os.path.join(B.ROOT, "bench.py"),
                                   "--gpus", "8", "--steps", "2", "--warmup", "0", "--length", "1000", "--regions-per-gpu", "4",
                                   "--batches-in-flight", "2", "--no-cpu", "--no-extras"],
                                  env=env, timeout=1500, stderr=subprocess.STDOUT)
    line = [l for l in out.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 8 and r["scaling"] == "weak" and r["steps"] == 2 and r["value"] > 0
    assert abs(r["value"] - 8 * 4 * 1.0 * 2 / (2 * r["ms_per_step"] / 1e3)) < 1e-6 * r["value"]   # all ranks' regions over the sum of the steps' slowest-rank times
    assert len(r["batch_done_s"]) == 2 and "runtimes" in r["library"]
    assert "device fraction of this process 0.125" in r["library"]


@pytest.mark.gpu
def test_hip_event_sharded_scoring_and_deltas():
    """per-event score terms from the HIP library (ps_score_mutation_deltas): equal to the oracle's, their ordered sum equal to
    ScoreMutations; the event-sharded driver on one rank (the two-rank exchange runs on gloo in tests/test_dist.py)"""
    import copy
    import numpy as np
    import backends as B
    from poreseq_amd import dist as psdist, synth
    from poreseq_amd.poreseqcpp import PSAlign
    from poreseq_amd.util import DEFAULT_PARAMS
    P = dict(DEFAULT_PARAMS, verbose=0)
    draft, events, truth = synth.make_region(700, 6, 4343, B.oracle_swalign, P)
    events[2].ref_align[:] = 0
    muts = synth.random_point_mutations(np.random.default_rng(4), draft, 300)
    muts[7].start = len(draft) + 1
    mk = lambda cls: B.make_pa(cls, draft, copy.deepcopy(events), P)
    dh, do = mk(PSAlign).ScoreMutationDeltas(muts), mk(B.OraclePSAlign).ScoreMutationDeltas(muts)
    assert dh.shape == (6, 300) and np.array_equal(dh, do)
    want = mk(PSAlign).ScoreMutations(muts)
    s = np.full(300, -1e-6)
    for e in range(6):
        s = s + dh[e]
    assert np.array_equal(s, np.array([w.score for w in want]))
    got = psdist.score_mutations_event_sharded(mk(PSAlign), muts)
    assert [g.score for g in got] == [w.score for w in want]


@pytest.mark.gpu
def test_hip_variant_regions_single_rank_equals_oracle():
    """`poreseq variant` through the package driver on the HIP library (one rank: the region-sharded branch; the event-sharded branch and
    both on two ranks run on gloo in tests/test_dist.py): absolute starts, scores bit-identical to the oracle's per-region ScoreMutations"""
    import copy
    import numpy as np
    import backends as B
    from poreseq_amd import dist as psdist, synth
    from poreseq_amd.consensus import variant_region
    from poreseq_amd.poreseqcpp import PSAlign
    from poreseq_amd.util import DEFAULT_PARAMS
    P = dict(DEFAULT_PARAMS, verbose=0, scoring_width=100.0)
    regions = [(0, 900), (5000, 5600), (9000, 10300)]
    made = {(a, b): synth.make_region(b - a, 6, 5100 + a, B.oracle_swalign, P) for a, b in regions}

    def muts_of(a, b):
        ms = synth.random_point_mutations(np.random.default_rng(a + 7), made[(a, b)][0], 120)
        for m in ms:
            m.start += a
        return ms
    got = psdist.variant_regions(regions, lambda a, b: B.make_pa(PSAlign, made[(a, b)][0], copy.deepcopy(made[(a, b)][1]), P), muts_of)
    for (a, b), g in zip(regions, got):
        want = variant_region(B.make_pa(B.OraclePSAlign, made[(a, b)][0], copy.deepcopy(made[(a, b)][1]), P), muts_of(a, b), region_start=a)
        assert [(x.start, x.orig, x.mut, x.score) for x in g] == [(y.start, y.orig, y.mut, y.score) for y in want]
        assert min(x.start for x in g) >= a
