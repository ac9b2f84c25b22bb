"""Multi-process tests of the region-sharding driver (gloo, world_size 2, CPU).  The per-region work
function here is the oracle-backed PSAlign — the distributed layer itself is backend-agnostic."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest

import backends as B
from poreseq_amd import consensus

ROOT = B.ROOT

WORKER = r'''
import copy, os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import backends as B
from poreseq_amd import dist as psdist, synth
from poreseq_amd.util import DEFAULT_PARAMS
rank, local, world = psdist.init(backend="gloo")
P = dict(DEFAULT_PARAMS, verbose=0)
regions = [(200 + 10 * k, 3, 500 + k) for k in range(5)]      # (length, events, seed)
def process(reg):
    L, E, seed = reg
    draft, events, truth = synth.make_region(L, E, seed, B.oracle_swalign, P)
    pa = B.make_pa(B.OraclePSAlign, draft, events, P)
    sc = pa.ScoreEvents()
    pa.Refine()
    return pa.sequence, np.array(sc)
res = psdist.run_regions(regions, process, max_events=8)
t = psdist.max_over_ranks(rank + 1.5)
psdist.barrier()
if rank == 0:
    print(json.dumps({"world": world, "tmax": t, "seqs": [r[0] for r in res], "scores": [r[1][:3].tolist() for r in res]}))
psdist.finalize()
'''


_PORT_SEQ = 0


def run_world(n):
    code = WORKER % {"root": ROOT}
    if n == 1:
        env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
        out = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=600)
    else:
        global _PORT_SEQ
        _PORT_SEQ += 1
        port = 29500 + (os.getpid() * 7 + _PORT_SEQ * 13) % 2000   # a fresh port per launch (the previous one may sit in TIME_WAIT)
        out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                                       "--master-addr", "127.0.0.1", "--master-port", str(port), "-c", code], timeout=900) \
            if False else _torchrun(code, n, port)
    import json
    line = [l for l in out.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def _torchrun(code, n, port):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
        path = f.name
    try:
        try:
            return subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                                            "--master-addr", "127.0.0.1", "--master-port", str(port), path],
                                           timeout=900, stderr=subprocess.STDOUT)
        except subprocess.CalledProcessError as e:
            print(e.output.decode(errors="replace")[-4000:])
            raise
    finally:
        os.unlink(path)


def test_region_sharding_world2_equals_world1():
    one = run_world(1)
    two = run_world(2)
    assert two["world"] == 2 and two["tmax"] == 2.5 and one["tmax"] == 1.5
    assert one["seqs"] == two["seqs"]                 # identical consensus whatever the sharding
    assert one["scores"] == two["scores"]             # bit-identical gathered log-likelihoods
    assert all(len(s) > 150 for s in one["seqs"])


WORKER_REFINE = r'''
import copy, os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import backends as B
from poreseq_amd import dist as psdist, synth
from poreseq_amd.util import DEFAULT_PARAMS
rank, local, world = psdist.init(backend="gloo")
P = dict(DEFAULT_PARAMS, verbose=0)
P.pop("end_trim")
regions = [(0, 260), (260, 330), (330, 600), (600, 690), (690, 900)]      # deliberately uneven lengths
def make(a, b):
    draft, events, truth = synth.make_region(b - a, 5, 800 + a, B.oracle_swalign, P)
    return B.make_pa(B.OraclePSAlign, draft, events, P)
mine = psdist.shard(regions, rank, world, weights=[b - a for a, b in regions])
res = psdist.refine_regions(regions, make, params=None, batch=2, reps=1)
loads = psdist.max_over_ranks(sum(b - a for _, (a, b) in mine))
if rank == 0:
    print(json.dumps({"world": world, "seqs": [r[0] for r in res], "accs": [r[1] for r in res], "maxload": loads}))
psdist.finalize()
'''


def test_refine_regions_uneven_world2_equals_world1():
    """lock-step refinement + longest-first sharding: same sequences whatever the world size; the heavier rank carries at
    most one region more than an even split"""
    global WORKER
    keep, WORKER = WORKER, WORKER_REFINE
    try:
        one, two = run_world(1), run_world(2)
    finally:
        WORKER = keep
    assert one["seqs"] == two["seqs"] and one["accs"] == two["accs"] and two["world"] == 2
    assert one["maxload"] == 900 and two["maxload"] <= 450 + 270


WORKER_POLISH = r'''
import copy, os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import backends as B
from poreseq_amd import consensus, dist as psdist, synth
from poreseq_amd.util import DEFAULT_PARAMS
rank, local, world = psdist.init(backend="gloo")
P = dict(DEFAULT_PARAMS, verbose=0)
P.pop("end_trim")
rng = np.random.default_rng(44)
truth = synth.random_sequence(rng, 1750)              # six overlapping region work-items, the last ones short (split_fasta.py:94-101 steps by length - 1000)
made = []
def make(a, b):
    made.append((a, b))
    d, events, _ = synth.make_region(b - a, 3, 900 + a, B.oracle_swalign, P, truth=truth[a:b], draft_error=0.01)
    return B.make_pa(B.OraclePSAlign, d, events, P)
seq, parts = consensus.polish(truth, make, params=None, region_length=1300, batch=2, reps=1, swalign=B.oracle_swalign)
loads = [None] * world
import torch.distributed as dist
if dist.is_initialized():
    dist.all_gather_object(loads, made)
else:
    loads = [made]
if rank == 0:
    print(json.dumps({"world": world, "seq": seq, "regions": [[a, b] for a, b, _, _ in parts], "accs": [p[3] for p in parts], "made": loads}))
psdist.finalize()
'''


def test_polish_six_regions_on_eight_ranks_equals_one_rank():
    """BASELINE config #4's shape (6 region work-items, 8 ranks) through the package-level driver split -> refine -> merge: the six
    regions go to six ranks, two ranks refine nothing and only take part in the final gather (the log says which), and every rank
    ends with the same stitched sequence as a single rank computes.  Reference: split_fasta.py:94-101, cmdline.py:182-195, merge_fasta.py:8-39."""
    global WORKER
    keep, WORKER = WORKER, WORKER_POLISH
    try:
        one = run_world(1)
        code = WORKER % {"root": ROOT}
        out = _torchrun(code, 8, 29500 + (os.getpid() * 11 + 977) % 2000).decode()
    finally:
        WORKER = keep
    import json
    eight = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert eight["world"] == 8 and eight["regions"] == [[0, 1300], [300, 1600], [600, 1750], [900, 1750], [1200, 1750], [1500, 1750]]
    assert eight["seq"] == one["seq"] and eight["accs"] == one["accs"]
    assert sorted(len(m) for m in eight["made"]) == [0, 0, 1, 1, 1, 1, 1, 1]              # one region per rank, two ranks idle
    assert sorted(tuple(r) for m in eight["made"] for r in m) == [tuple(r) for r in sorted(eight["regions"])]   # every region built once
    assert "6 regions on 8 ranks: ranks [6, 7] idle" in out


WORKER_SLOTS = r"""
import copy, os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import backends as B
from poreseq_amd import dist as psdist, synth
from poreseq_amd.util import DEFAULT_PARAMS
rank, local, world = psdist.init(backend="gloo")
P = dict(DEFAULT_PARAMS, verbose=0)
P.pop("end_trim")
# nine ragged region work-items (a chromosome's short tail included)
regions = [(0, 210), (210, 330), (330, 520), (520, 600), (600, 830), (830, 900), (900, 1100), (1100, 1150), (1150, 1390)]
def make(a, b):
    draft, events, truth = synth.make_region(b - a, 3, 1300 + a, B.oracle_swalign, P)
    return B.make_pa(B.OraclePSAlign, draft, events, P)
res = psdist.refine_regions(regions, make, params=None, batch=1, reps=1, in_flight=int(os.environ.get("TEST_IN_FLIGHT", "2")))
if rank == 0:
    print(json.dumps({"world": world, "seqs": [r[0] for r in res], "accs": [r[1] for r in res]}), flush=True)
# NO finalize(): a driver script that simply ends behind refine_regions must not take its peers down (the call ends with a
# barrier and the package tears the process group down at exit)
"""


def test_refine_regions_streaming_slots_world4_equals_world1_and_exits_cleanly():
    """Nine ragged regions over four gloo ranks, two batch slots (host threads) per rank: the same sequences as one rank with one
    slot — results do not depend on world size, in_flight or the slot — and every rank exits cleanly WITHOUT calling finalize()
    (cmdline.py:182-195: the reference's region processes end independently; ours must not abort one another)."""
    global WORKER
    keep, WORKER = WORKER, WORKER_SLOTS
    try:
        os.environ["TEST_IN_FLIGHT"] = "1"
        one = run_world(1)
        os.environ["TEST_IN_FLIGHT"] = "2"
        code = WORKER % {"root": ROOT}
        out = _torchrun(code, 4, 29500 + (os.getpid() * 17 + 433) % 2000).decode()      # (raises on any rank's non-zero exit)
    finally:
        WORKER = keep
        os.environ.pop("TEST_IN_FLIGHT", None)
    import json
    four = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert four["world"] == 4 and len(four["seqs"]) == 9
    assert four["seqs"] == one["seqs"] and four["accs"] == one["accs"]
    assert "terminate called" not in out and "SIGABRT" not in out


def test_split_and_merge_regions():
    assert consensus.split_regions(35000, 10000) == [(0, 10000), (9000, 19000), (18000, 28000), (27000, 35000)]
    assert consensus.split_regions(8000, 10000) == [(0, 8000)]
    assert consensus.split_regions(48500, 10000) == [(0, 10000), (9000, 19000), (18000, 28000), (27000, 37000),
                                                     (36000, 46000), (45000, 48500)]
    from poreseq_amd import synth
    rng = np.random.default_rng(3)
    s = synth.random_sequence(rng, 5000)
    a, b = s[:3000], s[2000:]
    assert consensus.merge_seqs(a, b, 1000, swalign=B.oracle_swalign) == s


def test_train_driver_picks_the_most_accurate_parameter_set():
    """cmdline.py:246-267 on the oracle backend: 3 fixed parameter sets, the best one must be returned."""
    import copy
    from poreseq_amd import synth
    from poreseq_amd.util import DEFAULT_PARAMS, VaryParams
    P = dict(DEFAULT_PARAMS, verbose=0)
    draft, events, truth = synth.make_region(220, 6, 91, B.oracle_swalign, P)

    def make_pa(p):
        evs = copy.deepcopy(events)
        for e in evs:
            e.setparams(p)
        return B.make_pa(B.OraclePSAlign, draft, evs, p)

    sets = [dict(P), dict(P, skip_t=0.6, skip_c=0.6, stay_t=0.5, stay_c=0.5), dict(P, insert_t=0.3, insert_c=0.3)]
    B.reset_rand()
    best, accs = consensus.train(make_pa, P, truth, iters=1, reps=2, paramlists=[sets])
    B.reset_rand()
    each = []
    for p in sets:
        each.append(consensus.consensus_region(make_pa(p), p, reps=2, refseq=truth)[1])
    assert best in sets and accs[0] >= max(each) - 1e-9
    vp = VaryParams(P)
    assert len(vp) == 16 and all(set(v) == set(P) for v in vp)
    assert all(sum(v[k] != P[k] for k in P) == 3 for v in vp)      # exactly three transition keys move


def test_run_regions_in_flight_keeps_order_and_reseeds_per_region():
    """Single process, worker threads: results come back in region order and the fresh-random-stream hook runs once
    per region in the worker that refines it (no native library involved)."""
    import threading
    from poreseq_amd import dist as psdist
    seen, lock = [], threading.Lock()

    def fresh():
        with lock:
            seen.append(threading.get_ident())

    def process(reg):
        return "ACGT" * reg, np.arange(3, dtype=np.float64) + reg

    regions = list(range(1, 8))
    out = psdist.run_regions(regions, process, max_events=4, in_flight=3, fresh_rand=fresh)
    assert [r[0] for r in out] == ["ACGT" * k for k in regions]
    assert all(np.array_equal(r[1][:3], np.arange(3) + k) for r, k in zip(out, regions))
    assert len(seen) == len(regions) and 1 <= len(set(seen)) <= 3
    seq = psdist.run_regions(regions, process, max_events=4, in_flight=1, fresh_rand=None)
    assert [r[0] for r in seq] == [r[0] for r in out]


WORKER_GATHER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from poreseq_amd import dist as psdist
rank, local, world = psdist.init(backend="gloo")
weights = %(weights)r
items = list(range(len(weights)))
mine = psdist.shard(items, rank, world, weights=weights)
local = [(i, "ACGT" * (1 + i %% 5) + "T" * i, np.array([float(i), 2.0 * i])) for i, _ in mine]
got = psdist.gather_regions(local, len(items), 4)
share = psdist.max_over_ranks(len(mine))
if rank == 0:
    print(json.dumps({"world": world, "share": share, "seqs": [g[0] for g in got], "sc": [g[1][:2].tolist() for g in got]}), flush=True)
psdist.finalize()
'''


def _gather_world(n, weights):
    global WORKER
    keep, WORKER = WORKER, WORKER_GATHER.replace("%(weights)r", repr(weights))
    try:
        return run_world(n)
    finally:
        WORKER = keep


def test_gather_with_skewed_weighted_shares():
    """a weighted deal may hand one rank more than ceil(n / world) regions ([10, 4, 4, 1] on two ranks: 1 + 3);
    the gather buffers follow the largest share instead of assuming a round-robin split"""
    w = [10, 4, 4, 1]
    two = _gather_world(2, w)
    assert two["world"] == 2 and two["share"] == 3
    assert two["seqs"] == ["ACGT" * (1 + i % 5) + "T" * i for i in range(4)]
    assert two["sc"] == [[float(i), 2.0 * i] for i in range(4)]


def test_world8_ragged_six_regions_and_gather():
    """config #4's shape on eight ranks: six regions, two ranks idle; every rank still takes part in the gather"""
    w = [10000, 10000, 10000, 10000, 10000, 3500]
    got = _gather_world(8, w)
    assert got["world"] == 8 and got["share"] == 1
    assert got["seqs"] == ["ACGT" * (1 + i % 5) + "T" * i for i in range(6)]


def test_shard_plan_config5_on_eight_ranks_is_balanced():
    """config #5: 512 regions (511 of 10 kb + a short tail) dealt to 8 ranks — loads within one region of each other, every
    region assigned exactly once, the same plan on every rank (pure function, no communication)"""
    from poreseq_amd import dist as psdist
    regions = consensus.split_regions(4_600_000, 10000)
    lens = [b - a for a, b in regions]
    seen, loads, counts = [], [], []
    for r in range(8):
        mine = psdist.shard(regions, r, 8, weights=lens)
        seen += [i for i, _ in mine]
        loads.append(sum(b - a for _, (a, b) in mine))
        counts.append(len(mine))
    assert sorted(seen) == list(range(len(regions)))
    assert max(loads) - min(loads) <= max(lens)
    assert max(counts) - min(counts) <= 1


def test_ranks_on_one_node_split_the_host_cores(monkeypatch):
    """dist._pin_host_cores: disjoint, equal slices of the node's cores per local rank and a helper-thread cap per rank"""
    from poreseq_amd import dist as psdist
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity")
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        pytest.skip("one core")
    try:
        monkeypatch.delenv("PORESEQ_HOST_THREADS", raising=False)
        monkeypatch.delenv("PORESEQ_NO_PIN", raising=False)
        a = psdist._pin_host_cores(0, 2)
        os.sched_setaffinity(0, before)
        b = psdist._pin_host_cores(1, 2)
        assert a and b and not set(a) & set(b) and len(a) == len(b) == len(before) // 2
        assert int(os.environ["PORESEQ_HOST_THREADS"]) == max(2, len(a) // 4)
        os.sched_setaffinity(0, before)
        assert psdist._pin_host_cores(0, 1) is None          # one rank per node: nothing to split
    finally:
        os.sched_setaffinity(0, before)
        os.environ.pop("PORESEQ_HOST_THREADS", None)


WORKER_EVENTS = r'''
import copy, os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import backends as B
from poreseq_amd import dist as psdist, synth
from poreseq_amd.util import DEFAULT_PARAMS
rank, local, world = psdist.init(backend="gloo")
P = dict(DEFAULT_PARAMS, verbose=0)
draft, events, truth = synth.make_region(260, 7, 4242, B.oracle_swalign, P)       # 7 events on 2 ranks: 4 + 3, interleaved
events[3].ref_align[:] = 0                                                       # an inert event in the middle (adds exactly 0)
muts = synth.random_point_mutations(np.random.default_rng(9), draft, 150)
muts[5].start = len(draft) + 3                                                    # past the end: skipped by every event
pa = B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(events), P)
got = psdist.score_mutations_event_sharded(pa, muts)
want = B.make_pa(B.OraclePSAlign, draft, copy.deepcopy(events), P).ScoreMutations(muts)
same = [g.score == w.score and (g.start, g.orig, g.mut) == (w.start, w.orig, w.mut) for g, w in zip(got, want)]
ok = psdist.max_over_ranks(0.0 if all(same) and len(got) == len(want) else 1.0)
if rank == 0:
    print(json.dumps({"world": world, "all_ranks_equal_unsharded": ok == 0.0, "n": len(got), "npos": sum(1 for g in got if g.score > 0)}), flush=True)
psdist.finalize()
'''


def test_event_sharded_scoring_equals_unsharded_world1_and_world2():
    """SURVEY 8(e), second axis: the events of one region dealt to the ranks, per-event score terms all-gathered and summed in the
    reference's event order on every rank — bit-identical to the unsharded ScoreMutations (oracle backend, gloo)"""
    global WORKER
    keep, WORKER = WORKER, WORKER_EVENTS
    try:
        one, two = run_world(1), run_world(2)
    finally:
        WORKER = keep
    assert one["all_ranks_equal_unsharded"] and two["all_ranks_equal_unsharded"] and two["world"] == 2
    assert one["n"] == two["n"] == 150 and one["npos"] == two["npos"] > 0


WORKER_VARIANT = r'''
import copy, os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import backends as B
from poreseq_amd import dist as psdist, synth
from poreseq_amd.consensus import variant_region
from poreseq_amd.util import DEFAULT_PARAMS
rank, local, world = psdist.init(backend="gloo")
P = dict(DEFAULT_PARAMS, verbose=0)
NREG = int(os.environ["NREG"])
regions = [(1000 * k, 1000 * k + 220 + 30 * k) for k in range(NREG)]
def make(a, b):
    draft, events, truth = synth.make_region(b - a, 5, 700 + a, B.oracle_swalign, P)
    return B.make_pa(B.OraclePSAlign, draft, events, P)
def muts_of(a, b):
    ms = synth.random_point_mutations(np.random.default_rng(a + 1), make(a, b).sequence, 60)
    for m in ms: m.start += a
    return ms
got = psdist.variant_regions(regions, make, muts_of)
want = [variant_region(make(a, b), muts_of(a, b), region_start=a) for a, b in regions]
same = all(len(g) == len(w) and all((x.start, x.orig, x.mut, x.score) == (y.start, y.orig, y.mut, y.score) for x, y in zip(g, w)) for g, w in zip(got, want))
ok = psdist.max_over_ranks(0.0 if same else 1.0)
if rank == 0:
    print(json.dumps({"world": world, "equal": ok == 0.0, "n": [len(g) for g in got], "first_start": [g[0].start for g in got]}), flush=True)
psdist.finalize()
'''


def test_variant_regions_shards_by_region_or_by_event():
    """`poreseq variant` over the ranks: three regions on two ranks shard by region; ONE region on two ranks shards its events (no rank
    idles) — both equal the single-process scores bit for bit, starts absolute (Variant.py:66-95; oracle backend, gloo)"""
    global WORKER
    keep, WORKER = WORKER, WORKER_VARIANT
    try:
        res = {}
        for nreg in (3, 1):
            os.environ["NREG"] = str(nreg)
            res[nreg] = (run_world(1), run_world(2))
    finally:
        WORKER = keep
        os.environ.pop("NREG", None)
    for nreg, (one, two) in res.items():
        assert one["equal"] and two["equal"] and two["world"] == 2, nreg
        assert one["n"] == two["n"] == [60] * nreg and one["first_start"] == two["first_start"]
    assert res[3][0]["first_start"][2] >= 2000                      # absolute starts


def test_stream_batches_keeps_item_order_and_reraises():
    """poreseq_amd.dist.stream_batches: items stream through `in_flight` host threads (a thread takes the next item when its own is
    done), results come back in item order, the first exception is re-raised; no GPU involved (the workers' entry hint is moot here)."""
    import threading
    import time
    from poreseq_amd import dist as psdist
    seen = []
    lock = threading.Lock()

    def work(k):
        time.sleep(0.02 * ((7 - k) % 3))
        with lock:
            seen.append(threading.get_ident())
        return k * k

    assert psdist.stream_batches(range(7), work, in_flight=3) == [k * k for k in range(7)]
    assert 1 < len(set(seen)) <= 3
    assert psdist.stream_batches(range(3), work, in_flight=1) == [0, 1, 4]
    assert psdist.stream_batches([5], work, in_flight=4) == [25]
    assert psdist.stream_batches(range(2), work, in_flight=8) == [0, 1]          # more slots than items

    def bad(k):
        if k == 2:
            raise ValueError("item 2")
        return k

    with pytest.raises(ValueError, match="item 2"):
        psdist.stream_batches(range(5), bad, in_flight=2)

    # after the first failure nothing new starts: of 40 slow items on two threads only the ones already running finish
    started = []

    def slow_bad(k):
        with lock:
            started.append(k)
        if k == 1:
            raise ValueError("item 1")
        time.sleep(0.05)
        return k

    with pytest.raises(ValueError, match="item 1"):
        psdist.stream_batches(range(40), slow_bad, in_flight=2)
    assert len(started) < 10

    # the entry hook runs once in every worker, before any item
    entered = []
    order = []

    def enter():
        with lock:
            entered.append(threading.get_ident())

    def work2(k):
        with lock:
            order.append(len(entered))
        return k

    assert psdist.stream_batches(range(6), work2, in_flight=3, enter=enter) == list(range(6))
    assert len(entered) == 3 and len(set(entered)) == 3 and min(order) == 3
