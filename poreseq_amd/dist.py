"""Multi-GPU execution: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  The path shards by *region*: regions are independent work-items by
construction (cmdline.py:182-195, split_fasta.py:50-133), so there is no data-path collective; the only
exchange is the final gather of each region's consensus sequence and per-event log-likelihoods.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # ONE mapping of local ranks to devices, shared by the library (PORESEQ_DEVICE, read when it first touches the GPU), the device
    # fraction below and torch's current device: local rank r runs on device r mod (devices of the node)
    os.environ.setdefault("PORESEQ_DEVICE", str(_device_of(local)))
    _pin_host_cores(local, local_world)
    _share_device(local, local_world)
    force = os.environ.get("PORESEQ_FORCE_PG") == "1"     # tests: exercise the collective path on one GPU
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("PORESEQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(_device_of(local))
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        global _ATEXIT
        if not _ATEXIT:
            import atexit
            atexit.register(_leave_quietly)
            _ATEXIT = True
    return rank, local, world


_ATEXIT = False


def _leave_quietly():
    """atexit: a script that ends without finalize() still tears its process group down (a process that exits with the group's
    worker threads alive ends in std::terminate -> SIGABRT).  No barrier here: a peer that has crashed must not hang the others'
    exit; the drivers themselves end with one (_gather_and_meet)."""
    try:
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass


def _ndev():
    try:
        return int(torch.cuda.device_count())        # (counts devices without initialising the GPU)
    except Exception:
        return 0


def _device_of(local):
    """device index of local rank `local`: local mod the node's device count (0 without a GPU)"""
    return int(local) % max(1, _ndev())


def _share_device(local, local_world):
    """More local ranks than GPUs (tests of the N-rank command line on one device; never the production layout): every rank plans its
    device memory — slabs, runtime shares, pool ceiling — for its fraction of the device it lands on (include/poreseq_hip.h,
    ps_set_device_fraction), through the environment because the library reads it at its first compute call."""
    ndev = _ndev()
    if ndev < 1 or local_world <= ndev:
        return 1.0
    mine = sum(1 for r in range(local_world) if _device_of(r) == _device_of(local))
    frac = 1.0 / max(1, mine)
    os.environ.setdefault("PORESEQ_DEVICE_FRACTION", "%.6f" % frac)
    return frac


def _pin_host_cores(local, local_world):
    """Several ranks on one node: each keeps to its own slice of the node's cores (7 driver threads + helper threads per rank
    would otherwise be ~300 runnable threads wandering over 256 cores at 8 ranks) and caps the library's helper threads per call
    at a quarter of the slice.  Done before anything touches the GPU; PORESEQ_NO_PIN=1 leaves affinity alone."""
    if local_world <= 1 or os.environ.get("PORESEQ_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 1:
        return None
    mine = cores[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    os.environ.setdefault("PORESEQ_HOST_THREADS", str(max(2, per // 4)))
    return mine


def device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def shard(items, rank, world, weights=None):
    """(index, item) pairs of this rank.  Without weights: round-robin.  With weights (e.g. region lengths): longest
    first onto the least loaded rank (ties: lowest rank) — every rank computes the same assignment, no communication;
    a ragged tail (the last, short region of a chromosome; config #5's 512 regions over 8 GPUs) then costs at most one
    region's worth of imbalance instead of a whole round."""
    if weights is None:
        return [(i, it) for i, it in enumerate(items) if i % world == rank]
    load = [0.0] * world
    mine = []
    for i in sorted(range(len(items)), key=lambda k: (-float(weights[k]), k)):
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += float(weights[i])
        if r == rank:
            mine.append((i, items[i]))
    return sorted(mine)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def finalize():
    """Barrier + tear-down of the process group (ranks that leave while others still hold gloo / RCCL connections abort)."""
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def max_over_ranks(x):
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_regions(local_results, n_regions, max_events):
    """All ranks receive every region's (sequence, per-event scores).

    local_results: list of (region_index, sequence str, scores float64[<= max_events]).
    Payload: one uint8 tensor (sequences, padded) and one float64 tensor (scores) per rank, exchanged
    with all_gather — RCCL over xGMI under the nccl backend; a few KB per region, latency-bound.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    # rows per rank: the largest share any rank holds (a weighted deal may give one rank more than ceil(n / world) regions)
    per = max(1, int(max_over_ranks(len(local_results))))
    maxlen = max([len(s) for _, s, _ in local_results] + [0])
    maxlen = int(max_over_ranks(maxlen))
    dev = device()
    # built on the host, one copy per tensor to the device (under nccl: three H2D copies per rank, not three per region)
    h_seqs = np.zeros((per, maxlen + 1), dtype=np.uint8)
    h_meta = np.full((per, 2), -1, dtype=np.int64)
    h_scores = np.zeros((per, max_events), dtype=np.float64)
    for k, (idx, s, sc) in enumerate(local_results):
        b = np.frombuffer(s.encode("ascii"), dtype=np.uint8)
        h_seqs[k, :len(b)] = b
        h_meta[k, 0], h_meta[k, 1] = idx, len(b)
        sc = np.asarray(sc, dtype=np.float64)
        h_scores[k, :len(sc)] = sc
    seqs = torch.from_numpy(h_seqs).to(dev)
    meta = torch.from_numpy(h_meta).to(dev)
    scores = torch.from_numpy(h_scores).to(dev)
    if world > 1:
        gs = [torch.empty_like(seqs) for _ in range(world)]
        gm = [torch.empty_like(meta) for _ in range(world)]
        gc = [torch.empty_like(scores) for _ in range(world)]
        dist.all_gather(gs, seqs)
        dist.all_gather(gm, meta)
        dist.all_gather(gc, scores)
    else:
        gs, gm, gc = [seqs], [meta], [scores]
    out = [None] * n_regions
    for r in range(world):
        m = gm[r].cpu().numpy()
        s = gs[r].cpu().numpy()
        c = gc[r].cpu().numpy()
        for k in range(per):
            idx, ln = int(m[k, 0]), int(m[k, 1])
            if idx >= 0:
                out[idx] = (s[k, :ln].tobytes().decode("ascii"), c[k].copy())
    return out


def score_mutations_event_sharded(pa, muts, make_pa=None):
    """`pa.ScoreMutations(muts)` with the EVENTS of the region dealt to the ranks (second axis: fewer regions
    than GPUs — config #4's six regions on eight — or one region's latency): rank r scores events r, r + world, ... (the fills are
    per event: nothing is computed twice), the per-event terms are exchanged with one all_gather (E x M doubles: 6.4 MB for a
    Refine list at 10 events; RCCL over xGMI under nccl), and every rank adds them up in the reference's event order
    (MakeMutations.cpp:51: score = -1e-6, then += delta per event) — bit-identical to the unsharded call, where an all-reduce
    would sum in another order.  `make_pa(events) -> PSAlign` builds this rank's sub-alignment (default: a copy of `pa` with the
    rank's events).  Returns the list of MutationScore on every rank.  The events of `pa` are not re-aligned (the reference's
    Python `ScoreMutations` discards the re-alignment too, pyx:310-345)."""
    import copy as _copy
    from .util import MutationScore
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    E, M = len(pa.events), len(muts)
    mine = list(range(rank, E, world))
    if make_pa is None:
        def make_pa(events):
            sub = type(pa)()
            sub.sequence, sub.events, sub.params = pa.sequence, [_copy.deepcopy(e) for e in events], dict(pa.params)
            return sub
    local = make_pa([pa.events[e] for e in mine]).ScoreMutationDeltas(muts) if mine and M else np.zeros((0, M))
    per = (E + world - 1) // world
    dev = device()
    buf = torch.zeros((per, max(M, 1)), dtype=torch.float64, device=dev)
    if local.size:
        buf[:local.shape[0], :M] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    if world > 1:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [buf]
    parts = [p.cpu().numpy() for p in parts]
    score = np.full(M, -1e-6, dtype=np.float64)
    for e in range(E):                               # the reference's order: event 0, 1, 2, ...
        score = score + parts[e % world][e // world, :M]
    out = []
    for m, sc in zip(muts, score):
        ms = MutationScore()
        ms.start, ms.orig, ms.mut, ms.score = m.start, m.orig, m.mut, float(sc)
        out.append(ms)
    return out


def variant_regions(regions, make_region_pa, muts_of):
    """`poreseq variant` (Variant.py:66-95: one ScoreMutations per region, starts absolute in the files and region-relative inside the call)
    over the ranks of the process group.  regions: [(start, end)]; make_region_pa(start, end) -> PSAlign of the region; muts_of(start, end) ->
    the MutationInfo list of the region with ABSOLUTE starts (not empty).

    With at least as many regions as ranks the path shards by region, like everything else (longest first; one all_gather of the scores).
    With FEWER regions than ranks — BASELINE config #3 / #4's six regions on eight GPUs — no rank idles: every region is scored by ALL ranks,
    its events dealt to them (`score_mutations_event_sharded`: per-event terms all-gathered and added in the reference's event order,
    bit-identical to the unsharded call).  Every rank returns [list of MutationScore] per region, starts absolute."""
    from .consensus import variant_region
    from .util import MutationScore
    import copy as _copy
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lists = [muts_of(a, b) for a, b in regions]
    if any(len(l) == 0 for l in lists):
        raise ValueError("variant_regions: every region needs an explicit edit list (for all point edits call PSAlign.ScorePoints per region)")
    if world > len(regions):
        if rank == 0:
            import sys
            sys.stderr.write("[poreseq_amd.dist] %d regions on %d ranks: variant scoring shards each region's events over all ranks\n" % (len(regions), world))
        out = []
        for (a, b), muts in zip(regions, lists):
            pa = make_region_pa(a, b)
            rel = _copy.deepcopy(muts)
            for m in rel:
                m.start -= a
            sc = score_mutations_event_sharded(pa, rel)
            for m in sc:
                m.start += a
            out.append(sc)
        return out
    mine = shard(regions, rank, world, weights=[b - a for a, b in regions])
    local = []
    for idx, (a, b) in mine:
        sc = variant_region(make_region_pa(a, b), _copy.deepcopy(lists[idx]), region_start=a)
        local.append((idx, "", np.array([m.score for m in sc], dtype=np.float64)))
    got = gather_regions(local, len(regions), max(len(l) for l in lists))
    out = []
    for (a, b), muts, (_, sc) in zip(regions, lists, got):
        res = []
        for m, v in zip(muts, sc[:len(muts)]):
            ms = MutationScore()
            ms.start, ms.orig, ms.mut, ms.score = m.start, m.orig, m.mut, float(v)
            res.append(ms)
        out.append(res)
    return out


def _fresh_region_rand():
    """Every region starts from the random stream of a fresh process — the reference runs one `poreseq consensus`
    process per region file and never seeds rand() (Viterbi.cpp:108) — whichever thread refines it."""
    from . import _capi
    _capi.load_hip().srand(1)


def run_regions(regions, process, max_events=64, in_flight=1, fresh_rand=_fresh_region_rand):
    """Shard `regions` over the ranks, run process(region) -> (sequence, scores) locally, gather.

    `process` is any per-region callable; in_flight > 1 runs that many of this rank's regions concurrently, one host
    thread each (the library gives every thread its own stream and device pools).  Results do not depend on in_flight:
    `fresh_rand` (default: ps_srand(1) on the HIP library) is called in the worker before each region.  For the
    consensus schedule itself prefer `refine_regions`: lock-step batches keep the GPU full from one thread.
    """
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = shard(regions, rank, world)

    def one(item):
        idx, reg = item
        if fresh_rand is not None:
            fresh_rand()
        seq, sc = process(reg)
        return (idx, seq, sc)

    if in_flight > 1 and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(in_flight)) as pool:
            local = list(pool.map(one, mine))
    else:
        local = [one(it) for it in mine]
    return _gather_and_meet(local, len(regions), max_events)


def _gather_and_meet(local, n_regions, max_events):
    """the final gather, then a barrier: every rank holds every result before any rank returns, so a driver script that ends right
    behind its call (cmdline.py:182-195: the reference's region processes exit independently) cannot tear down connections a slower
    peer is still receiving on"""
    got = gather_regions(local, n_regions, max_events)
    barrier()
    return got


def refine_regions(regions, make_region_pa, params=None, batch=16, reps=4, max_events=64, in_flight=1):
    """Consensus for a list of (start, end) regions on all ranks: regions are dealt longest-first to the ranks
    (`shard` with weights), each rank refines its share in lock-step batches of `batch` regions on its GPU
    (poreseq_amd.batch), and every rank receives every region's result.  Returns [(sequence, accuracy)] in region order.

    in_flight > 1 keeps that many lock-step batches on the GPU at once, one host thread (slot) each — the library gives every
    thread its own stream and device pools, and a slot takes the next batch when its own is done, so a rank's regions stream
    through the slots (bench.py: 14 slots of 20 regions on one MI355X).  Every region draws from its own random stream
    (poreseq_amd.batch), so the results do not depend on batch, in_flight or on which slot refines a region."""
    from .consensus import consensus_regions
    from .poreseqcpp import PSAlign
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = shard(regions, rank, world, weights=[b - a for a, b in regions])
    if world > len(regions) and rank == 0:
        # fewer region work-items than ranks (BASELINE config #4: 6 regions of a 48.5 kb genome on 8 GPUs): the path shards by region
        # (cmdline.py:182-195), so the ranks beyond the regions have nothing to refine and only take part in the final gather.  A
        # `variant`-type call (one ScoreMutations per region) can use them: score_mutations_event_sharded deals a region's EVENTS.
        import sys
        busy = sorted({r for r in range(world) if shard(regions, r, world, weights=[b - a for a, b in regions])})
        sys.stderr.write("[poreseq_amd.dist] %d regions on %d ranks: ranks %s idle during the refinement (the schedule shards by region)\n"
                         % (len(regions), world, [r for r in range(world) if r not in busy]))
    hip_backend = False   # (decided on the first region's object: a driver over a CPU checker must not start HIP runtimes)
    step = max(1, int(batch))
    chunks = [mine[k:k + step] for k in range(0, len(mine), step)]

    probe = {}            # the first region's object, built once: it tells the backend and then IS that region (a real loader reads BAM + fast5)
    if mine and in_flight > 1:
        a0, b0 = mine[0][1]
        probe[mine[0][0]] = make_region_pa(a0, b0)
        hip_backend = isinstance(probe[mine[0][0]], PSAlign) and type(probe[mine[0][0]])._native is PSAlign._native

    def one(chunk):
        pas = [probe.pop(idx) if idx in probe else make_region_pa(a, b) for idx, (a, b) in chunk]
        res = consensus_regions(pas, params, reps=reps)
        return [(idx, seq, np.array([acc])) for (idx, _), (seq, acc) in zip(chunk, res)]

    local = [r for part in stream_batches(chunks, one, in_flight, enter=_enter_hip_library if hip_backend else None) for r in part]
    got = _gather_and_meet(local, len(regions), max_events)
    return [(s, float(c[0])) for s, c in got]


def _enter_hip_library():
    """a cheap call into the HIP library: the calling thread takes its runtime (stream + device pools)"""
    from . import _capi
    _capi.load_hip().prof_reset()


def stream_batches(items, work, in_flight=1, enter=None):
    """work(item) for every item on up to `in_flight` host threads; a thread takes the next item when its own is done.
    Results in item order.  On the first exception the items that have not started are cancelled, the running ones finish, and the
    exception is re-raised.

    enter: optional callable run once in every worker thread, all of them together, before the first item starts — refine_regions
    passes a call into the HIP library, so that every worker owns its runtime before the first batch sizes its device pools (the
    device is shared n ways from the start; a runtime that sized its pools while it was alone would hold twice its share).  Drivers
    over another backend leave it out: nothing then touches the GPU on their behalf."""
    items = list(items)
    if in_flight <= 1 or len(items) <= 1:
        return [work(it) for it in items]
    import threading
    from concurrent.futures import ThreadPoolExecutor
    n = min(int(in_flight), len(items))
    pool = ThreadPoolExecutor(max_workers=n)
    try:
        if enter is not None:
            gate = threading.Barrier(n)

            def enter_all(_):
                try:
                    enter()
                finally:
                    gate.wait()
            list(pool.map(enter_all, range(n)))
        futures = [pool.submit(work, it) for it in items]
        out, err = [], None
        for f in futures:
            if err is not None:
                f.cancel()             # (a no-op for the ones already running: they finish below)
                continue
            try:
                out.append(f.result())
            except BaseException as e:   # noqa: BLE001 — re-raised below, after the pool has drained
                err = e
        if err is not None:
            raise err
        return out
    finally:
        pool.shutdown(wait=True, cancel_futures=True)
