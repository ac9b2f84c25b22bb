#!/usr/bin/env python3
"""bench.py — kb of consensus refined per second at 10x coverage (BASELINE.json metric).

One "step" = the full `poreseq consensus` schedule (Mutate('self') then up to 4 x {Mutate('viterbi'), Refine()},
poreseq/Mutate.py:70-85) over one BATCH of R independent synthetic regions, each 10 kb with 10 event streams
(BASELINE.json configs[1]), through the drop-in API and the C ABI.  Regions are the reference's own unit of
parallelism (one process per region file, README.md:48-54).  One GPU refines the R regions of a step as B lock-step
batches (poreseq_amd.batch / the ps_batch_* entry points; one host thread and one stream per batch): every phase of the
schedule is one launch chain over all of a batch's events, with the default HIP environment (no extra hardware queues,
no per-region threads or streams).  Default: 280 regions as 14 batches of 20.
The clock covers what SURVEY.md section 8(d) defines as the metric: marshalling of the events + H2D (RegionBatch.load, the copy a
PSAlign call does, once per region), the whole schedule incl. the host's greedy steps, and D2H of sequences and alignments; the
synthetic data are generated outside it.  `resident` in the JSON line is the same measurement with the load left out of the
clock (what the rules call the kernel-side figure).  With N GPUs every rank refines its own batch per step (weak scaling, no
data-path collective) and the value is the whole-job rate N * R * region_kb * K / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--length L] [--events E] [--regions-per-gpu R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 with
  roofline        the dominant kernel class: HIP events around every launch on the launching stream INSIDE the timed steps
                  (queued, read after each host thread's work): algorithmic bytes per launch / average launch duration
                  against the HBM peak; `traffic` from the committed rocprofv3 PMC measurement of the same command
                  (profiles/, tools/pmc_bench.sh), refused when its launch shape disagrees with the live measurement
  parity_in_run   region 0 of EVERY timed step against the digest the reference's own C++ produced for that region
                  (tests/golden/bench_regions.json, tests/golden/make_golden_bench.py): true / false / null (another workload)
  flat keys       "roofline.valu_busy_frac", "roofline.fp64_frac", "roofline.lane_insts_per_band_cell", "roofline.alone_frac",
                  "cpu_baseline.*", "single_region_10kb_s", ...: scalar duplicates of the nested figures (a record that keeps scalars only)
  north_star_1kb  1 kb / 10x: one region alone, a lock-step batch, and the reference C++ at the same size
  cpu_baseline    the reference's C++ (oracle/_ref; the oracle restatement when that is not built) on the host cores:
                  one thread, one process per core, and a same-size (10 kb) extrapolation from measured unit costs
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_OF = {"fill": "k_fill", "sweep": "k_sweep", "score": "k_score", "sw": "k_sw_fill", "viterbi": "k_vit_steps"}   # "sweep" = the strip-sweep class: k_sweeps(_w) (kept columns), k_sweep(_w) (forward-only), k_sweep2(_w) (full records); _w = two / four wavefronts per sweep
SIMDS, FP64_LANES_PER_CLK, SHADER_GHZ = 1024, 16, 2.4          # MI355X: 256 CUs x 4 SIMDs, 16 FP64 lanes per SIMD and clock (78.6 TFLOP/s with FMA)
FP64_OPS_PER_CELL = 45                                          # the reference's arithmetic per DP cell: emission 29 (AlignUtil.h:34-53), recurrence 16 (Alignment.cpp:196-267)


def _cpu_region_worker(job):
    """one full consensus schedule on the CPU checker (own process: the reference is single-threaded)"""
    length, events, seed, use_ref = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import backends as B
    from poreseq_amd import synth
    from poreseq_amd.consensus import consensus_region
    from poreseq_amd.util import DEFAULT_PARAMS
    params = dict(DEFAULT_PARAMS, verbose=0)
    cls = B.RefPSAlign if use_ref else B.OraclePSAlign
    sw = B.ref_swalign if use_ref else B.oracle_swalign
    d, ev, _ = synth.make_region(length, events, seed, sw, params)
    pa = B.make_pa(cls, d, ev, params)
    B.reset_rand()
    t = time.perf_counter()
    consensus_region(pa, params)
    return time.perf_counter() - t


def _cpu_baseline_children(args):
    """(1) one thread, full schedule, one region of --cpu-length bases; (2) one single-threaded process per core"""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import backends as B
    use_ref = B.have_ref()
    if not use_ref and not os.path.exists(B.ORACLE_SO):
        return None
    L, E = args.cpu_length, args.events
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        ct = pool.map(_cpu_region_worker, [(L, E, 1002, use_ref)])[0]
        t1k = ct if L == 1000 else pool.map(_cpu_region_worker, [(1000, E, 5000, use_ref)])[0]
    cpu = {"value": (L / 1000.0) / ct, "unit": "kb/s", "cores": 1, "kind": "reference" if use_ref else "port",
           "sample": "full consensus schedule on one %d-base region, %d events, single thread (%.1f s)" % (L, E, ct),
           "host_cores_available": os.cpu_count()}
    nproc = max(1, min(os.cpu_count() or 1, 64))
    try:
        with ctx.Pool(nproc) as pool:
            pool.map(_cpu_region_worker, [(200, 5, 1, use_ref)] * nproc)      # start-up (imports) outside the clock
            t = time.perf_counter()
            pool.map(_cpu_region_worker, [(L, E, 3000 + k, use_ref) for k in range(nproc)], chunksize=1)
            wall = time.perf_counter() - t
        cpu["one_process_per_core"] = {"processes": nproc, "value": nproc * (L / 1000.0) / wall, "unit": "kb/s",
                                       "sample": "%d regions of %d bases, one single-threaded process each, %.1f s wall" % (nproc, L, wall)}
    except Exception as e:   # pragma: no cover
        cpu["one_process_per_core"] = {"error": str(e)}
    return {"cpu": cpu, "t1k": t1k, "use_ref": use_ref}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--length", type=int, default=10000)
    ap.add_argument("--events", type=int, default=10)
    ap.add_argument("--regions-per-gpu", type=int, default=280, help="independent regions refined on one GPU per step")
    ap.add_argument("--batches-in-flight", type=int, default=14,
                    help="lock-step batches per GPU (one host thread each): while one batch is in a thin phase or on the host, "
                         "the others keep the GPU full; the regions of a step are dealt round-robin to the batches")
    ap.add_argument("--stream", action="store_true",
                    help="let the timed steps' batches stream through the slots (default: every step finishes before the next one's batches "
                         "start; measured the same on one MI355X: 141.1 against 140.5 kb/s)")
    ap.add_argument("--cpu-length", type=int, default=1000, help="region length of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip everything that touches the CPU checkers (oracle / reference)")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-region, 1 kb and profiled passes (profiling runs)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_env:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch with python -m torch.distributed.run --nproc-per-node %d ...\n"
                         % (args.gpus, world_env, args.gpus))
        sys.exit(2)

    # The CPU baseline runs in child processes (the reference is single-threaded; its only parallel mode is one process per
    # region, README.md:48-54).  They are started HERE, before anything in this process touches the GPU.
    cpu_pre = None
    if world_env == 1 and not args.no_cpu:
        cpu_pre = _cpu_baseline_children(args)

    from poreseq_amd import dist as psdist
    rank, local, world = psdist.init()
    import torch
    from poreseq_amd import _capi, synth
    from poreseq_amd.batch import RegionBatch
    from poreseq_amd.consensus import consensus_region, consensus_regions
    from poreseq_amd.poreseqcpp import PSAlign, swalign
    from poreseq_amd.util import DEFAULT_PARAMS

    params = dict(DEFAULT_PARAMS, verbose=0)
    api = _capi.load_hip()
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))   # (ranks that outnumber the devices share them: tests)

    def make(seed, length=None):
        return synth.make_region(length or args.length, args.events, seed, swalign, params)

    def as_pa(region):
        pa = PSAlign()
        pa.sequence, pa.events, pa.params = region[0], copy.deepcopy(region[1]), dict(params)
        return pa

    live_prof = []      # per timed run: {class: [ms, launches, bytes, units]} summed over the slots
    batch_done_s = []   # per timed run: when each lock-step batch finished, seconds after the clock started
    load_s = []         # per timed run: seconds the slots spent marshalling + copying events to the device (inside the clock), / slots

    # Batch slots: NB host threads that live for the whole run (each owns a library runtime: stream + device pools).  A run hands
    # them the lock-step batches of one or more steps through a queue; a slot takes the next batch when its own is done — the
    # production shape (poreseq_amd.dist.refine_regions(in_flight=NB)): regions stream through the slots, there is no barrier
    # between the batches of consecutive steps with --stream (the default finishes every step first).
    import queue
    import threading
    NB = max(1, args.batches_in_flight)
    jobs = queue.Queue()
    errs = []

    def slot_main():
        while True:
            fn = jobs.get()
            try:
                if fn is None:
                    return
                fn()
            except Exception as e:   # pragma: no cover
                errs.append(e)
            finally:
                jobs.task_done()

    slots = []

    def start_slots(n):
        """(the single-region passes run with ONE slot alive: a thread alone inside the library gets the whole device and a second
        stream for its Smith-Waterman batches; the other slots start before the warm-up steps)"""
        while len(slots) < n:
            slots.append(threading.Thread(target=slot_main, daemon=True))
            slots[-1].start()

    def on_every_slot(fn):
        """fn() once in each slot thread (per-thread profiling state lives in the library's thread-local runtime)"""
        gate = threading.Barrier(len(slots))
        for _ in range(len(slots)):
            jobs.put(lambda: (gate.wait(), fn()))
        jobs.join()

    start_slots(1)

    def in_slot(fn):
        """fn() in one of the slot threads: the main thread never enters the library, so the device is shared by NB runtimes"""
        box = []
        jobs.put(lambda: box.append(fn()))
        jobs.join()
        if errs:
            raise errs[0]
        return box[0]

    def run_steps(step_regs, timed=False, nb=NB):
        """the batches of the given steps (each step's regions dealt round-robin to `nb` lock-step batches) through the slots;
        the clock (when timed) starts BEFORE any batch's events are marshalled and copied to the device.
        Returns (seconds, [per step: results in region order])."""
        items = []
        for si, regs in enumerate(step_regs):
            n = max(1, min(nb, len(regs)))
            for k in range(n):
                pas = [as_pa(r) for r in regs[k::n]]
                items.append({"step": si, "k": k, "n": n, "pas": pas, "rb": RegionBatch(pas), "out": None, "done": 0.0, "load": 0.0})
        profs = []
        plock = threading.Lock()
        if timed:
            on_every_slot(lambda: (api.prof_reset(), api.prof_enable(2)))   # HIP events around every hot-kernel launch, read after the work
            psdist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        t0 = time.perf_counter()

        def work(it):
            t = time.perf_counter()
            it["rb"].load()          # marshalling + H2D of the batch's events
            it["load"] = time.perf_counter() - t
            it["out"] = consensus_regions(it["pas"], params, batch=it["rb"])
            it["done"] = time.perf_counter() - t0

        for it in items:
            jobs.put(lambda it=it: work(it))
        jobs.join()
        if timed and torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if errs:
            raise errs[0]
        if timed:
            def collect():
                p = {c: list(api.prof_get(c)) + [api.prof_units(c)] for c in KERNEL_OF}
                p["_forms"] = [api.prof_get(c)[1] for c in ("sweep_w2", "sweep_w4", "sweep_kept")]   # host-side launch counts by form
                api.prof_enable(0)
                with plock:
                    profs.append(p)
            on_every_slot(collect)
            batch_done_s.append(sorted(round(it["done"], 3) for it in items))
            load_s.append(sum(it["load"] for it in items) / NB)
            live_prof.append({c: [sum(p[c][i] for p in profs) for i in range(4)] for c in KERNEL_OF})
            live_prof[-1]["_forms"] = [sum(p["_forms"][i] for p in profs) for i in range(3)]
        outs = []
        for si, regs in enumerate(step_regs):
            out = [None] * len(regs)
            for it in items:
                if it["step"] == si:
                    out[it["k"]::it["n"]] = it["out"]
            outs.append(out)
        return dt, outs

    def run_batch(regs, nb=1):
        dt, outs = run_steps([regs], nb=nb)
        return dt, outs[0]

    # synthetic inputs for every step of this rank, generated outside the timed region
    R = max(1, args.regions_per_gpu)
    nsteps = args.warmup + args.steps
    # (at most three distinct sets: generating 25 x 64 regions would take longer than refining them; later steps go round the sets —
    #  every step recomputes everything, nothing is cached between steps)
    nsets = min(nsteps, 3)
    sets = in_slot(lambda: [[make(1002 + 100000 * rank + 1000 * k + s) for k in range(R)] for s in range(nsets)])
    regions = [sets[s % nsets] for s in range(nsteps)]

    pre = {}
    if rank == 0 and not args.no_extras:
        # ---- one region alone (latency) ----
        run_batch(regions[0][:1])                      # warm: device pools, code objects
        lone = sorted(run_batch(regions[0][k:k + 1])[0] for k in range(min(3, R)))   # schedules differ in length from region to region
        pre["single_region_s"] = lone[len(lone) // 2]
        pre["single_region_s_all"] = lone
        # ---- north star comparison point: 1 kb / 10x ----
        k1 = in_slot(lambda: [make(5000 + k, 1000) for k in range(64)])
        run_batch(k1[:1])
        t1 = min(run_batch(k1[:1])[0] for _ in range(3))
        run_batch(k1)
        tb = run_batch(k1)[0]
        pre["north_star_1kb"] = {"region_bases": 1000, "events": args.events, "single_region_s": t1,
                                 "single_region_kb_s": 1.0 / t1, "lock_step_regions": len(k1),
                                 "lock_step_kb_s": len(k1) * 1.0 / tb}

        # ---- BASELINE config #3's unit of work (poreseq variant, Variant.py:66-95: one ScoreMutations per region): one 10 kb region of
        #      the 48 kb reference, 30 events, its share of the 10 000 point edits (1:3:4 del / sub / ins at uniform positions),
        #      scoring_width 100 — the dense edit-scoring kernel's own figure ----
        def variant3():
            import numpy as np
            from poreseq_amd.consensus import variant_region
            vp = dict(params, scoring_width=100.0)
            d, ev, tr = synth.make_region(10000, 30, 3003, swalign, vp, draft_error=0.0)
            muts = synth.random_point_mutations(np.random.default_rng(3), d, 10000 // 6)
            pa = PSAlign(); pa.sequence, pa.events, pa.params = d, ev, dict(vp)
            variant_region(pa, copy.deepcopy(muts))                   # warm: pools, code objects
            api.prof_reset(); api.prof_enable(1)
            t = time.perf_counter(); variant_region(pa, copy.deepcopy(muts)); wall = time.perf_counter() - t
            api.prof_enable(0)
            ms, n, nb = api.prof_get("score")
            items = api.prof_units("score")
            return {"region_bases": 10000, "events": len(ev), "edits": len(muts), "scoring_width": 100, "call_s": wall, "items": items,
                    "items_per_s_call": items / wall, "k_score_ms": ms, "k_score_launches": int(n), "items_per_s_kernel": items / (ms / 1e3) if ms > 0 else None,
                    "alg_bytes_per_item": nb / items if items else None, "achieved_gbs": (nb / 1e9) / (ms / 1e3) if ms > 0 else None,
                    "frac_of_hbm_peak": (nb / 1e9) / (ms / 1e3) / HBM_PEAK_GBS if ms > 0 else None,
                    "note": "k_score (+ k_old): SURVEY 8(d) bytes per (event, edit) item, 16(Bs+1) + 16 Br + 24(Bs+c) + 32 Br/k + 8 with Bs = 201, Br = 601 "
                            "(37.1 KB for edits at distinct positions); call_s also holds the forward + backward fills and the backtrace of the 30 events"}
        pre["variant_config3"] = in_slot(variant3)

    start_slots(NB)
    on_every_slot(api.prof_reset)     # every slot owns its runtime before the first batch sizes its pools: the device is shared NB ways from the start
    run_steps([regions[s] for s in range(args.warmup)])
    timed_regs = [regions[s] for s in range(args.warmup, nsteps)]
    if os.environ.get("PORESEQ_TRACE"):
        sys.stderr.write("=== MEASURED RUN ===\n")       # tools/tracesum.py sums the host phases after this line
    if not args.stream:
        # every step starts at a barrier: the job's time for a step is the slowest rank's, and the run's the sum of those
        # (max over ranks of the per-rank sums would be optimistic: the slowest rank may differ from step to step)
        dt, dt_res, last = 0.0, 0.0, None
        first_of_step = []              # region 0 of every timed step: (sequence, accuracy)
        for regs in timed_regs:
            t, outs = run_steps([regs], timed=True)
            dt += psdist.max_over_ranks(t)
            dt_res += psdist.max_over_ranks(t - load_s[-1])
            last = outs[-1]
            first_of_step.append(outs[-1][0])
        psdist.barrier()
    else:
        dt, outs = run_steps(timed_regs, timed=True)
        last = outs[-1]
        first_of_step = [o[0] for o in outs]
        psdist.barrier()
        dt_res = psdist.max_over_ranks(dt - sum(load_s))
        dt = psdist.max_over_ranks(dt)
    kb = args.length / 1000.0
    value = world * R * kb * args.steps / dt

    out = {
        "metric": "kb consensus refined/sec at 10x coverage", "value": value, "unit": "kb/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / max(args.steps, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "poreseq consensus, %d kb region, %dx synthetic coverage (BASELINE configs[1]), full Mutate.py "
                               "schedule per region; a step is %d independent regions per GPU = %d lock-step batches; the batches of the timed "
                               "steps %s through %d batch slots (host threads) per GPU; marshalling + H2D of the events, host greedy "
                               "steps and D2H inside the clock"
                               % (args.length // 1000, args.events, R, NB,
                                  "run step by step (a step's batches all finish before the next step's start)" if not args.stream
                                  else "stream (a slot takes the next batch when its own is done; no barrier between steps)", NB),
                   "region_bases": args.length, "events": args.events, "regions_per_gpu": R, "batches_in_flight": NB, "step_barrier": not args.stream,
                   "not_timed_between_steps": "construction of the next step's PSAlign objects from the pre-generated synthetic regions (a deep copy of "
                                              "the event arrays) and reading the queued HIP-event pairs of the finished step",
                   "parallelism": "%d regions x %d GPU(s), %d host thread(s) per GPU, no data-path collective" % (R, world, NB)},
    }

    out["resident"] = {"value": world * R * kb * args.steps / dt_res, "unit": "kb/s", "ms_per_step": 1000.0 * dt_res / max(args.steps, 1),
                       "load_ms_per_step": 1000.0 * sum(load_s) / max(args.steps, 1),
                       "note": "estimate: the clock minus the slots' average marshalling + H2D time (loads overlap other slots' work; not the metric's definition)"}

    out["batch_done_s"] = batch_done_s
    out["library"] = api.info()   # stream / hardware-queue mode, runtimes, memory plan (ps_info)
    if torch.cuda.is_available():   # the slots' pools only grow: what is free now is the headroom the steps ran with
        fr, tt = torch.cuda.mem_get_info()
        out["device_memory"] = {"total_gb": tt / 1e9, "free_gb_after_timed_steps": fr / 1e9}
    if rank == 0:
        a0 = in_slot(lambda: swalign(regions[-1][0][0], regions[-1][0][2])[0])
        a1 = in_slot(lambda: swalign(last[0][0], regions[-1][0][2])[0])
        out["accuracy"] = {"draft_percent": a0, "consensus_percent": a1}
        out.update(pre)
        # ---- parity inside the run: region 0 of EVERY timed step against the digest the reference's own C++ produced for that region
        #      (tests/golden/bench_regions.json, made by tests/golden/make_golden_bench.py: full schedule, fresh rand() stream) ----
        try:
            import hashlib
            with open(os.path.join(ROOT, "tests", "golden", "bench_regions.json")) as fh:
                gold = {(g["length"], g["events"], g["seed"]): g for g in json.load(fh)["regions"]}
            checked, bad = 0, []
            for k, (seq, acc) in enumerate(first_of_step):
                seed = 1002 + (args.warmup + k) % nsets          # region 0 of the step's set on rank 0 (see `sets` above)
                g = gold.get((args.length, args.events, seed))
                if g is None:
                    continue
                checked += 1
                if hashlib.sha256(seq.encode("ascii")).hexdigest() != g["sequence_sha256"] or len(seq) != g["sequence_len"]:
                    bad.append(k)
            out["parity_in_run"] = (checked > 0 and not bad) if checked else None
            out["parity_in_run_detail"] = {"steps_checked": checked, "steps_timed": len(first_of_step), "mismatching_steps": bad,
                                           "what": "SHA-256 of region 0's consensus sequence after the full schedule, every timed step, against "
                                                   "the reference C++ (oracle/_ref) run on the same region: tests/golden/bench_regions.json"}
            if bad:
                sys.stderr.write("bench.py: PARITY FAILURE in timed steps %s\n" % bad)
        except (OSError, ValueError, KeyError) as e:
            out["parity_in_run"] = None
            out["parity_in_run_detail"] = {"error": str(e)}

        sched = None
        if live_prof:
            # ---- roofline of the dominant kernel class: HIP events around every launch of the class on the launching stream,
            #      inside the timed steps (all host threads of this rank; queued, read after each thread's work) ----
            tot = {c: [sum(lp[c][i] for lp in live_prof) for i in range(4)] for c in KERNEL_OF}
            dom = max(tot, key=lambda c: tot[c][0])
            ms, launches, nbytes, units = tot[dom]
            achieved = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
            kname, kclass = KERNEL_OF[dom], None
            if dom == "sweep":
                # the class's dominant kernel: kept-column sweeps of ScoreMutations (k_sweeps) or forward-only sweeps (k_sweep), one wavefront
                # per sweep or two / four (_w); by launch counts the library keeps per form
                w2, w4, kept = (sum(lp["_forms"][i] for lp in live_prof) for i in range(3))
                kname = ("k_sweeps" if 2 * kept >= launches else "k_sweep") + ("_w" if 2 * (w2 + w4) >= launches else "")
                kclass = {"name": "strip sweeps: k_sweeps(_w) kept columns, k_sweep(_w) forward-only", "launches": int(launches), "kept_column_launches": int(kept),
                          "two_wavefronts_per_sweep": int(w2), "four_wavefronts_per_sweep": int(w4), "one_wavefront_per_sweep": int(launches - w2 - w4)}
            roof = {"bound": "hbm", "kernel": kname, "class": kclass, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launches": int(launches),
                    "avg_launch_ms": ms / max(launches, 1), "alg_bytes_per_launch": nbytes / max(launches, 1),
                    "sweeps_per_launch": units / max(launches, 1) if dom in ("fill", "sweep") else None,
                    "measured": "HIP events per launch inside the %d timed step(s), %d lock-step batches in flight (one stream each): a "
                                "launch's duration includes what it shares the chip with" % (args.steps, NB),
                    "all_kernel_classes_ms_per_step": {c: v[0] / max(args.steps, 1) for c, v in tot.items()}}
            # aggregate rate of the class over the wall time of the steps (launches of different batches overlap)
            roof["aggregate_alg_gbs"] = (nbytes / 1e9) / dt if dt > 0 else None
            roof["aggregate_frac"] = roof["aggregate_alg_gbs"] / HBM_PEAK_GBS if dt > 0 else None
            fills_b = tot["fill"][2] + tot["sweep"][2]
            roof["fill_classes_aggregate"] = {"alg_gbs": (fills_b / 1e9) / dt if dt > 0 else None, "frac": (fills_b / 1e9) / dt / HBM_PEAK_GBS if dt > 0 else None,
                                              "note": "both DP-fill classes (k_fill + k_sweep*) by the SURVEY 8(d) accounting over the wall time of the timed steps"}
            roof["launches_in_flight_mean"] = (ms / 1e3) / dt if dt > 0 else None
            # both DP-fill kernels by the same accounting (18 B per forward cell, 16 per backward cell): k_fill = a workgroup
            # per sweep, score matrices written (Alignment::update batches and small forward batches); k_sweep = a wavefront per sweep,
            # one byte per cell written (forward-only batches of 400 alignments and more)
            roof["fill_kernels"] = {}
            for c in ("fill", "sweep"):
                cms, cl, cb, cu = tot[c]
                if cms > 0:
                    roof["fill_kernels"][KERNEL_OF[c]] = {"launches": int(cl), "avg_launch_ms": cms / max(cl, 1), "sweeps_per_launch": cu / max(cl, 1),
                                                          "alg_bytes_per_launch": cb / max(cl, 1), "achieved": (cb / 1e9) / (cms / 1e3),
                                                          "frac": (cb / 1e9) / (cms / 1e3) / HBM_PEAK_GBS, "sweeps_per_step": cu / max(args.steps, 1)}
            if not args.no_extras:
                # the same kernel class with the chip to itself: one lock-step batch of the timed size, alone, event pair read after
                # every launch (the kernel's own quality; the figures above include what a launch shares the chip with)
                def alone():
                    pas = [as_pa(r) for r in regions[-1][:max(1, R // NB)]]
                    api.prof_reset()
                    api.prof_enable(1)
                    consensus_regions(pas, params)
                    api.prof_enable(0)
                    return api.prof_get(dom)
                ims, il, ib = in_slot(alone)
                if ims > 0:
                    roof["one_batch_alone"] = {"avg_launch_ms": ims / max(il, 1), "alg_bytes_per_launch": ib / max(il, 1),
                                               "achieved": (ib / 1e9) / (ims / 1e3), "frac": (ib / 1e9) / (ims / 1e3) / HBM_PEAK_GBS,
                                               "launches": int(il)}
            # HBM-side bytes per launch and the vector-issue counters cannot be collected from inside this process: rocprofv3 --pmc on
            # this very command (tools/pmc_round.sh) writes profiles/rNN_traffic.json and profiles/rNN_valu.json; the newest of each is
            # used, and only if it was measured on this launch shape
            import glob
            shape_ok = lambda tj: (tj.get("length") == args.length and tj.get("events") == args.events and tj.get("regions_per_gpu") == R
                                   and tj.get("batches_in_flight") == NB)
            try:
                path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
                rel = os.path.relpath(path, ROOT)
                with open(path) as fh:
                    tj = json.load(fh)
                tk = tj["kernels"].get("class:" + KERNEL_OF[dom]) or tj["kernels"].get(KERNEL_OF[dom])
                if tk and shape_ok(tj) and abs(tk.get("alg_bytes_per_launch", 0) / roof["alg_bytes_per_launch"] - 1.0) < 0.10:
                    roof["traffic"] = tk["fetch_bytes_per_launch"] + tk["write_bytes_per_launch"]
                    roof["traffic_source"] = rel + ": " + tj.get("source", "")
                    roof["schedule_traffic_gb_per_region"] = tj.get("whole_schedule", {}).get("gb_per_region")
                else:
                    roof["traffic_source"] = rel + " ignored: measured on a different launch shape"
            except (OSError, ValueError, KeyError, ZeroDivisionError, IndexError):
                pass
            # what actually bounds these kernels is FP64 vector issue (max-plus recurrences, ~45 FP64 operations per cell, 4 cycles per
            # wave-instruction), not HBM: the share of the step the chip's vector pipes were busy, and the instructions a sweep issues
            try:
                path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu.json")))[-1]
                rel = os.path.relpath(path, ROOT)
                with open(path) as fh:
                    vj = json.load(fh)
                if shape_ok(vj):
                    sk = vj.get("sweep_kernels", {})
                    roof["valu"] = {"busy_frac": vj["valu_busy_simd_seconds_at_2p4ghz"] / (dt / max(args.steps, 1)),
                                    "busy_simd_seconds_per_step": vj["valu_busy_simd_seconds_at_2p4ghz"],
                                    "valu_per_cell": sk.get("insts_per_lane_cell"), "lane_insts_per_band_cell": sk.get("lane_insts_per_band_cell"),
                                    "wave_insts_per_sweep": sk.get("wave_insts_per_sweep"), "fp64_ops_per_cell": sk.get("fp64_ops_per_cell_reference_arithmetic"),
                                    "source": rel + " (PMC pass of this command; busy_frac = its vector-busy SIMD-seconds over THIS run's step time)"}
                else:
                    roof["valu"] = {"source": rel + " ignored: measured on a different launch shape"}
            except (OSError, ValueError, KeyError, ZeroDivisionError, IndexError):
                pass
            # what the DP fills are short of is FP64 vector issue, not bytes: useful FP64 operations of the reference's arithmetic (45 per band
            # cell, every sweep of both fill classes) over the wall time, against the chip's non-FMA FP64 issue peak
            band = min(2.0 * float(params.get("realign_width", 300)) + 1.0, 0.95 * args.length)
            cells = (tot["fill"][3] + tot["sweep"][3]) * (args.length - 4) * band
            peak_ops = SIMDS * FP64_LANES_PER_CLK * SHADER_GHZ * 1e9
            roof["fp64"] = {"useful_ops_per_s": cells * FP64_OPS_PER_CELL / dt if dt > 0 else None, "peak_ops_per_s": peak_ops,
                            "frac": cells * FP64_OPS_PER_CELL / dt / peak_ops if dt > 0 else None, "band_cells_per_step": cells / max(args.steps, 1),
                            "note": "DP fills only (edit scoring, Smith-Waterman and Viterbi do other work on the same pipes); peak = %d SIMDs x %d FP64 lanes "
                                    "x %.1f GHz, one operation per lane and clock (no FMA: the reference's arithmetic is not fused)" % (SIMDS, FP64_LANES_PER_CLK, SHADER_GHZ)}
            out["roofline"] = roof
            sched = {"fill_sweeps": (tot["fill"][3] + tot["sweep"][3]) / max(args.steps, 1), "score_items": tot["score"][3] / max(args.steps, 1)}

        # ---- parity spot-check + CPU baseline (oracle / reference: checker and baseline only, never the thing measured) ----
        if cpu_pre is not None:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import backends as B
            B.build_oracle = lambda: None     # this process holds the GPU: never start a build from here (the children above did, if needed)
            use_ref = cpu_pre["use_ref"]
            cls = B.RefPSAlign if use_ref else B.OraclePSAlign
            cpu_sw = B.ref_swalign if use_ref else B.oracle_swalign
            d, ev, tr = synth.make_region(400, 6, 77, cpu_sw, params)
            g = in_slot(lambda: B.make_pa(PSAlign, d, copy.deepcopy(ev), params).ScoreEvents())
            c = B.make_pa(cls, d, copy.deepcopy(ev), params).ScoreEvents()
            gp = in_slot(lambda: B.make_pa(PSAlign, d, copy.deepcopy(ev), params).ScorePoints())
            cp = B.make_pa(cls, d, copy.deepcopy(ev), params).ScorePoints()
            out["logl_max_rel_err_vs_cpu"] = max([abs(x - y) / max(abs(y), 1e-300) for x, y in zip(g, c)] +
                                                 [abs(x.score - y.score) / abs(y.score) for x, y in zip(gp, cp) if abs(y.score) > 1e-3])
            cpu = cpu_pre["cpu"]
            # the SAME size as the GPU workload: measured unit costs at this size, times the unit counts of the GPU's own schedule
            if not args.no_extras and sched:
                import numpy as np
                d, ev, tr = regions[-1][0]
                pa = B.make_pa(cls, d, copy.deepcopy(ev), params)
                t = time.perf_counter(); pa.ScoreEvents(); t_fill = (time.perf_counter() - t) / len(ev)
                muts = synth.random_point_mutations(np.random.default_rng(5), d, 3000)
                pw = dict(params, scoring_width=params["point_width"])
                t = time.perf_counter(); B.make_pa(cls, d, copy.deepcopy(ev), pw).ScoreMutations(muts)
                t_item = max(time.perf_counter() - t - 2 * len(ev) * t_fill, 0.0) / (len(muts) * len(ev))
                est = (sched["fill_sweeps"] * t_fill + sched["score_items"] * t_item) / R
                cpu["same_size"] = {"region_bases": args.length, "seconds_per_sweep": t_fill, "seconds_per_scored_item": t_item,
                                    "sweeps_per_region": sched["fill_sweeps"] / R, "scored_items_per_region": sched["score_items"] / R,
                                    "extrapolated_seconds_per_region": est, "value": kb / est, "unit": "kb/s",
                                    "note": "banded fills and edit scoring only (Smith-Waterman, Viterbi and list handling left out: a lower "
                                            "bound on the reference's time); unit costs measured on one core at this size, unit counts "
                                            "taken from the GPU run's own schedule"}
                if "north_star_1kb" in out:
                    ns = out["north_star_1kb"]
                    ns["cpu_reference_s"] = cpu_pre["t1k"]
                    ns["speedup_single_region"] = cpu_pre["t1k"] / ns["single_region_s"]
                    ns["speedup_lock_step"] = ns["lock_step_kb_s"] * cpu_pre["t1k"]
            try:   # one MEASURED run of the reference's full schedule at this size (tests/golden/make_golden_large.py, build container, one core)
                import numpy as np
                z = np.load(os.path.join(ROOT, "tests", "golden", "consensus_L10000_E10.npz"), allow_pickle=False)
                if int(z["L"]) == args.length and int(z["E"]) == args.events:
                    rs = float(z["reference_seconds"])
                    cpu["measured_full_schedule_same_size"] = {
                        "seconds": rs, "value": kb / rs, "unit": "kb/s", "cores": 1,
                        "where": "build container (8-core host), the reference's Cython PSAlign, one region of %d bases x %d events; "
                                 "the fixture the -m gpu test test_hip_consensus_schedule_golden[consensus_L10000_E10] replays" % (args.length, args.events)}
            except (OSError, KeyError, ValueError):
                pass
            out["cpu_baseline"] = cpu
        # ---- flat scalar duplicates of what bounds the path: the driver's record keeps scalar fields only ----
        flat = {}
        r = out.get("roofline") or {}
        v = r.get("valu") or {}
        flat["roofline.valu_busy_frac"] = v.get("busy_frac")
        flat["roofline.lane_insts_per_band_cell"] = v.get("lane_insts_per_band_cell")
        flat["roofline.fp64_frac"] = (r.get("fp64") or {}).get("frac")
        flat["roofline.alone_frac"] = (r.get("one_batch_alone") or {}).get("frac")
        flat["roofline.aggregate_frac"] = r.get("aggregate_frac")
        flat["roofline.fill_classes_aggregate_frac"] = (r.get("fill_classes_aggregate") or {}).get("frac")
        c = out.get("cpu_baseline") or {}
        ns = out.get("north_star_1kb") or {}
        flat["cpu_baseline.same_size_measured_kb_s"] = (c.get("measured_full_schedule_same_size") or {}).get("value")
        flat["cpu_baseline.same_size_extrapolated_kb_s"] = (c.get("same_size") or {}).get("value")
        flat["cpu_baseline.per_core_64proc_kb_s"] = (c.get("one_process_per_core") or {}).get("value")
        flat["cpu_baseline.speedup_1kb_single_region"] = ns.get("speedup_single_region")
        flat["cpu_baseline.speedup_1kb_lock_step"] = ns.get("speedup_lock_step")
        flat["cpu_baseline.gpu_1kb_single_region_s"] = ns.get("single_region_s")
        flat["cpu_baseline.cpu_1kb_single_region_s"] = ns.get("cpu_reference_s")
        flat["single_region_10kb_s"] = out.get("single_region_s")
        flat["variant_config3.items_per_s_kernel"] = (out.get("variant_config3") or {}).get("items_per_s_kernel")
        flat["variant_config3.frac_of_hbm_peak"] = (out.get("variant_config3") or {}).get("frac_of_hbm_peak")
        out.update(flat)
        print(json.dumps(out), flush=True)
    for _ in slots:          # the slot threads leave (and hand their runtimes back) before the process group goes
        jobs.put(None)
    for t in slots:
        t.join()
    psdist.finalize()


if __name__ == "__main__":
    main()
