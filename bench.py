#!/usr/bin/env python3
"""bench.py — kb of consensus refined per second at 10x coverage (BASELINE.json metric).

One "step" = the full `poreseq consensus` schedule (Mutate('self') then up to 4 x {Mutate('viterbi'),
Refine()}, poreseq/Mutate.py:70-85) over one BATCH of R independent synthetic regions, each 10 kb with 10
event streams (BASELINE.json configs[1]), through the drop-in PSAlign API and the C ABI.  Regions are the
reference's own unit of parallelism (one process per region file, README.md:48-54); a single region keeps
only a few dozen of the 256 CUs busy, so one GPU refines R regions concurrently (default 16; one host thread
with its own HIP stream(s), device pools and random stream per region).  With N GPUs every rank refines its own batch per step (weak scaling) and the value
is the whole-job rate:  N * R * region_kb * K / max-over-ranks time.  The latency of one region processed
alone is reported as well.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--length L] [--events E]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0, with the `roofline` of the dominant kernel (HIP-event time on the
library's own stream in a separate profiled pass) and the `cpu_baseline` (the reference's C++ when
oracle/_ref is built, else the oracle restatement) timed on a bounded sample.
"""
import argparse
import copy
import json
import os
import sys
import threading
import time

# concurrent regions need more than HIP's default 4 hardware queues; must be set before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")


def _pre_hip_env(argv):
    """Per-thread memory budget depends on how many regions share the GPU; read when the library first runs."""
    R = 16
    for k, a in enumerate(argv):
        if a == "--regions-per-gpu" and k + 1 < len(argv):
            R = int(argv[k + 1])
        elif a.startswith("--regions-per-gpu="):
            R = int(a.split("=", 1)[1])
    # cap the DP-matrix bytes of one seed batch per host thread (R threads share one GPU's HBM)
    os.environ.setdefault("PORESEQ_MAX_BATCH_GB", "8" if R <= 8 else "4")
    # more than ~10 regions in flight: one HIP stream per region (second streams would oversubscribe the hardware
    # queues; a region's Smith-Waterman batch then overlaps with other regions' work instead of its own realign)
    if R > 10:
        os.environ.setdefault("PORESEQ_ONE_STREAM", "1")


_pre_hip_env(sys.argv)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--length", type=int, default=10000)
    ap.add_argument("--events", type=int, default=10)
    ap.add_argument("--regions-per-gpu", type=int, default=16, help="independent regions refined concurrently on one GPU")
    ap.add_argument("--cpu-length", type=int, default=1000, help="region length of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    from poreseq_amd import dist as psdist
    rank, local, world = psdist.init()
    import torch
    from poreseq_amd import _capi, synth
    from poreseq_amd.consensus import consensus_region
    from poreseq_amd.poreseqcpp import PSAlign, swalign
    from poreseq_amd.util import DEFAULT_PARAMS

    params = dict(DEFAULT_PARAMS, verbose=0)
    api = _capi.load_hip()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)

    def make(seed):
        draft, events, truth = synth.make_region(args.length, args.events, seed, swalign, params)
        return draft, events, truth

    def run(region):
        draft, events, truth = region
        pa = PSAlign()
        pa.sequence, pa.events, pa.params = draft, copy.deepcopy(events), dict(params)
        seq, _ = consensus_region(pa, params)
        return seq, truth

    # synthetic inputs for every step of this rank, generated outside the timed region
    R = max(1, args.regions_per_gpu)
    nsteps = args.warmup + args.steps
    regions = [[make(1002 + 100000 * rank + 1000 * k + s) for s in range(nsteps)] for k in range(R)]
    # ---- before any worker thread exists: one region alone (latency), then the profiled pass for the roofline ----
    pre = {}
    if rank == 0:
        run(regions[0][0])                      # warm: device pools
        t1 = time.perf_counter()
        run(regions[0][-1])
        pre["single_region_s"] = time.perf_counter() - t1

        # ---- roofline of the dominant kernel: separate profiled pass (HIP events around each launch) ----
        api.prof_reset()
        api.prof_enable(True)
        run(regions[0][-1])
        api.prof_enable(False)
        prof = {k: api.prof_get(k) for k in ("fill", "score", "sw", "viterbi")}
        dom = max(prof, key=lambda k: prof[k][0])
        ms, launches, nbytes = prof[dom]
        achieved = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        kname = {"fill": "k_recur", "score": "k_score", "sw": "k_sw_fill", "viterbi": "k_vit_steps"}[dom]
        # HBM-side bytes per launch from the PMC counters cannot be collected from inside this process; they are
        # measured offline with rocprofv3 on the same workload shape (tools/pmc_total.sh) and committed under profiles/
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_g_traffic_per_launch.json")) as fh:
                tk = json.load(fh)["kernels"].get(kname)
            if tk and args.length == 10000 and args.events == 10:
                traffic = tk["fetch_bytes_per_launch"] + tk["write_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        pre["roofline"] = {"bound": "hbm", "kernel": kname,
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": "profiles/r01_g_traffic_per_launch.json (rocprofv3 --pmc FETCH_SIZE + "
                                                                  "WRITE_SIZE per launch, offline, same workload shape)" if traffic else None,
                           "launches": launches, "avg_launch_ms": ms / max(launches, 1),
                           "alg_bytes_per_launch": nbytes / max(launches, 1),
                           "all_kernel_classes_ms": {k: v[0] for k, v in prof.items()}}

    results = [None] * R
    gate = threading.Barrier(R + 1)
    errors = []

    def worker(k):
        try:
            for s in range(args.warmup):
                run(regions[k][s])
            gate.wait()          # everyone is warm (device pools allocated)
            gate.wait()          # the clock has started
            for s in range(args.warmup, nsteps):
                results[k] = run(regions[k][s])
        except Exception as e:   # pragma: no cover
            errors.append(e)
            gate.abort()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(R)]
    for t in threads:
        t.start()
    gate.wait()
    psdist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    gate.wait()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    psdist.barrier()
    dt = psdist.max_over_ranks(time.perf_counter() - t0)
    kb = args.length / 1000.0
    value = world * R * kb * args.steps / dt
    accs = [results[0]]

    out = {
        "metric": "kb consensus refined/sec at 10x coverage", "value": value, "unit": "kb/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / max(args.steps, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "poreseq consensus, %d kb region, %dx synthetic coverage (BASELINE configs[1]), full "
                               "Mutate.py schedule per region; a step is a batch of %d independent regions refined concurrently per GPU"
                               % (args.length // 1000, args.events, R),
                   "region_bases": args.length, "events": args.events, "regions_per_gpu": R,
                   "parallelism": "%d regions x %d GPU(s), no data-path collective" % (R, world)},
    }

    if rank == 0:
        # accuracy of the refined consensus (trimmed by end_trim) against the synthetic truth
        a0 = swalign(regions[0][-1][0], regions[0][-1][2])[0]
        a1 = swalign(accs[-1][0], accs[-1][1])[0]
        out["accuracy"] = {"draft_percent": a0, "consensus_percent": a1}
        out.update(pre)
        # all regions in flight together: HBM-side bytes of one whole schedule (offline PMC sums) x regions per second
        try:
            with open(os.path.join(ROOT, "profiles", "r01_g_traffic_per_launch.json")) as fh:
                ws = json.load(fh)["whole_schedule"]
            if args.length == 10000 and args.events == 10 and "roofline" in out:
                gbs = (ws["fetch_bytes"] + ws["write_bytes"]) / 1e9 * (value / kb) / world
                out["roofline"]["aggregate_hbm_gbs_per_gpu"] = gbs
                out["roofline"]["aggregate_hbm_frac"] = gbs / HBM_PEAK_GBS
        except (OSError, ValueError, KeyError):
            pass

        # ---- parity spot-check + CPU baseline (oracle / reference: checker and baseline only) ----
        if world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import backends as B
            use_ref = B.have_ref()
            cls = B.RefPSAlign if use_ref else B.OraclePSAlign
            cpu_sw = B.ref_swalign if use_ref else B.oracle_swalign
            d, ev, tr = synth.make_region(400, 6, 77, cpu_sw, params)
            g = B.make_pa(PSAlign, d, copy.deepcopy(ev), params).ScoreEvents()
            c = B.make_pa(cls, d, copy.deepcopy(ev), params).ScoreEvents()
            gp = B.make_pa(PSAlign, d, copy.deepcopy(ev), params).ScorePoints()
            cp = B.make_pa(cls, d, copy.deepcopy(ev), params).ScorePoints()
            rel = max([abs(x - y) / max(abs(y), 1e-300) for x, y in zip(g, c)] +
                      [abs(x.score - y.score) / abs(y.score) for x, y in zip(gp, cp) if abs(y.score) > 1e-3])
            out["logl_max_rel_err_vs_cpu"] = rel
            if not args.no_cpu:
                d, ev, tr = synth.make_region(args.cpu_length, args.events, 1002, cpu_sw, params)
                pa = B.make_pa(cls, d, copy.deepcopy(ev), params)
                B.reset_rand()
                t = time.perf_counter()
                consensus_region(pa, params)
                ct = time.perf_counter() - t
                out["cpu_baseline"] = {
                    "value": (args.cpu_length / 1000.0) / ct, "unit": "kb/s", "cores": 1,
                    "kind": "reference" if use_ref else "port",
                    "sample": "full consensus schedule on one %d-base region, %d events, single thread (%.1f s); the "
                              "reference is O(L^2) per region, so its 10 kb rate is ~5x lower per kb (BASELINE.md 2a)"
                              % (args.cpu_length, args.events, ct),
                    "host_cores_available": os.cpu_count()}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
