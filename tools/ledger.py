#!/usr/bin/env python3
"""SIMD-time ledger of one timed bench step.

    python3 tools/ledger.py <trace.csv.gz> <kernel_resources.json> <valu.json> [step_ms] > profiles/rNN_ledger.md

trace.csv.gz  tools/prof_bench.sh's compact kernel trace (queue, kernel, start ns, end ns, workgroups) of `bench.py --steps 1 --warmup 1`
resources     per kernel: VGPRs, LDS bytes, threads per workgroup (profiles/r06_kernel_resources.json, from -Rpass-analysis=kernel-resource-usage)
valu.json     tools/pmc_round.sh's per-kernel vector-issue counters of the same command (profiles/rNN_valu.json): issuing SIMD-time per kernel

Model.  The chip has 1024 SIMDs of 8 wave slots.  A launch of W wavefronts whose kernel fits `occ` wavefronts per SIMD (512 VGPRs / its
allocation, at most 8) can hold min(W, 1024 occ) slots; it is taken to hold them for its whole duration (true of the sweeps, walkers and
Smith-Waterman strips — one long-lived wave per unit of work — an over-estimate for kernels of many short waves, marked ~).  From the trace:
per kernel RESIDENT wave-seconds (slots held x time) and, from the PMC pass, ISSUING SIMD-seconds (SQ_ACTIVE_INST_VALU x 4 cycles: time a
SIMD's vector pipe was busy for that kernel).  A wave that is resident and not issuing waits — for its own dependent instruction, a
neighbour wave's turn, memory, a barrier, or (Smith-Waterman strips) another workgroup.  Chip-wide, per instant: the SIMDs that hold no
wave at all are at least 1024 - (resident waves), those are "slots nobody asked for"; instants with no kernel in flight are launch gaps."""
import collections, gzip, json, sys

trace, resf, valuf = sys.argv[1:4]
res = json.load(open(resf))
valu = json.load(open(valuf))
THREADS = {"k_sweeps_w<4, 2>": 128, "k_sweep_w<4, 2>": 128, "k_sweeps_w<5, 2>": 128, "k_sweep_w<5, 2>": 128, "k_sw_fill_pk<8>": 512, "k_sw_fill<4, 8>": 512,
           "k_sw_trace<8>": 64, "k_sw_trace<4>": 64, "k_backtrace_s<4>": 256, "k_backtrace_s<5>": 256, "k_like_b<true>": 64, "k_like_a<true>": 256,
           "k_score<7, true>": 256, "k_score<8, true>": 256, "k_score<16, true>": 256, "k_score<32, true>": 256, "k_score<64, true>": 256,
           "k_fill<768, true, true, false>": 768, "k_fill<512, false, true, false>": 512, "k_vit_steps": 1024, "k_vit_trace": 512, "k_vit_obs_lds": 256,
           "k_vit_log": 256, "k_oldall": 256, "k_oldfin": 256, "k_backtrace": 256, "k_fill_like": 256, "k_best": 64, "k_prefix": 64, "k_old": 64,
           "k_updaterefs": 256, "k_lb": 256, "k_band": 256, "k_qlo": 256, "k_lo": 256, "k_likes": 256, "k_reduce": 256, "k_begin": 64, "k_gather_best": 64}
SHORT = {"k_score", "k_oldall", "k_oldfin", "k_vit_obs_lds", "k_vit_log", "k_like_a", "k_lb", "k_band", "k_qlo", "k_updaterefs", "k_reduce", "k_lo", "k_fill_like", "__amd"}   # many short waves per launch

rows = []
for line in gzip.open(trace, "rt"):
    q, name, s, e, wg = line.rstrip("\n").split(",")
    rows.append((int(s), int(e), name.replace(";", ","), int(wg)))
hi = max(r[1] for r in rows)
step_ms = float(sys.argv[4]) if len(sys.argv) > 4 else valu.get("step_seconds_unprofiled", 13.7) * 1e3
lo = hi - int(step_ms * 1e6)
rows = [r for r in rows if r[0] >= lo]
T = (hi - lo) / 1e9


def occ(name):
    r = res.get(name) or {}
    v = r.get("vgprs", 64)
    v = (v + 7) // 8 * 8
    return max(1, min(8, 512 // max(v, 8)))


def waves(name, wg):
    return wg * max(1, THREADS.get(name, 256) // 64)


per = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])   # launches, launch-seconds, resident wave-seconds, waves
ev = []
for s, e, n, wg in rows:
    w = waves(n, wg)
    held = min(w, 1024 * occ(n))
    d = (e - s) / 1e9
    p = per[n]
    p[0] += 1; p[1] += d; p[2] += held * d; p[3] += w
    cls = "sweep" if n.startswith("k_sweep") else "sw" if n.startswith("k_sw_") else "other"
    ev.append((s, held, cls)); ev.append((e, -held, cls))
ev.sort()
# time integrals: resident waves (all kernels; sweeps alone), SIMDs that can hold no wave, launch gaps
cur = collections.Counter(); last = lo
empty_simd = 0.0; gap = 0.0; hist = collections.Counter(); sweep_hist = collections.Counter(); inflight = 0
for t, dw, cls in ev:
    dt = (t - last) / 1e9
    tot = sum(cur.values())
    empty_simd += max(0, 1024 - tot) * dt
    if inflight == 0: gap += dt
    b = 0 if tot == 0 else 1 if tot < 512 else 2 if tot < 1024 else 3 if tot < 2048 else 4 if tot < 4096 else 5
    hist[b] += dt
    sb = 0 if cur["sweep"] == 0 else 1 if cur["sweep"] < 512 else 2 if cur["sweep"] < 1024 else 3 if cur["sweep"] < 2048 else 4
    sweep_hist[sb] += dt
    cur[cls] += dw; inflight += 1 if dw > 0 else -1; last = t

busy_total = valu["valu_busy_simd_seconds_at_2p4ghz"]          # chip-seconds (SIMD-seconds / 1024)
byk = valu.get("by_kernel", {})
print("# SIMD-time ledger of one timed bench step\n")
print("Step %.2f s (trace), %d kernels; chip = 1024 SIMDs x 8 wave slots = %.0f SIMD-seconds, %.0f slot-seconds.\n" % (T, len(rows), 1024 * T, 8192 * T))
print("| kernel | launches | avg ms | launch-s | waves / launch | occ | resident wave-s | issuing SIMD-s | issuing / resident |")
print("|---|---|---|---|---|---|---|---|---|")
tot_res = 0.0; tot_iss = 0.0
for n, (c, ls, rs, w) in sorted(per.items(), key=lambda kv: -kv[1][2]):
    share = (byk.get(n) or {}).get("valu_busy_share")
    iss = share * busy_total * 1024 if share is not None else None
    tot_res += rs
    if iss: tot_iss += iss
    if rs < 0.002 * 8192 * T and not iss: continue
    short = any(n.startswith(s) for s in SHORT)
    print("| `%s` | %d | %.2f | %.1f | %.0f | %d | %s%.0f | %s | %s |" % (n, c, 1e3 * ls / c, ls, w / c, occ(n), "~" if short else "", rs,
          "%.0f" % iss if iss else "-", "%.2f" % (iss / rs) if iss and rs and not short else "-"))
print("\nResident wave-seconds, all kernels: %.0f of %.0f slot-seconds (%.0f %%); issuing SIMD-seconds (PMC pass): %.0f of %.0f (%.0f %%)." % (
    tot_res, 8192 * T, 100 * tot_res / (8192 * T), busy_total * 1024, 1024 * T, 100 * busy_total / T))
print("\n## Where the non-issuing SIMD time is\n")
idle = 1024 * T - busy_total * 1024
print("Non-issuing SIMD-seconds: %.0f (%.0f %% of the step).\n" % (idle, 100 * idle / (1024 * T)))
print("* (iii) launch gaps — no kernel in flight: %.2f s = %.0f SIMD-seconds (%.1f %% of the step)" % (gap, 1024 * gap, 100 * gap / T))
print("* (ii) SIMDs nobody asked for — instants when fewer than 1024 waves are resident chip-wide, at least 1024 - resident SIMDs hold nothing: "
      "%.0f SIMD-seconds (%.1f %% of the step; launch gaps included)" % (empty_simd, 100 * empty_simd / (1024 * T)))
print("* (i) SIMDs that hold waves which do not issue: the rest, %.0f SIMD-seconds (%.1f %%)" % (idle - empty_simd, 100 * (idle - empty_simd) / (1024 * T)))
names = ["none", "< 512", "512 - 1023", "1024 - 2047", "2048 - 4095", ">= 4096"]
print("\nResident waves chip-wide, share of the step: " + "; ".join("%s: %.1f %%" % (names[b], 100 * hist[b] / T) for b in range(6)))
print("\nResident sweep waves, share of the step: " + "; ".join("%s: %.1f %%" % (names[b], 100 * sweep_hist[b] / T) for b in range(5)))
