#!/bin/bash
# The round's PMC summaries of bench.py's default command, in the shapes bench.py reads (profiles/rNN_traffic.json, profiles/rNN_valu.json):
#   plain run (the bench line the counters are compared with), FETCH_SIZE pass, WRITE_SIZE pass (separate passes, kernel-trace only, as
#   the pool requires), one pass with the SQ counters of vector issue.
# usage on the GPU box: bash tools/pmc_round.sh <tag>   -> gpurun_out/prof/<tag>_traffic.json, <tag>_valu.json, <tag>_bench_line.json
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r05}
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof"; mkdir -p "$OUT"
python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu > /tmp/pmcr_plain.log 2>&1
grep -q '^{' /tmp/pmcr_plain.log || { echo "pmc_round: the plain bench run printed no JSON line" >&2; tail -5 /tmp/pmcr_plain.log >&2; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "/tmp/pmcr_$c"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "/tmp/pmcr_$c" -o r -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu --no-extras > "/tmp/pmcr_$c.log" 2>&1
done
rm -rf /tmp/pmcr_valu
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pmcr_valu -o r -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-extras > /tmp/pmcr_valu.log 2>&1
python3 - "$OUT/${TAG}" <<'PY'
import csv, glob, json, sys, collections
pre = sys.argv[1]
plain = [json.loads(l) for l in open("/tmp/pmcr_plain.log") if l.startswith("{")][-1]
cfg = plain["config"]
short = lambda n: n.split("(")[0].replace("void ", "").replace("ps::", "").split("<")[0]   # (k_sweeps_w<4, 2> -> k_sweeps_w)
# ---- traffic
tot = {}; n = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmcr_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    t = collections.Counter(); k = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        t[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0; k[short(r["Kernel_Name"])] += 1        # counter unit: KB
    tot[c] = t; n[c] = k
regions = cfg["regions_per_gpu"] * (plain["steps"] + plain["warmup"])
kern = {}
for name in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]), key=lambda x: -(tot["FETCH_SIZE"][x] + tot["WRITE_SIZE"][x])):
    kern[name] = {"fetch_bytes_per_launch": tot["FETCH_SIZE"][name] / max(n["FETCH_SIZE"][name], 1),
                  "write_bytes_per_launch": tot["WRITE_SIZE"][name] / max(n["WRITE_SIZE"][name], 1), "launches": max(n["FETCH_SIZE"][name], n["WRITE_SIZE"][name], 1),
                  "fetch_gb": tot["FETCH_SIZE"][name] / 1e9, "write_gb": tot["WRITE_SIZE"][name] / 1e9}
# the profile class "sweep" of bench.py covers k_sweep (forward-only), k_sweeps (kept columns) and k_sweep2 (full records): one entry for the class
cls = {"k_sweep": ("k_sweep", "k_sweeps", "k_sweep2", "k_sweep_w", "k_sweeps_w", "k_sweep2_w"), "k_fill": ("k_fill", "k_fill_wide")}
for cname, members in cls.items():
    fb = sum(tot["FETCH_SIZE"][m] for m in members); wb = sum(tot["WRITE_SIZE"][m] for m in members)
    L = sum(max(n["FETCH_SIZE"][m], n["WRITE_SIZE"][m]) for m in members)
    if L:
        kern["class:" + cname] = {"members": [m for m in members if n["FETCH_SIZE"][m] or n["WRITE_SIZE"][m]], "launches": L,
                                  "fetch_bytes_per_launch": fb / L, "write_bytes_per_launch": wb / L, "fetch_gb": fb / 1e9, "write_gb": wb / 1e9}
roof = plain.get("roofline", {})
rk = "class:k_sweep" if roof.get("kernel", "").startswith("k_sweep") else "class:" + roof.get("kernel", "")
if rk in kern:
    kern[rk]["alg_bytes_per_launch"] = roof["alg_bytes_per_launch"]
allb = sum(tot["FETCH_SIZE"].values()) + sum(tot["WRITE_SIZE"].values())
shape = {"length": cfg["region_bases"], "events": cfg["events"], "regions_per_gpu": cfg["regions_per_gpu"], "batches_in_flight": cfg["batches_in_flight"],
         "regions_per_batch": max(1, cfg["regions_per_gpu"] // cfg["batches_in_flight"])}
out = dict(shape)
out.update({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-extras; tools/pmc_round.sh",
            "note": "bytes = counter (KB) x 1024, per launch = sum over the run's dispatches / dispatches; FETCH_SIZE uncorrected (the guide's x2 applies to "
                    "16-byte-per-lane streams; the reads here are table gathers and 8-byte column reads), WRITE_SIZE is exact for the 16-byte record stores",
            "kernels": kern,
            "whole_schedule": {"regions": regions, "fetch_gb": sum(tot["FETCH_SIZE"].values()) / 1e9, "write_gb": sum(tot["WRITE_SIZE"].values()) / 1e9, "gb_per_region": allb / 1e9 / regions},
            "bench_line": {k: plain[k] for k in ("value", "ms_per_step") if k in plain}})
json.dump(out, open(pre + "_traffic.json", "w"), indent=1)
print("whole schedule: %.1f GB per region (fetch %.1f + write %.1f GB over %d regions)" % (allb / 1e9 / regions, out["whole_schedule"]["fetch_gb"], out["whole_schedule"]["write_gb"], regions))
for name, v in list(kern.items())[:12]:
    print("  %-16s %6d launches  fetch %9.2f GB  write %9.2f GB   per launch %.3f + %.3f GB" % (name, v["launches"], v["fetch_gb"], v["write_gb"], v["fetch_bytes_per_launch"] / 1e9, v["write_bytes_per_launch"] / 1e9))
# ---- vector issue
f = glob.glob("/tmp/pmcr_valu/**/*counter_collection.csv", recursive=True)[0]
t = collections.defaultdict(collections.Counter); nl = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", "")
    t[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": nl[name] += 1
busy = sum(v["SQ_ACTIVE_INST_VALU"] for v in t.values())
W = 2 * 300 + 1
cells = float(cfg["region_bases"] - 4) * W                      # band cells of one 10 kb sweep at realign_width 300
import re
sweepk = [k for k in t if k.startswith("k_sweep")]
def form(k):   # rows per lane, wavefronts per sweep: k_sweeps<10, true> / k_sweeps_w<4, 2>
    m = re.match(r"k_sweep\w*<(\d+), (\w+)>", k)
    return (int(m.group(1)), int(m.group(2)) if m.group(2).isdigit() else 1) if m else (10, 1)
sw_inst = sum(t[k]["SQ_INSTS_VALU"] for k in sweepk)
sw_waves = sum(t[k]["SQ_WAVES"] / form(k)[1] for k in sweepk)                                   # sweeps, not wavefronts
lane_cells = sum(t[k]["SQ_WAVES"] * (cfg["region_bases"] - 4 + 0.95 * cfg["region_bases"] / form(k)[0]) * form(k)[0] for k in sweepk)   # steps x rows per lane, per wavefront
step_s = plain["ms_per_step"] / 1e3
valu = dict(shape)
valu.update({"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extras; tools/pmc_round.sh",
             "note": "one step of the bench under the profiler (kernels serialised: counts are the step's, durations are not); busy_frac = sum of SQ_ACTIVE_INST_VALU x 4 cycles over "
                     "1024 SIMDs x 2.4 GHz x the step time of the unprofiled run",
             "valu_busy_quad_cycles": busy, "valu_busy_simd_seconds_at_2p4ghz": busy * 4 / 1024 / 2.4e9, "step_seconds_unprofiled": step_s,
             "busy_frac": busy * 4 / 1024 / 2.4e9 / step_s,
             "sweep_kernels": {"wave_insts_per_sweep": sw_inst / max(sw_waves, 1), "sweeps": sw_waves,
                               "lane_insts_per_band_cell": 64.0 * sw_inst / max(sw_waves, 1) / cells,
                               "insts_per_lane_cell": sw_inst / max(lane_cells, 1),
                               "forms": {k: {"rows_per_lane": form(k)[0], "wavefronts_per_sweep": form(k)[1], "sweeps": t[k]["SQ_WAVES"] / form(k)[1]} for k in sweepk},
                               "fp64_ops_per_cell_reference_arithmetic": 45},
             "by_kernel": {k: {"launches": nl[k], "valu_insts_g": v["SQ_INSTS_VALU"] / 1e9, "valu_busy_share": v["SQ_ACTIVE_INST_VALU"] / busy, "waves_m": v["SQ_WAVES"] / 1e6}
                           for k, v in sorted(t.items(), key=lambda kv: -kv[1]["SQ_ACTIVE_INST_VALU"])[:16]}})
json.dump(valu, open(pre + "_valu.json", "w"), indent=1)
print("VALU busy: %.2f s of all SIMDs at 2.4 GHz over a %.2f s step = %.2f; sweeps: %.2f M wave-instructions each, %.1f lane-instructions per band cell" % (
    valu["valu_busy_simd_seconds_at_2p4ghz"], step_s, valu["busy_frac"], sw_inst / max(sw_waves, 1) / 1e6, valu["sweep_kernels"]["lane_insts_per_band_cell"]))
for k, v in list(valu["by_kernel"].items())[:10]:
    print("  %-34s %6d launches %9.2f G insts  %5.1f %%" % (k[:34], v["launches"], v["valu_insts_g"], 100 * v["valu_busy_share"]))
PY
grep '^{' /tmp/pmcr_plain.log | tail -1 > "$OUT/${TAG}_bench_line.json"
