#!/bin/bash
# HBM-side traffic of bench.py's default command from the rocprofv3 PMC counters (separate passes, kernel-trace only, as the
# pool requires), next to the algorithmic bytes per k_fill launch that the same command reports.
# usage on the GPU box: bash tools/pmc_bench.sh <tag>   -> gpurun_out/prof/<tag>_traffic.json (+ printed table)
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu > /tmp/pmcb_plain.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-extras > /tmp/pmcb_$c.log 2>&1
done
python3 - $OUT/${TAG}_traffic.json <<'PY'
import csv, glob, json, sys, collections
plain = [json.loads(l) for l in open("/tmp/pmcb_plain.log") if l.startswith("{")][-1]
tot = {}; n = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmcb_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    t = collections.Counter(); k = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", "").split("<")[0]
        t[name] += float(r["Counter_Value"]) * 1024.0; k[name] += 1        # counter unit: KB
    tot[c] = t; n[c] = k
    runlog = [json.loads(l) for l in open("/tmp/pmcb_%s.log" % c) if l.startswith("{")][-1]
cfg = plain["config"]
regions = cfg["regions_per_gpu"] * (plain["steps"] + plain["warmup"])
kern = {}
for name in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]), key=lambda x: -(tot["FETCH_SIZE"][x] + tot["WRITE_SIZE"][x])):
    L = max(n["FETCH_SIZE"][name], n["WRITE_SIZE"][name], 1)
    kern[name] = {"fetch_bytes_per_launch": tot["FETCH_SIZE"][name] / max(n["FETCH_SIZE"][name], 1),
                  "write_bytes_per_launch": tot["WRITE_SIZE"][name] / max(n["WRITE_SIZE"][name], 1), "launches": L,
                  "fetch_gb": tot["FETCH_SIZE"][name] / 1e9, "write_gb": tot["WRITE_SIZE"][name] / 1e9}
if "roofline" in plain and plain["roofline"]["kernel"] in kern:
    kern[plain["roofline"]["kernel"]]["alg_bytes_per_launch"] = plain["roofline"]["alg_bytes_per_launch"]
allb = sum(tot["FETCH_SIZE"].values()) + sum(tot["WRITE_SIZE"].values())
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-extras; tools/pmc_bench.sh",
       "note": "bytes = counter (KB) x 1024, per launch = sum over the run's dispatches / dispatches; FETCH_SIZE uncorrected (the guide's x2 applies to "
               "16-byte-per-lane streams; the reads here are table gathers and 8-byte column reads), WRITE_SIZE is exact for the 16-byte record stores",
       "length": cfg["region_bases"], "events": cfg["events"], "regions_per_gpu": cfg["regions_per_gpu"], "batches_in_flight": cfg["batches_in_flight"],
       "regions_per_batch": max(1, cfg["regions_per_gpu"] // cfg["batches_in_flight"]), "kernels": kern,
       "whole_schedule": {"regions": regions, "fetch_gb": sum(tot["FETCH_SIZE"].values()) / 1e9, "write_gb": sum(tot["WRITE_SIZE"].values()) / 1e9,
                          "gb_per_region": allb / 1e9 / regions},
       "bench_line": {k: plain[k] for k in ("value", "ms_per_step") if k in plain}}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("whole schedule: %.1f GB per region (fetch %.1f + write %.1f GB over %d regions)" % (allb / 1e9 / regions, out["whole_schedule"]["fetch_gb"], out["whole_schedule"]["write_gb"], regions))
for name, v in list(kern.items())[:10]:
    print("  %-14s %5d launches  fetch %8.2f GB  write %8.2f GB   per launch %.3f + %.3f GB%s" % (name, v["launches"], v["fetch_gb"], v["write_gb"],
          v["fetch_bytes_per_launch"] / 1e9, v["write_bytes_per_launch"] / 1e9, ("  (algorithmic %.3f GB)" % (v["alg_bytes_per_launch"] / 1e9)) if "alg_bytes_per_launch" in v else ""))
PY
grep '^{' /tmp/pmcb_plain.log | tail -1 > $OUT/${TAG}_bench_line.json
