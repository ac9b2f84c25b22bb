#!/bin/bash
# Memory-path counters of one bench step by kernel: L1 (TCP) accesses, L1 -> L2 read / write requests, L2 requests, vector-memory instructions.
# (kernels are serialised under --pmc: the COUNTS are the step's, the durations are not)
# usage (GPU box): bash tools/pmc_mem.sh <tag>  -> gpurun_out/prof/<tag>_mem.json + a table
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r06}
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof"; mkdir -p "$OUT"
rm -rf /tmp/pmcm
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d /tmp/pmcm -o r -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-extras > /tmp/pmcm.log 2>&1
python3 - "$OUT/${TAG}_mem.json" <<'PY'
import csv, glob, json, sys, collections
f = glob.glob("/tmp/pmcm/**/*counter_collection.csv", recursive=True)[0]
t = collections.defaultdict(collections.Counter); nl = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", "")
    t[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_REQ_sum": nl[name] += 1
tot = collections.Counter()
for v in t.values(): tot.update(v)
out = {"source": "rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extras; tools/pmc_mem.sh",
       "totals": dict(tot), "by_kernel": {}}
print("%-34s %7s %12s %12s %12s %12s %10s %10s" % ("kernel", "launch", "L1 acc (G)", "L1->L2 rd(G)", "L1->L2 wr(G)", "L2 req (G)", "vmem rd(G)", "lds (G)"))
for name, v in sorted(t.items(), key=lambda kv: -kv[1]["TCP_TOTAL_CACHE_ACCESSES_sum"])[:22]:
    out["by_kernel"][name] = dict(v, launches=nl[name])
    print("%-34s %7d %12.2f %12.2f %12.2f %12.2f %10.2f %10.2f" % (name[:34], nl[name], v["TCP_TOTAL_CACHE_ACCESSES_sum"] / 1e9, v["TCP_TCC_READ_REQ_sum"] / 1e9, v["TCP_TCC_WRITE_REQ_sum"] / 1e9,
          v["TCC_REQ_sum"] / 1e9, v["SQ_INSTS_VMEM_RD"] / 1e9, v["SQ_INSTS_LDS"] / 1e9))
print("%-34s %7s %12.2f %12.2f %12.2f %12.2f %10.2f %10.2f" % ("all kernels", "", tot["TCP_TOTAL_CACHE_ACCESSES_sum"] / 1e9, tot["TCP_TCC_READ_REQ_sum"] / 1e9, tot["TCP_TCC_WRITE_REQ_sum"] / 1e9,
      tot["TCC_REQ_sum"] / 1e9, tot["SQ_INSTS_VMEM_RD"] / 1e9, tot["SQ_INSTS_LDS"] / 1e9))
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
