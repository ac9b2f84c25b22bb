#!/bin/bash
# total VALU work of bench.py's default command by kernel (rocprofv3 PMC pass, kernels serialised by the profiler: instruction and
# busy-cycle counts are per kernel, durations are not the bench's).  usage on the GPU box: bash tools/pmc_valu.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
TAG=${1:-valu}; shift
rm -rf /tmp/pv_$TAG
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pv_$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-extras "$@" > /tmp/pv_$TAG.log 2>&1
grep '^{' /tmp/pv_$TAG.log | tail -1 | cut -c1-140
python3 - $(find /tmp/pv_$TAG -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys, collections
t = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", "")
    t[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[name] += 1
tot = sum(v["SQ_ACTIVE_INST_VALU"] for v in t.values())
print("kernel, launches, VALU insts (G wave-insts), VALU busy (G quad-cycles = 4 cycles), share, wave-cycles (G quad), waves (M)")
for name, v in sorted(t.items(), key=lambda kv: -kv[1]["SQ_ACTIVE_INST_VALU"])[:16]:
    print("  %-30s %6d  %8.2f  %8.2f  %5.1f %%  %9.2f  %7.2f" % (name[:30], n[name], v["SQ_INSTS_VALU"] / 1e9, v["SQ_ACTIVE_INST_VALU"] / 1e9, 100 * v["SQ_ACTIVE_INST_VALU"] / tot, v["SQ_WAVE_CYCLES"] / 1e9, v["SQ_WAVES"] / 1e6))
print("total VALU busy: %.1f G quad-cycles = %.2f s of all 1024 SIMDs at 2.4 GHz" % (tot / 1e9, tot * 4 / 1024 / 2.4e9))
PY
