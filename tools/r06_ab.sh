#!/bin/bash
# round 6: A/B of prebuilt variant libraries (tools/_libs/lib_<name>.so) on ONE box.
#   bash tools/r06_ab.sh micro "<names>"        the sweep micro benchmark (20 / 2400 forward sweeps, 400 / 2400 both directions) per variant
#   bash tools/r06_ab.sh bench "<names>" [reps] [bench args]   bench.py --steps 2 --warmup 1 per variant, alternating, reps times
#   a name of the form  lib@VAR=value,VAR2=value  runs library variant `lib` (`.` = the tree's own) under those environment variables
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT"
mode=$1; names=$2
lib=poreseq_amd/csrc/libporeseq_hip.so
cp $lib /tmp/lib_keep.so || exit 1
trap 'cp /tmp/lib_keep.so "$lib"' EXIT
mkdir -p gpurun_out
if [ "$mode" = kstats ]; then
  # per-kernel average durations (rocprofv3 --kernel-trace --stats) of the forward micro benchmark, 20 regions x 10 events, per variant
  export PORESEQ_SWEEP_MIN=0 PORESEQ_SPARSE_MIN=0 PORESEQ_SWEEP_FORM=4,2
  for v in $names; do
    cp tools/_libs/lib_$v.so $lib || continue
    rm -rf /tmp/ks_$v; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o r -- python3 $GRAFT_REPO_ROOT/tools/gpu_fillbatch.py ${3:-20} ${4:-fwd} > /tmp/ks_$v.log 2>&1)
    echo "== $v: $(tail -1 /tmp/ks_$v.log)"
    python3 - "$(find /tmp/ks_$v -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:9]:
    print("   %-40s calls %5s  avg %9.1f us" % (r["Name"].split("(")[0].replace("void ", "").replace("ps::", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
elif [ "$mode" = micro ]; then
  export PORESEQ_SWEEP_MIN=0 PORESEQ_SPARSE_MIN=0
  for v in $names; do
    cp tools/_libs/lib_$v.so $lib || continue
    for spec in "20 fwd 4,2" "240 fwd 4,2" "20 both 4,2" "120 both 4,2" "20 fwd 2,4"; do
      set -- $spec
      echo "$v: $(PORESEQ_SWEEP_FORM=$3 timeout 600 python3 tools/gpu_fillbatch.py $1 $2 2>&1 | tail -1)"
    done
  done
else
  reps=${3:-2}; shift; shift; shift
  for r in $(seq 1 $reps); do
    for spec in $names; do
      l=${spec%%@*}; envs=""; [ "$spec" != "$l" ] && envs=$(echo "${spec#*@}" | tr ',' ' ')
      v=$(echo "$spec" | tr '@=,/' '____')
      if [ "$l" = "." ]; then cp /tmp/lib_keep.so $lib; else cp tools/_libs/lib_$l.so $lib || continue; fi
      env $envs timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-extras "$@" > gpurun_out/ab_${v}_$r.json 2> gpurun_out/ab_${v}_$r.err
      echo "$v rep $r: $(python - gpurun_out/ab_${v}_$r.json <<'PY'
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    r = d.get("roofline", {})
    print("%.1f kb/s, %.0f ms/step; sweep class avg launch %.1f ms, classes ms/step %s" % (d["value"], d["ms_per_step"], r.get("avg_launch_ms", 0), {k: round(v) for k, v in r.get("all_kernel_classes_ms_per_step", {}).items()}))
except Exception as e:
    print("failed", e)
PY
)"
    done
  done
fi
