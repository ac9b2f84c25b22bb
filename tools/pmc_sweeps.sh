#!/bin/bash
# Where a strip sweep's cycles go: SQ / SQC / TCP / TCC counters of the kept-column ScoreMutations fills (k_sweeps, k_sweeps_w) alone on
# the chip, per form (K rows per lane, NW wavefronts per sweep), saturated (R regions x 10 events x 2 directions) — VERDICT r4 "next" 3.
# usage (GPU box): bash tools/pmc_sweeps.sh <tag> [R] ["forms"]   -> gpurun_out/pmc_sweeps_<tag>.txt
set -u
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
tag=${1:-x}; R=${2:-120}; FORMS=${3:-"10,1 4,2 2,4"}
out="$GRAFT_REPO_ROOT/gpurun_out/pmc_sweeps_$tag.txt"
: > "$out"
export PORESEQ_SWEEP_MIN=0 PORESEQ_SPARSE_MIN=0
for form in $FORMS; do
  export PORESEQ_SWEEP_FORM=$form
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pmcs
    timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcs -o r -- python3 "$GRAFT_REPO_ROOT/tools/gpu_fillbatch.py" $R both > /tmp/pmcs.log 2>&1
    f=$(find /tmp/pmcs -name "*counter_collection.csv" | head -1)
    [ -z "$f" ] && { echo "form $form: no counters for [$set]" >> "$out"; tail -2 /tmp/pmcs.log >> "$out"; continue; }
    python3 - "$f" "$form" >> "$out" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", "")
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    if k.startswith("k_sweeps"):
        for c, v in acc[k].items():
            print("form %-5s %-26s %-30s per launch %.6g  (launches %d)" % (sys.argv[2], k[:26], c, v / n[(k, c)], n[(k, c)]))
PY
  done
done
cat "$out"
