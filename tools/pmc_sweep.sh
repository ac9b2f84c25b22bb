#!/bin/bash
# SQ counters of the DP-fill kernels on the lock-step fill micro benchmark (tools/gpu_fillbatch.py R fwd): k_sweep (one wave per
# alignment) against k_fill (one workgroup per alignment / pair) on the same 240 x 10 forward alignments of 10 kb.
# usage (on the GPU box): bash tools/pmc_sweep.sh <tag> [R]  -> gpurun_out/pmc_sweep_<tag>.txt
cd /tmp && export TMPDIR=/tmp
tag=${1:-x}; R=${2:-240}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sweep_$tag.txt
: > $out
for mode in sweep fill; do
  if [ $mode = fill ]; then export PORESEQ_NO_SWEEP=1; else unset PORESEQ_NO_SWEEP; export PORESEQ_SWEEP_MIN=0; fi
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "WRITE_SIZE" "FETCH_SIZE"; do
    rm -rf /tmp/pmcs
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcs -o r -- python3 $GRAFT_REPO_ROOT/tools/gpu_fillbatch.py $R fwd > /tmp/pmcs.log 2>&1
    f=$(find /tmp/pmcs -name "*counter_collection.csv" | head -1)
    python3 - "$f" $mode >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", "")
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    if k.startswith("k_fill<") or k.startswith("k_sweep<"):
        for c, v in acc[k].items():
            print("%-6s %-34s %-22s per launch %.5g  (launches %d)" % (sys.argv[2], k[:34], c, v / n[(k, c)], n[(k, c)]))
PY
  done
done
cat $out
