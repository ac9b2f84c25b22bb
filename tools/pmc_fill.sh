#!/bin/bash
# SQ counters of k_fill on the forward-fill micro benchmark (tools/gpu_fillbench.py): where the cycles of a step go.
# usage (on the GPU box): tools/pmc_fill.sh <tag> ; writes gpurun_out/pmc_fill_<tag>.txt
cd /tmp && export TMPDIR=/tmp
tag=${1:-x}; MODE=${2:-}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_fill_$tag.txt
: > $out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmcf
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcf -o r -- python3 $GRAFT_REPO_ROOT/tools/gpu_fillbench.py 10000 $MODE > /tmp/pmcf.log 2>&1
  f=$(find /tmp/pmcf -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    if "k_fill" in k and "like" not in k:
        for c, v in acc[k].items():
            print("%-40s %-24s per launch %.4g  (launches %d)" % (k[:40], c, v / n[(k, c)], n[(k, c)]))
PY
done
cat $out
