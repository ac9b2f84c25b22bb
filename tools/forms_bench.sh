#!/bin/bash
# strip-sweep forms (K rows per lane on NW wavefronts) on the lock-step fill micro benchmark: ms per launch at several launch sizes
# usage (GPU box): bash tools/forms_bench.sh <tag> ["R list"] ["form list"] -> gpurun_out/forms_<tag>.txt
set -u
: "${GRAFT_REPO_ROOT:?}"
out="$GRAFT_REPO_ROOT/gpurun_out/forms_${1:-x}.txt"
RS=${2:-"2 8 20 64 240"}
FORMS=${3:-"10,1 4,2 5,2 2,4 3,4"}
: > "$out"
export PORESEQ_SWEEP_MIN=0 PORESEQ_SPARSE_MIN=0
for mode in fwd both; do
  for R in $RS; do
    [ $mode = both ] && [ $R = 240 ] && R=120
    for form in $FORMS; do
      PORESEQ_SWEEP_FORM=$form timeout 600 python3 "$GRAFT_REPO_ROOT/tools/gpu_fillbatch.py" $R $mode 2>&1 | tail -1 >> "$out"
    done
    [ -z "${NOFILL:-}" ] && PORESEQ_NO_SWEEP=1 timeout 600 python3 "$GRAFT_REPO_ROOT/tools/gpu_fillbatch.py" $R $mode 2>&1 | tail -1 >> "$out"
  done
done
cat "$out"
