#!/bin/bash
# rocprofv3 kernel stats of ONE region refined alone (tools/gpu_onerun.py: one warm-up schedule + one measured); usage: bash tools/prof_single.sh <tag> [length]
cd /tmp && export TMPDIR=/tmp
TAG=${1:-single}; L=${2:-10000}
rm -rf /tmp/ps_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$TAG -o r -- python3 $GRAFT_REPO_ROOT/tools/gpu_onerun.py $L > /tmp/ps_$TAG.log 2>&1
tail -2 /tmp/ps_$TAG.log
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
cp $(find /tmp/ps_$TAG -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/prof/${TAG}_kernel_stats.csv
head -14 $GRAFT_REPO_ROOT/gpurun_out/prof/${TAG}_kernel_stats.csv | cut -d, -f1-5 | cut -c1-120
