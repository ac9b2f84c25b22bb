#!/bin/bash
# round 6 measurement call: the whole -m gpu suite, the round's PMC passes (traffic / vector issue), the kernel trace of the bench,
# the saturated-sweep counters.  usage: gpurun --timeout 3000 -- bash tools/r06_final.sh
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06_gputests.log
bash tools/pmc_round.sh r06 > gpurun_out/r06_pmc_round.log 2>&1; echo "pmc_round rc=$?"; tail -25 gpurun_out/r06_pmc_round.log
bash tools/prof_bench.sh r06_e > gpurun_out/r06_prof_bench.log 2>&1; tail -18 gpurun_out/r06_prof_bench.log
