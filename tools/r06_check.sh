#!/bin/bash
# round-6 working call on the GPU box: the whole -m gpu suite, the sweep forms' micro benchmark at two launch sizes, a short bench.
# usage: gpurun -- bash tools/r06_check.sh <tag> [notest] [nobench]
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT"
tag=${1:-x}
mkdir -p gpurun_out
if [[ " $* " != *" notest "* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
  echo "tests rc=$?"; tail -4 gpurun_out/${tag}_tests.log
fi
bash tools/forms_bench.sh ${tag} "20 240" "4,2 2,4" > /dev/null 2>&1
cat gpurun_out/forms_${tag}.txt
if [[ " $* " != *" nobench "* ]]; then
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-extras > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
  python tools/show_bench.py gpurun_out/${tag}_bench.json 2>&1 | head -30
fi
