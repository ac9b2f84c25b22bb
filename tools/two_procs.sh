#!/bin/bash
# two bench processes on ONE device at once (7 batches of 20 regions each, half of the device's memory plan each): is the device or the
# process (host threads, hardware queues) what caps the bench?  usage (GPU box): bash tools/two_procs.sh  -> gpurun_out/two_a.json, two_b.json
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export PORESEQ_DEVICE_FRACTION=0.5
args="--steps 2 --warmup 1 --no-cpu --no-extras --regions-per-gpu 140 --batches-in-flight 7"
timeout 900 python bench.py $args > gpurun_out/two_a.json 2> gpurun_out/two_a.err &
a=$!
timeout 900 python bench.py $args > gpurun_out/two_b.json 2> gpurun_out/two_b.err
wait $a
python - <<'PY'
import json
for f in ("gpurun_out/two_a.json","gpurun_out/two_b.json"):
    try:
        d=[json.loads(l) for l in open(f) if l.startswith("{")][-1]; print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
