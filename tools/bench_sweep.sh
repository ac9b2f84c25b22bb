#!/bin/bash
# a few bench.py configurations back to back (tuning aid): bash tools/bench_sweep.sh <tag> "<env and args>" ...
TAG=$1; shift
mkdir -p gpurun_out
: > gpurun_out/${TAG}.log
for cfg in "$@"; do
    echo "== $cfg" >> gpurun_out/${TAG}.log
    env $(echo "$cfg" | tr ' ' '\n' | grep '=' | grep -v '^--' | tr '\n' ' ') timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu --no-extras $(echo "$cfg" | tr ' ' '\n' | grep -v '^[A-Z_0-9]*=' | tr '\n' ' ') 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('value %.1f kb/s  step %.0f ms  fill: %d launches avg %.1f ms, %.0f sweeps/launch, frac %.3f; classes %s' % (d['value'], d['ms_per_step'], r.get('launches', 0), r.get('avg_launch_ms', 0), r.get('sweeps_per_launch') or 0, r.get('frac', 0), {k: round(v) for k, v in r.get('all_kernel_classes_ms_per_step', {}).items()}))
" >> gpurun_out/${TAG}.log
done
cat gpurun_out/${TAG}.log
