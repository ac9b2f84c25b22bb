#!/bin/bash
# A/B of the strip sweep's form in the bench: one and two wavefronts per sweep, alternating, three timed steps each (one box, one call).
# usage (GPU box): bash tools/ab_nw.sh
for cfg in "PORESEQ_SWEEP_NW=1" "PORESEQ_SWEEP_NW=2" "PORESEQ_SWEEP_NW=1" "PORESEQ_SWEEP_NW=2"; do
  echo "== $cfg"
  env $cfg timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('value %.1f kb/s step %.0f ms; sweep avg %.1f ms; classes %s; fp64 %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], {k: round(v) for k, v in r['all_kernel_classes_ms_per_step'].items()}, r['fp64']['frac']))
"
done
