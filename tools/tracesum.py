#!/usr/bin/env python3
"""Summarise PORESEQ_TRACE output of tools/gpu_hostprof.py (measured run only)."""
import re, sys, collections
for fn in sys.argv[1:]:
    txt = open(fn).read()
    if '=== MEASURED RUN ===' in txt:
        txt = txt.split('=== MEASURED RUN ===', 1)[1]
    t = collections.defaultdict(float); c = collections.Counter()
    for l in txt.splitlines():
        m = re.match(r"\[ps\] (\S+)\s+(.*?)\s+([\d.]+) ms", l)
        if m:
            t[(m.group(1), m.group(2))] += float(m.group(3)); c[(m.group(1), m.group(2))] += 1
    print(fn, [l for l in txt.splitlines() if l.startswith('wall')], "sum of phases %.1f ms" % sum(t.values()))
    for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:16]:
        print("   %-18s %-22s n=%4d total=%8.1f ms  avg=%7.2f" % (k[0], k[1], c[k], v, v / c[k]))
