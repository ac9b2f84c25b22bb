#!/usr/bin/env python3
"""Concurrency summary of a rocprofv3 kernel_trace.csv: union busy time, average overlap, time by overlap degree,
and the share of wall time in which no kernel runs."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r.get("Queue_Id", 0) or 0)))
t0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # fraction of the trace to skip at the start (warm-up)
lo = min(r[0] for r in rows); hi = max(r[1] for r in rows)
cut = lo + t0 * (hi - lo)
rows = [r for r in rows if r[0] >= cut]
lo = min(r[0] for r in rows)
ev = []
for s, e, n, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
by = collections.Counter(); cur = 0; last = lo
for t, d in ev:
    by[cur] += t - last; last = t; cur += d
wall = hi - lo
tot = sum(e - s for s, e, _, _ in rows)
print("kernels %d  wall %.3f s  sum of durations %.3f s  busy (>=1 kernel) %.3f s  mean overlap while busy %.2f" % (
    len(rows), wall / 1e9, tot / 1e9, (wall - by[0]) / 1e9, tot / max(wall - by[0], 1)))
for k in sorted(by):
    if by[k] / wall > 0.005:
        print("   %2d kernels in flight: %5.1f %% of wall" % (k, 100.0 * by[k] / wall))
print("queues used:", len(set(r[3] for r in rows)))
