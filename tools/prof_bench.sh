#!/bin/bash
# rocprofv3 kernel trace of bench.py's default command (lock-step batches); usage on the GPU box: bash tools/prof_bench.sh <tag> [bench args]
# writes gpurun_out/prof/<tag>_kernel_stats.csv and prints the per-kernel table, the k_fill launches by grid size and the share
# of the timed step in which the GPU ran no kernel
cd /tmp && export TMPDIR=/tmp
TAG=${1:-bench}; shift
rm -rf /tmp/pb_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-extras "$@" > /tmp/pb_$TAG.log 2>&1
grep '^{' /tmp/pb_$TAG.log | tail -1 | cut -c1-160
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
cp $(find /tmp/pb_$TAG -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/prof/${TAG}_kernel_stats.csv
python3 - $(find /tmp/pb_$TAG -name '*kernel_trace.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/prof/${TAG}_trace.csv.gz <<'PY'   # (queue, kernel, times, workgroups: for offline timelines)
import csv, gzip, sys
w = gzip.open(sys.argv[2], "wt")
for r in csv.DictReader(open(sys.argv[1])):
    w.write("%s,%s,%s,%s,%d\n" % (r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", "").replace(",", ";"), r["Start_Timestamp"], r["End_Timestamp"],
                                  int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
w.close()
PY
python3 - $(find /tmp/pb_$TAG -name '*kernel_trace.csv' | head -1) /tmp/pb_$TAG.log <<'PY'
import csv, sys, json, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", ""),
                 int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
step_ms = [json.loads(l) for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1]["ms_per_step"]
hi = max(r[1] for r in rows); lo = hi - int(step_ms * 1e6)
rows = [r for r in rows if r[0] >= lo]
ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
cur = 0; last = lo; idle = 0
for t, d in ev:
    if cur == 0: idle += t - last
    last = t; cur += d
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, g in rows: tot[n] += e - s; cnt[n] += 1
print("timed step %.0f ms: %d kernels, sum of durations %.0f ms, no kernel in flight %.1f %%" % (step_ms, len(rows), sum(tot.values()) / 1e6, 100.0 * idle / (hi - lo)))
for n, v in tot.most_common(12): print("  %-26s %5d launches %9.1f ms  avg %8.1f us" % (n[:26], cnt[n], v / 1e6, v / cnt[n] / 1e3))
h = collections.defaultdict(lambda: [0, 0])
for s, e, n, g in rows:
    if n.startswith("k_fill<") or n.startswith("k_sweep"):
        b = 1
        while b < g: b *= 2
        h[b][0] += 1; h[b][1] += e - s
print("k_fill / k_sweep launches by workgroups (<= bucket): " + "  ".join("%d: %d x %.1f ms" % (b, c, d / c / 1e6) for b, (c, d) in sorted(h.items())))
PY
