#!/bin/bash
# Where do the batch slots' host threads spend a timed step?  rocprofv3 HIP-runtime + kernel trace of bench.py (1 warm-up + 1 timed step),
# reduced on the box to a per-thread summary: time blocked in stream / event synchronisation, in copies, in launches, and OUTSIDE every
# HIP call (the library's own host work + Python), plus the stream gaps by their neighbouring kernels.
# usage (GPU box): bash tools/prof_host.sh <tag> [bench args]   -> gpurun_out/<tag>_host.txt
cd /tmp && export TMPDIR=/tmp
TAG=${1:-host}; shift
rm -rf /tmp/ph_$TAG
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d /tmp/ph_$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-extras "$@" > /tmp/ph_$TAG.log 2>&1
grep '^{' /tmp/ph_$TAG.log | tail -1 | cut -c1-200
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 - "$(find /tmp/ph_$TAG -name '*hip_api_trace.csv' | head -1)" "$(find /tmp/ph_$TAG -name '*kernel_trace.csv' | head -1)" /tmp/ph_$TAG.log > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_host.txt <<'PY'
import csv, sys, json, collections
api_f, ker_f, log = sys.argv[1:4]
step_ms = [json.loads(l) for l in open(log).read().splitlines() if l.startswith("{")][-1]["ms_per_step"]
kr = []
for r in csv.DictReader(open(ker_f)):
    kr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ps::", "").split("<")[0], r["Queue_Id"]))
hi = max(k[1] for k in kr); lo = hi - int(step_ms * 1e6)
T = (hi - lo) / 1e9
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
spans = collections.defaultdict(list)
fn_tot = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(api_f)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < lo or s > hi: continue
    s = max(s, lo); e = min(e, hi)
    f = r["Function"]; t = r["Thread_Id"]
    cls = ("sync" if "Synchronize" in f else "copy" if "Memcpy" in f else "memset" if "Memset" in f else "launch" if "Launch" in f or "hipModuleLaunch" in f else
           "event" if "Event" in f else "malloc" if "Malloc" in f or "Free" in f else "other")
    per[t][cls] += (e - s) / 1e9; cnt[t][cls] += 1
    spans[t].append((s, e))
    fn_tot[f][0] += 1; fn_tot[f][1] += (e - s) / 1e9
print("timed step %.2f s; threads with HIP calls in it: %d" % (T, len(per)))
print("%-10s %8s %8s %8s %8s %8s %8s %8s | %9s %s" % ("thread", "sync", "copy", "memset", "launch", "event", "malloc", "other", "outside", "(calls: sync copy memset launch)"))
tot = collections.defaultdict(float)
rows = []
for t, d in per.items():
    inside = 0.0
    sp = sorted(spans[t]); cs, ce = sp[0]
    for s, e in sp[1:]:
        if s <= ce: ce = max(ce, e)
        else: inside += ce - cs; cs, ce = s, e
    inside += ce - cs
    outside = T - inside / 1e9
    rows.append((sum(cnt[t].values()), t, d, outside))
for n, t, d, outside in sorted(rows, reverse=True)[:20]:
    print("%-10s %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f | %9.2f (%d %d %d %d)" % (t, d["sync"], d["copy"], d["memset"], d["launch"], d["event"], d["malloc"], d["other"], outside,
          cnt[t]["sync"], cnt[t]["copy"], cnt[t]["memset"], cnt[t]["launch"]))
    if n > 1000:
        for k, v in d.items(): tot[k] += v
        tot["outside"] += outside; tot["n"] += 1
print("\nmean over the %d slot threads (seconds of a %.2f s step): " % (tot["n"], T) + ", ".join("%s %.2f" % (k, v / max(tot["n"], 1)) for k, v in tot.items() if k != "n"))
print("\nHIP calls inside the timed step, all threads: " + "; ".join("%s %d x = %.2f s" % (f, v[0], v[1]) for f, v in sorted(fn_tot.items(), key=lambda kv: -kv[1][1])[:14]))
# stream gaps by neighbours
byq = collections.defaultdict(list)
for k in kr:
    if k[0] >= lo: byq[k[3]].append(k)
agg = collections.defaultdict(lambda: [0, 0.0])
for q, rs in byq.items():
    rs.sort(); last = rs[0][1]; lastn = rs[0][2]
    for s, e, n, _ in rs[1:]:
        if s > last + 100000: a = agg[(lastn, n)]; a[0] += 1; a[1] += (s - last) / 1e9
        if e > last: last = e; lastn = n
print("\nstream gaps > 0.1 ms (no kernel of the stream running): %.1f s over %d streams" % (sum(v[1] for v in agg.values()), len(byq)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-26s -> %-26s n=%5d total %6.2f s avg %6.2f ms" % (k[0][:26], k[1][:26], v[0], v[1], 1e3 * v[1] / v[0]))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_host.txt
