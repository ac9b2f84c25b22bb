# rocprofv3 kernel stats of an arbitrary python script: bash tools/prof_cmd.sh <tag> <script> [args...]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$TAG -o r -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/pc_$TAG.log 2>&1
tail -3 /tmp/pc_$TAG.log
f=$(find /tmp/pc_$TAG -name '*kernel_stats.csv' | head -1)
head -${TOPN:-12} $f | sed "s/([^)]*)/()/g" | cut -d, -f1-6
python3 - $(find /tmp/pc_$TAG -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_recur" in r["Kernel_Name"]:
        d[(r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(d.items()):
    print("k_recur grid %s wg %s: n=%d avg %.2f ms min %.2f max %.2f" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v)))
PY
