cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/fb3 -o r -- python3 $GRAFT_REPO_ROOT/tools/gpu_fillbench.py 2>&1 | grep -i "score mode\|RW="
python3 - $(find /tmp/fb3 -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_recur" in r["Kernel_Name"]:
        d[(r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(d.items()):
    print("k_recur grid %s wg %s lds %s vgpr %s sgpr %s: n=%d avg %.2f ms min %.2f max %.2f" % (k + (len(v), sum(v) / len(v), min(v), max(v))))
PY
