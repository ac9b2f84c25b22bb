# total HBM-side bytes of one consensus run (two rocprofv3 passes: FETCH_SIZE, WRITE_SIZE); prints per-kernel sums
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o r -- python3 $GRAFT_REPO_ROOT/tools/gpu_onerun.py > /tmp/pmc_$c.log 2>&1
  tail -1 /tmp/pmc_$c.log
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"].split("(")[0]
    tot[k] += float(r["Counter_Value"]); n[k] += 1
print(sys.argv[2], "total %.1f GB (counter unit KB x 1024)" % (sum(tot.values()) * 1024 / 1e9))
for k, v in tot.most_common(10):
    print("   %-30s %8.2f GB  in %5d dispatches" % (k[:30], v * 1024 / 1e9, n[k]))
import json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof", "traffic_%s.json" % sys.argv[2])
json.dump({k: {"bytes_per_launch": tot[k] * 1024 / n[k], "launches": n[k]} for k in tot}, open(out, "w"), indent=1)
PY
done
