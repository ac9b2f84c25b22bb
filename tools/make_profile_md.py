#!/usr/bin/env python3
"""Turn a rocprofv3 kernel_stats.csv into the markdown summary kept under profiles/.
usage: make_profile_md.py <stats.csv> <out-stem> <title> <command> [note]"""
import csv, shutil, sys
src, out, title, cmd = sys.argv[1:5]
note = sys.argv[5] if len(sys.argv) > 5 else ""
shutil.copy(src, out + ".csv")
rows = list(csv.DictReader(open(src)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
with open(out + ".md", "w") as f:
    f.write("# %s\n\nCommand: `%s`\n%s\n\nSum of kernel durations: %.1f ms.\n\n| kernel | calls | total ms | avg us | %% |\n|---|---|---|---|---|\n" % (title, cmd, note, tot / 1e6))
    for r in rows[:26]:
        f.write("| `%s` | %s | %.3f | %.1f | %.2f |\n" % (r["Name"].split("(")[0], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
