#!/usr/bin/env python3
"""Print the kernel timeline around the n-th launch of a kernel from a rocprofv3 kernel_trace.csv.
usage: timeline.py trace.csv <kernel substring> <n> <before> <after>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ps::", "").replace("void ", ""), r["Queue_Id"], r["Grid_Size_X"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[2]]
k = idx[int(sys.argv[3])]
t0 = rows[k][0]
for r in rows[max(0, k - int(sys.argv[4])):k + int(sys.argv[5])]:
    print("%9.3f ms  +%8.3f ms  q%-3s grid %-7s %s" % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e6, r[3], r[4], r[2]))
