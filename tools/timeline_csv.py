#!/usr/bin/env python3
"""Per-dispatch timeline of the last dispatches from a rocprofv3 kernel_trace.csv: tools/timeline_csv.py file.csv [count]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"]
    m = re.search(r"(k_\w+)<([^>]*)", nm)
    short = (m.group(1) + "<" + m.group(2)[:30] + ">") if m else nm[:50]
    gap = (s - prev) / 1e3 if prev else 0.0
    grid = "x".join(str(int(r.get(f"Grid_Size_{a}", 0)) // max(int(r.get(f"Workgroup_Size_{a}", 1)), 1)) for a in "XYZ")
    print(f"{(s - t0) / 1e3:9.1f} +{gap:6.1f} {(e - s) / 1e3:8.1f} us  grid {grid:14s} {short}")
    prev = e
