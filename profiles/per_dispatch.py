#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace CSV, grouped by grid size (= mesh level).
  python profiles/per_dispatch.py <prefix>_kernel_trace.csv k_wgrad"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
groups = defaultdict(list)
for r in rows:
    if pat in r["Kernel_Name"]:
        groups[(r["Kernel_Name"][:60], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), d in sorted(groups.items(), key=lambda kv: -kv[0][1]):
    d.sort()
    print(f"{name:60s} grid {grid:8d}  n={len(d):4d}  median {d[len(d)//2]:8.1f} us  min {d[0]:8.1f}")
