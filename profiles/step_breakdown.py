#!/usr/bin/env python3
"""One training step of a rocprofv3 kernel trace of bench.py: time per kernel family on the caller's stream and on the
side lanes.   python profiles/step_breakdown.py <dir>/r_kernel_trace.csv"""
import csv, re, sys
from collections import defaultdict
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_sim_prologue" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
seg = rows[a:b]
short = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n)).split("(")[0][:40]
qs = defaultdict(list)
for r in seg:
    qs[r["Queue_Id"]].append(r)
main = max(qs, key=lambda q: len(qs[q]))
print(f"step {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.0f} us, {len(seg)} launches")
for q, rs in sorted(qs.items(), key=lambda kv: -len(kv[1])):
    fam = defaultdict(lambda: [0, 0.0])
    for r in rs:
        k = short(r["Kernel_Name"])
        fam[k][0] += 1
        fam[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in fam.values())
    print(f"queue {q} ({'caller' if q == main else 'side'}): {len(rs)} launches, busy {tot:.0f} us")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {v[1]:8.1f} us  {v[0]:3d} x  {k}")
