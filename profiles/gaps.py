#!/usr/bin/env python3
"""Idle gaps of the caller's stream inside one training step, from a rocprofv3 kernel trace of bench.py:
  python profiles/gaps.py <dir>/r_kernel_trace.csv [min_gap_us]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
idx = [i for i, r in enumerate(rows) if "k_sim_prologue" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
seg = rows[a:b]
short = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n))[:44]
qs = {}
for r in seg:
    qs.setdefault(r["Queue_Id"], []).append(r)
main = max(qs, key=lambda q: len(qs[q]))
prev, tot, n = None, 0.0, 0
hist = {}
for r in qs[main]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if prev is not None and s - prev[0] > thr:
        key = (prev[1].split("<")[0].split("(")[0], short(r["Kernel_Name"]).split("<")[0].split("(")[0])
        hist.setdefault(key, []).append(s - prev[0])
        tot += s - prev[0]; n += 1
    prev = (e, short(r["Kernel_Name"]))
step = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in qs[main])
print(f"step {step:.0f} us, caller's stream busy {busy:.0f} us, {n} gaps > {thr} us = {tot:.0f} us")
for k, v in sorted(hist.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {len(v):3d} x avg {sum(v) / len(v):6.1f} us  after {k[0]:28s} before {k[1]}")
