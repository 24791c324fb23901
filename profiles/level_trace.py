#!/usr/bin/env python3
"""Per-launch durations of one kernel family by position in the step (11 launches per step = levels L0..L5..L0):
  python profiles/level_trace.py <dir>/r_kernel_trace.csv "k_chain_fwd<8, 1, 0" [steps]"""
import csv, sys
trace = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
v = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace if key in r["Kernel_Name"]]
per = len(v) // steps
unet = [0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0]
out = []
for pos in range(per):
    d = v[pos::per]
    out.append(f"L{unet[pos] if per == 11 else pos}:{sum(d) / len(d) / 1e3:.1f}")
print(key, per, "per step |", " ".join(out), "| total", round(sum(v) / steps / 1e3, 1), "us/step")
