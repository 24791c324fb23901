#!/usr/bin/env python3
"""Per-launch durations of the edge-MLP chain kernels (k_edge_fwd / k_edge_bwd / k_chain_fwd<8, 3 / k_chain_bwd<8, 1) from a
rocprofv3 kernel trace of bench.py, grouped by position in the step (= mesh level L0..L5..L0).
  python profiles/edge_trace.py gpurun_out/prof/x_kernel_trace.csv [steps]"""
import csv
import sys
from collections import defaultdict

trace = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fam = defaultdict(list)
for r in trace:
    n = r["Kernel_Name"]
    for key, tag in (("k_edge_fwd", "fwd"), ("k_edge_bwd", "bwd"), ("k_edge32_fwd", "fwd"), ("k_edge32_bwd", "bwd"),
                     ("k_chain_fwd<8, 3, 0", "fwd"), ("k_chain_bwd<8, 1, 0", "bwd")):
        if key in n:
            fam[tag].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n.split("k_")[1][:22]))
unet = [0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0]
for tag, v in fam.items():
    per = len(v) // steps
    print(tag, len(v), "launches,", per, "per step")
    if per != 11:
        continue
    tot = 0
    for pos in range(11):
        d = [x[0] for x in v[pos::11]]
        tot += sum(d) / len(d)
        print(f"  pos {pos:2d} L{unet[pos]} {v[pos][1]:24s} avg {sum(d) / len(d) / 1e3:7.1f} us  min {min(d) / 1e3:7.1f}")
    print(f"  total {tot / 1e3:.1f} us/step")
