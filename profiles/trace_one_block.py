#!/usr/bin/env python3
"""Print the kernel timeline of ONE GMP forward+backward from a rocprofv3 --kernel-trace CSV of
`LEVEL=5 ITERS=6 python profiles/micro_gmp.py`: start offset, duration, stream, kernel (last iteration)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last iteration = kernels after the second to last k_prepack
idx = [i for i, r in enumerate(rows) if "k_prepack" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {max(0, s - prev_end) / 1e3:5.1f}  q{r['Queue_Id']:>2} grid {int(r['Grid_Size_X']):7d}  {name}")
    prev_end = max(prev_end, e)
print(f"total {(prev_end - t0) / 1e3:.1f} us")
