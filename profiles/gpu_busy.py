#!/usr/bin/env python3
"""GPU busy fraction from a rocprofv3 --kernel-trace CSV: union of kernel intervals / wall span of the last N steps.
  python profiles/gpu_busy.py <prefix>_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
# keep the second half (steady state)
iv = iv[len(iv) // 2:]
span = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps.sort()
print(f"kernels {len(iv)}, span {span / 1e6:.2f} ms, busy {busy / span * 100:.1f} %, idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps "
      f"(median {gaps[len(gaps) // 2] / 1e3:.1f} us, p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.1f} us, sum of gaps > 10 us: {sum(g for g in gaps if g > 10000) / 1e6:.2f} ms)")
