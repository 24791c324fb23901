#!/bin/bash
# the idle gap of the caller's stream at the start of every training step (round 6): host-ahead probe, same-box A/B of the library
# with the lanes enqueued in front of / behind block 0, kernel trace of the new order.    gpurun -- 'bash profiles/r06_gap.sh'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
out=gpurun_out/gap; mkdir -p $out
python profiles/host_ahead.py f32 2>&1 | grep -v amdgpu.ids | tee $out/host_ahead.txt
bash profiles/ab_libs.sh base cur 2>&1 | tee $out/ab_f32.txt
BENCH="--dtype bf16" bash profiles/ab_libs.sh base cur 2>&1 | tail -4 | tee $out/ab_bf16.txt
rm -rf $out/prof; rocprofv3 --kernel-trace --output-format csv -d $out/prof -o r -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-other-lines > $out/prof.log 2>&1
python /dev/stdin $out/prof/r_kernel_trace.csv <<'PY' | tee $out/gaps.txt
import csv, sys
from collections import Counter
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_sim_prologue" in r["Kernel_Name"]]
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    main = Counter(r["Queue_Id"] for r in seg).most_common(1)[0][0]
    t0 = int(seg[0]["Start_Timestamp"]); pe = None; big = []; tot = 0
    for r in seg:
        if r["Queue_Id"] != main: continue
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if pe is not None:
            tot += max(0, s - pe)
            if s - pe > 8000: big.append((round(pe / 1e3), round((s - pe) / 1e3)))
        pe = e
    print(f"step {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:7.0f} us  caller's stream idle {tot / 1e3:6.0f} us  gaps > 8 us (at, length): {big}")
PY
