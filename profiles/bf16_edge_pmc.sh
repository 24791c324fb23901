#!/bin/bash
# SQ counters of the bf16 edge chain kernels inside the bf16 training step (per launch; the L0 launches are the largest)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
out=gpurun_out/bf16_edge_pmc; mkdir -p $out
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
timeout 300 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $out/p1 -o x -- python bench.py --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $out/p2 -o x -- python bench.py --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $out/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/bf16_edge_pmc/{p}/**/x_counter_collection.csv", recursive=True)
    if not f: print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        for key in ("k_chain_bwd<8, 1, 0, true", "k_chain_fwd<8, 3, 0, false, true", "k_edge_fwd<8, 1", "k_edge_bwd<8, 2"):
            if key in n: acc[(key, int(r["Grid_Size"]) if "Grid_Size" in r else 0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        print(p, k, {c: round(max(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
