#!/bin/bash
# SQ counters of the edge-MLP chain kernels at airfoil L0 (profiles/edge_ablate.py: 13 training + 10 inference forwards per graph).
# Counters in their own passes (rocprofv3 --kernel-trace --pmc ...), 8 SQ slots per pass.   gpurun -- 'bash profiles/edge_pmc.sh'
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/edge_pmc
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
avail() { for c in "$@"; do grep -q "\b$c\b" $out/counters.txt && printf "%s " $c; done; }
P1=$(avail SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES)
P2=$(avail SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU)
echo "pass 1: $P1"; echo "pass 2: $P2"
timeout 300 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $out/p1 -o x -- python profiles/edge_ablate.py > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $out/p2 -o x -- python profiles/edge_ablate.py > $out/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/edge_pmc/{p}/**/x_counter_collection.csv", recursive=True)
    if not f:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        key = "k_edge_fwd" if "k_edge_fwd" in n else "k_edge_bwd" if "k_edge_bwd" in n else "k_chain_fwd<8, 1, 0" if "k_chain_fwd<8, 1, 0" in n else None
        if key: acc[key + n.split(key)[1][:12]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
