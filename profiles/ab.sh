#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box variance of the GPU pool is +-3 %, larger than most tuning steps):
#   cp bsms-gnn_amd/libbsms_hip.so bsms-gnn_amd/lib_A.so.keep     # build A, then build B likewise
#   gpurun -- 'bash profiles/ab.sh A B'          (BENCH_ARGS="--dtype bf16" for another bench line)
# alternates the two builds three times, 100 timed steps each, and prints steps/s.
set -e
cd "$(dirname "$0")/../bsms-gnn_amd"
for r in 1 2 3; do
  for v in "$@"; do
    cp lib_$v.so.keep libbsms_hip.so
    printf "%s " "$v"
    (cd ..; timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],2))")
  done
done
