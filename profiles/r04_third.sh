#!/bin/bash
# loader-wave / ring sweep of the single-round (LONE) launches: is the B=1 floor the weight stream or the compute chain?
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
cp bsms-gnn_amd/libbsms_hip.so bsms-gnn_amd/lib_cur.so.keep
cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
for e in "BSMS_LONE_NL=1" "BSMS_LONE_NL=2" "BSMS_LONE_NL=3" "BSMS_LONE_NL=4" "BSMS_LONE_NL=2 BSMS_RING_DEEP=3" "BSMS_LONE_NL=4 BSMS_RING_DEEP=4" "BSMS_EDGE_CW=4 BSMS_LONE_NL=4" "BSMS_EDGE_CW=5 BSMS_LONE_NL=3" "BSMS_EDGE_CW=6 BSMS_LONE_NL=2" "BSMS_LONE_NL=2"; do
  printf "%-40s " "$e"; env $e timeout 200 python profiles/b1_rates.py 2>&1 | tail -1
done | tee gpurun_out/r04c/lone_sweep.txt
cp bsms-gnn_amd/lib_cur.so.keep bsms-gnn_amd/libbsms_hip.so
