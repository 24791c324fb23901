#!/bin/bash
# workgroups per weight-gradient launch, separately for the two side lanes (experiment build).   gpurun -- 'bash profiles/r06_wgs.sh'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/wgs
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
sed -i 's/--steps 100 --warmup 10/--steps 80 --warmup 15/' profiles/ab_env.sh
BENCH_ARGS="--no-other-lines" bash profiles/ab_env.sh "-" "BSMS_WGRAD_WGS_BF3=64" "BSMS_WGRAD_WGS_BF3=96" "BSMS_WGRAD_WGS_BF3=32" "BSMS_WGRAD_WGS=96 BSMS_WGRAD_WGS_BF3=64" "BSMS_WGRAD_WGS=112 BSMS_WGRAD_WGS_BF3=64" "BSMS_WGRAD_WGS=160 BSMS_WGRAD_WGS_BF3=64" 2>&1 | tee gpurun_out/wgs/ab.txt
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
