#!/bin/bash
# threshold of k_wgrad_wide (total rows of a launch), experiment build, surface workload.   gpurun -- 'bash profiles/r06_wgrad_wide2.sh'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/ww
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
{
sed -i 's/--steps 100 --warmup 10/--steps 40 --warmup 8/' profiles/ab_env.sh
BENCH_ARGS="--no-other-lines --workload surface --batch 2" bash profiles/ab_env.sh "BSMS_WGRAD_WIDE=0" "BSMS_WGRAD_WIDE_MIN=0" "BSMS_WGRAD_WIDE_MIN=100000" "BSMS_WGRAD_WIDE_MIN=262144" "BSMS_WGRAD_WIDE_MIN=500000"
BENCH_ARGS="--no-other-lines --workload surface --batch 2 --dtype bf16" bash profiles/ab_env.sh "BSMS_WGRAD_WIDE=0" "BSMS_WGRAD_WIDE_MIN=0" "BSMS_WGRAD_WIDE_MIN=262144"
} 2>&1 | tee gpurun_out/ww/ab2.txt
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
