#!/bin/bash
# k_wgrad_bf64_wide on the surface bf16 steps (experiment build), then the D = 256 / bf16 tests on the product library.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/ww
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
{
sed -i 's/--steps 100 --warmup 10/--steps 40 --warmup 8/' profiles/ab_env.sh
BENCH_ARGS="--no-other-lines --workload surface --batch 2 --dtype bf16" bash profiles/ab_env.sh "BSMS_WGRAD_WIDE=0" "-"
BENCH_ARGS="--no-other-lines --workload surface --batch 2 --dtype bf16_nodes" bash profiles/ab_env.sh "BSMS_WGRAD_WIDE=0" "-"
} 2>&1 | tee gpurun_out/ww/ab3.txt
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "256 or surface or bf16" 2>&1 | tail -5 | tee gpurun_out/ww/tests3.txt
