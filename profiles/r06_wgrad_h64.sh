#!/bin/bash
# k_wgrad_h64 (64-row chunks for the fp16 x 2 weight-gradient jobs) against k_wgrad's 32-row chunks: experiment build, BSMS_WGRAD_H64=0/1
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/h64
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
{
BSMS_WGRAD_H64=0 python profiles/model_ab.py save /tmp/h0.pt 2>&1 | grep -v amdgpu.ids | tail -1
BSMS_WGRAD_H64=1 python profiles/model_ab.py save /tmp/h1.pt 2>&1 | grep -v amdgpu.ids | tail -1
python profiles/model_ab.py cmp /tmp/h0.pt /tmp/h1.pt
sed -i 's/--steps 100 --warmup 10/--steps 80 --warmup 15/' profiles/ab_env.sh
BENCH_ARGS="--no-other-lines" bash profiles/ab_env.sh "BSMS_WGRAD_H64=0" "BSMS_WGRAD_H64=1"
BENCH_ARGS="--no-other-lines --dtype bf16" bash profiles/ab_env.sh "BSMS_WGRAD_H64=0" "BSMS_WGRAD_H64=1"
BENCH_ARGS="--no-other-lines --workload cylinder" bash profiles/ab_env.sh "BSMS_WGRAD_H64=0" "BSMS_WGRAD_H64=1"
BENCH_ARGS="--no-other-lines --workload surface --batch 2" bash profiles/ab_env.sh "BSMS_WGRAD_H64=0" "BSMS_WGRAD_H64=1"
} 2>&1 | tee gpurun_out/h64/ab.txt
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
