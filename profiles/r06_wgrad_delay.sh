#!/bin/bash
# Experiment: enqueue the batched weight gradients of backward block k in front of block k+1's edge backward instead of behind block k's
# (BSMS_WGRAD_DELAY=1, experiment build).  Bit identity of a step, then steps/s alternating.   gpurun -- 'bash profiles/r06_wgrad_delay.sh'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/wd
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
{
BSMS_WGRAD_DELAY=0 python profiles/model_ab.py save /tmp/wd0.pt 2>&1 | grep -v amdgpu.ids | tail -1
BSMS_WGRAD_DELAY=1 python profiles/model_ab.py save /tmp/wd1.pt 2>&1 | grep -v amdgpu.ids | tail -1
python profiles/model_ab.py cmp /tmp/wd0.pt /tmp/wd1.pt
export BENCH_ARGS="--no-other-lines ${DT}"
sed -i 's/--steps 100 --warmup 10/--steps 80 --warmup 15/' profiles/ab_env.sh
bash profiles/ab_env.sh "BSMS_WGRAD_DELAY=0" "BSMS_WGRAD_DELAY=1"
BENCH_ARGS="--no-other-lines --dtype bf16" bash profiles/ab_env.sh "BSMS_WGRAD_DELAY=0" "BSMS_WGRAD_DELAY=1"
BENCH_ARGS="--no-other-lines --workload cylinder" bash profiles/ab_env.sh "BSMS_WGRAD_DELAY=0" "BSMS_WGRAD_DELAY=1"
} 2>&1 | tee gpurun_out/wd/ab.txt
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
