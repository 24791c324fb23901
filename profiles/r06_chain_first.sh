#!/bin/bash
# the input-gradient chain enqueued BEFORE the projections' weight-gradient strand forks (experiment build, BSMS_PROJ_WGRAD_LATE=0/1)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/cf
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
{
BSMS_PROJ_WGRAD_LATE=0 python profiles/model_ab.py save /tmp/c0.pt 2>&1 | grep -v amdgpu.ids | tail -1
BSMS_PROJ_WGRAD_LATE=1 python profiles/model_ab.py save /tmp/c1.pt 2>&1 | grep -v amdgpu.ids | tail -1
python profiles/model_ab.py cmp /tmp/c0.pt /tmp/c1.pt
sed -i 's/--steps 100 --warmup 10/--steps 80 --warmup 15/' profiles/ab_env.sh
BENCH_ARGS="--no-other-lines" bash profiles/ab_env.sh "BSMS_PROJ_WGRAD_LATE=0" "BSMS_PROJ_WGRAD_LATE=1"
BENCH_ARGS="--no-other-lines --dtype bf16" bash profiles/ab_env.sh "BSMS_PROJ_WGRAD_LATE=0" "BSMS_PROJ_WGRAD_LATE=1"
BENCH_ARGS="--no-other-lines --workload cylinder" bash profiles/ab_env.sh "BSMS_PROJ_WGRAD_LATE=0" "BSMS_PROJ_WGRAD_LATE=1"
sed -i 's/--steps 80 --warmup 15/--steps 40 --warmup 8/' profiles/ab_env.sh
BENCH_ARGS="--no-other-lines --workload surface --batch 2" bash profiles/ab_env.sh "BSMS_PROJ_WGRAD_LATE=0" "BSMS_PROJ_WGRAD_LATE=1"
} 2>&1 | tee gpurun_out/cf/ab.txt
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
