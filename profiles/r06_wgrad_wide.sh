#!/bin/bash
# k_wgrad_wide (128 x 256 tiles at D = 256) against the four 128 x 128 blocks: bit identity of a surface training step, steps/s alternating
# (experiment build: BSMS_WGRAD_WIDE=0 switches it off), then the D = 256 tests on the product library.   gpurun -- 'bash profiles/r06_wgrad_wide.sh'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/ww
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
{
BSMS_WGRAD_WIDE=0 python profiles/model_ab.py save /tmp/ww0.pt surface 2 2>&1 | grep -v amdgpu.ids | tail -1
BSMS_WGRAD_WIDE=1 python profiles/model_ab.py save /tmp/ww1.pt surface 2 2>&1 | grep -v amdgpu.ids | tail -1
python profiles/model_ab.py cmp /tmp/ww0.pt /tmp/ww1.pt
sed -i 's/--steps 100 --warmup 10/--steps 40 --warmup 8/' profiles/ab_env.sh
BENCH_ARGS="--no-other-lines --workload surface --batch 2" bash profiles/ab_env.sh "BSMS_WGRAD_WIDE=0" "BSMS_WGRAD_WIDE=1"
BENCH_ARGS="--no-other-lines --workload surface --batch 2 --dtype bf16" bash profiles/ab_env.sh "BSMS_WGRAD_WIDE=0" "BSMS_WGRAD_WIDE=1"
} 2>&1 | tee gpurun_out/ww/ab.txt
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "256 or surface or d256 or wgrad" 2>&1 | tail -5 | tee gpurun_out/ww/tests.txt
