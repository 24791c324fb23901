#!/bin/bash
# the input-gradient launch through the two edge projections (k_chain_fwd<8, IN_ROWS2, OUT_PLAIN>, accumulate) on the single-round build
# (x2 kept in registers): experiment build, BSMS_ROWS2_LONE=0/1.   gpurun -- 'bash profiles/r06_rows2.sh'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/rows2
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
{
BSMS_ROWS2_LONE=0 python profiles/model_ab.py save /tmp/q0.pt 2>&1 | grep -v amdgpu.ids | tail -1
BSMS_ROWS2_LONE=1 python profiles/model_ab.py save /tmp/q1.pt 2>&1 | grep -v amdgpu.ids | tail -1
python profiles/model_ab.py cmp /tmp/q0.pt /tmp/q1.pt
sed -i 's/--steps 100 --warmup 10/--steps 80 --warmup 15/' profiles/ab_env.sh
BENCH_ARGS="--no-other-lines" bash profiles/ab_env.sh "BSMS_ROWS2_LONE=0" "BSMS_ROWS2_LONE=1"
BENCH_ARGS="--no-other-lines --dtype bf16" bash profiles/ab_env.sh "BSMS_ROWS2_LONE=0" "BSMS_ROWS2_LONE=1"
BENCH_ARGS="--no-other-lines --workload cylinder" bash profiles/ab_env.sh "BSMS_ROWS2_LONE=0" "BSMS_ROWS2_LONE=1"
} 2>&1 | tee gpurun_out/rows2/ab.txt
rm -rf gpurun_out/rows2/prof; BSMS_ROWS2_LONE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rows2/prof -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-lines > gpurun_out/rows2/prof.log 2>&1
grep "k_chain_fwd<8, 1, 1" gpurun_out/rows2/prof/r_kernel_stats.csv | cut -c1-200
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
