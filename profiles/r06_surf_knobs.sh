#!/bin/bash
# D = 256 launch-shape sweep on the surface workload (experiment build: the knobs are read from the environment).
#   gpurun -- 'bash profiles/r06_surf_knobs.sh'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/surf
cp bsms-gnn_amd/libbsms_hip.so /tmp/prod.so; cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
export BENCH_ARGS="--workload surface --batch 2 --no-other-lines ${DT}"
sed -i 's/--steps 100 --warmup 10/--steps 40 --warmup 8/' profiles/ab_env.sh
bash profiles/ab_env.sh "$@" 2>&1 | tee gpurun_out/surf/knobs${TAG}.txt
cp /tmp/prod.so bsms-gnn_amd/libbsms_hip.so
