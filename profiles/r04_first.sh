#!/bin/bash
# round-4 first GPU call: test suite, aggregation variants, bench lines, B=1 / B=8 traces
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r04a/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04a/tests.log
tail -n 5 gpurun_out/r04a/tests.log
timeout 200 profiles/census/agg_rows.bin > gpurun_out/r04a/agg_rows.txt 2>&1; cat gpurun_out/r04a/agg_rows.txt
timeout 600 python bench.py > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err; tail -c 600 gpurun_out/r04a/bench_default.err
BENCH_ARGS="--batch 1" bash profiles/prof1.sh r04a_b1 > gpurun_out/r04a/breakdown_b1.txt 2>&1
bash profiles/prof1.sh r04a_b8 > gpurun_out/r04a/breakdown_b8.txt 2>&1
head -c 1500 gpurun_out/r04a/bench_default.json
