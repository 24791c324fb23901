#!/bin/bash
# XCD pairing of the two workgroups of a slab (k_wgrad_wide): product library vs the previous product library (lib_ww1.so.keep), same box
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/ww
{
BENCH="--workload surface --batch 2 --no-other-lines" bash profiles/ab_libs.sh ww1 cur 2>&1 | tail -5
BENCH="--workload surface --batch 2 --no-other-lines --dtype bf16" bash profiles/ab_libs.sh ww1 cur 2>&1 | tail -4
} | tee gpurun_out/ww/ab4.txt
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "256 or surface" 2>&1 | tail -3 | tee gpurun_out/ww/tests4.txt
