#!/bin/bash
# XCD pairing of the FOUR workgroups of a slab in the 128 x 128 tilings at D = 256 (k_wgrad, k_wgrad_bf64): product library vs the previous one
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/ww
{
BENCH="--workload surface --batch 2 --no-other-lines" bash profiles/ab_libs.sh ww2 cur 2>&1 | tail -5
BENCH="--workload surface --batch 2 --no-other-lines --dtype bf16" bash profiles/ab_libs.sh ww2 cur 2>&1 | tail -4
BENCH="--no-other-lines" bash profiles/ab_libs.sh ww2 cur 2>&1 | tail -4
} | tee gpurun_out/ww/ab5.txt
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "256 or surface or other_widths" 2>&1 | tail -3 | tee gpurun_out/ww/tests5.txt
