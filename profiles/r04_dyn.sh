#!/bin/bash
# claimed tiles in the backward chains (BSMS_DYN_TILES): parity tests, then same-box A/B
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { d=$1; shift; env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --dtype $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
timeout 600 python -m pytest tests/test_hip_bf16.py tests/test_hip_parity.py tests/test_hip_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for r in 1 2 3; do for v in 0 1; do for d in ${DTYPES:-bf16 bf16_nodes f32}; do rate $d BSMS_DYN_TILES=$v; done; done; done
} 2>&1 | tee gpurun_out/r04_dyn.txt
