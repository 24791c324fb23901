#!/bin/bash
# chunk order of the weight-gradient workgroups (BSMS_WGRAD_ORDER: 0 slabs, 1 interleaved from the last rows down, 2 interleaved ascending)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { d=$1; shift; env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --dtype $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
BSMS_WGRAD_ORDER=1 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for r in 1 2 3; do for v in 0 1 2; do rate f32 BSMS_WGRAD_ORDER=$v; done; done
} 2>&1 | tee gpurun_out/r04_order.txt
