#!/bin/bash
# do many short low-priority weight-gradient workgroups leave the chip to the chain kernels?  (experiment build)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { env "$@" bash profiles/with_exp.sh timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
for r in 1 2; do
rate X=0
rate BSMS_LANE_PRIO=-1
rate BSMS_WGRAD_WGS=256
rate BSMS_WGRAD_WGS=256 BSMS_LANE_PRIO=-1
rate BSMS_WGRAD_WGS=512 BSMS_LANE_PRIO=-1
rate BSMS_WGRAD_WGS=1024 BSMS_LANE_PRIO=-1
rate BSMS_WGRAD_WGS=512 BSMS_LANE_PRIO=1
done
} 2>&1 | tee gpurun_out/r04_prio.txt
