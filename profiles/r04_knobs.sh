#!/bin/bash
# end-of-round knob sweeps on the final kernels (same box): bf16 edge-chain compute waves, ring depth, weight-gradient workgroups
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { d=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --dtype $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
ratex() { d=$1; shift; env "$@" bash profiles/with_exp.sh timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --dtype $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp-build $d $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
for cw in 4 5 6 7; do rate bf16 BSMS_BFEDGE_CW=$cw; done
for cw in 4 6; do rate bf16_nodes BSMS_BFEDGE_CW=$cw; done
rate bf16 BSMS_RING=4
rate f32 BSMS_RING=4
for w in 96 112 128 144 160; do ratex f32 BSMS_WGRAD_WGS=$w; done
for w in 96 128 160 192; do ratex bf16 BSMS_WGRAD_WGS=$w; done
} 2>&1 | tee gpurun_out/r04_knobs.txt
