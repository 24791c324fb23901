#!/bin/bash
# compute waves per workgroup of the bf16 edge chains (BSMS_BFEDGE_CW; experiment build: knob() reads the environment there only)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { env "$@" timeout 300 python bench.py --steps 50 --warmup 12 --no-cpu-baseline --no-roofline $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp-build $ARGS $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
for r in 1 2; do
for cw in 4 6 7; do ARGS="--dtype bf16_nodes" rate BSMS_BFEDGE_CW=$cw; done
for cw in 4 6 7; do ARGS="--workload surface --batch 2 --dtype bf16" rate BSMS_BFEDGE_CW=$cw; done
for cw in 4 7; do ARGS="--workload surface --batch 2 --dtype bf16_nodes" rate BSMS_BFEDGE_CW=$cw; done
for cw in 4 7; do ARGS="--workload cylinder --dtype bf16" rate BSMS_BFEDGE_CW=$cw; done
done
