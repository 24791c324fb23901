#!/bin/bash
# remaining launcher knobs re-checked with the experiment build (f32, batch 8 and batch 1)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
for r in 1 2; do
for cfg in "X=0" "BSMS_FS_ROWS=0" "BSMS_FS_ROWS=12288" "BSMS_FS_ROWS_BWD=0" "BSMS_FS_ROWS_BWD=6144" "BSMS_RING_DEEP=4" "BSMS_LONE_NL=1" "BSMS_CHAIN_NL=2"; do rate $cfg; done
for cfg in "X=0" "BSMS_FS_ROWS_BWD=6144" "BSMS_RING_DEEP=4" "BSMS_LONE_NL=1"; do echo -n "$cfg "; env $cfg timeout 300 python profiles/b1_rates.py airfoil 1 2>&1 | tail -1; done
done
