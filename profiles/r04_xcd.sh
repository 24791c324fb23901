#!/bin/bash
# D = 256 weight gradients: the four blocks of a slab on one XCD (BSMS_WGRAD_XCD)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { d=$1; shift; env "$@" timeout 200 python bench.py --workload surface --batch 2 --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --dtype $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('surface B=2 $d $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
for v in 0 1; do BSMS_WGRAD_XCD=$v timeout 300 python profiles/model_ab.py save /tmp/xc$v.pt surface 2 2>&1 | grep -v amdgpu | tail -1; done
python profiles/model_ab.py cmp /tmp/xc0.pt /tmp/xc1.pt
for r in 1 2; do for v in 0 1; do for d in f32 bf16 bf16_nodes; do rate $d BSMS_WGRAD_XCD=$v; done; done; done
} 2>&1 | tee gpurun_out/r04_xcd.txt
