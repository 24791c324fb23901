#!/bin/bash
# single-round edge kernels that pass the chunk barrier early (BSMS_EDGE_LONE): bit identity, parity, B=1 and B=8 rates
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
for v in 0 1; do BSMS_EDGE_LONE=$v timeout 300 python profiles/model_ab.py save /tmp/el8_$v.pt 2>&1 | grep -v amdgpu | tail -1; BSMS_EDGE_LONE=$v timeout 300 python profiles/model_ab.py save /tmp/el1_$v.pt airfoil 1 2>&1 | grep -v amdgpu | tail -1; BSMS_EDGE_LONE=$v timeout 300 python profiles/model_ab.py save /tmp/elc_$v.pt cylinder 1 2>&1 | grep -v amdgpu | tail -1; done
python profiles/model_ab.py cmp /tmp/el8_0.pt /tmp/el8_1.pt; python profiles/model_ab.py cmp /tmp/el1_0.pt /tmp/el1_1.pt; python profiles/model_ab.py cmp /tmp/elc_0.pt /tmp/elc_1.pt
for r in 1 2 3; do for v in 0 1; do echo -n "BSMS_EDGE_LONE=$v "; BSMS_EDGE_LONE=$v timeout 300 python profiles/b1_rates.py airfoil 1 2>&1 | tail -1; done; done
for r in 1 2; do for v in 0 1; do rate BSMS_EDGE_LONE=$v; done; done
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r04_edge_lone.txt
