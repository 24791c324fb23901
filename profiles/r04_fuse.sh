#!/bin/bash
# aggregation fused into the feature-split node kernel (BSMS_FUSE_AGG): bit identity, B=1 / B=8 rates, parity tests
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
for v in 0 1; do BSMS_FUSE_AGG=$v timeout 300 python profiles/model_ab.py save /tmp/fa$v.pt 2>&1 | grep -v amdgpu | tail -1; done
python profiles/model_ab.py cmp /tmp/fa0.pt /tmp/fa1.pt
for v in 0 1; do BSMS_FUSE_AGG=$v timeout 300 python profiles/model_ab.py save /tmp/fb$v.pt airfoil 1 2>&1 | grep -v amdgpu | tail -1; done
python profiles/model_ab.py cmp /tmp/fb0.pt /tmp/fb1.pt
for r in 1 2; do for v in 0 1; do echo -n "BSMS_FUSE_AGG=$v "; BSMS_FUSE_AGG=$v timeout 300 python profiles/b1_rates.py airfoil 1 2>&1 | tail -1; echo -n "BSMS_FUSE_AGG=$v "; BSMS_FUSE_AGG=$v timeout 300 python profiles/b1_rates.py cylinder 1 2>&1 | tail -1; rate BSMS_FUSE_AGG=$v; done; done
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_bf16.py tests/test_hip_rollout.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r04_fuse.txt
