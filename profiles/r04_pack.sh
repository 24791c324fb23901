#!/bin/bash
# k_pack_scale + k_prepack as one launch (BSMS_PACK_FUSED): bit identity of a training step, B=1 and B=8 rates
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rate() { env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
for v in 0 1; do BSMS_PACK_FUSED=$v timeout 300 python profiles/model_ab.py save /tmp/pk$v.pt 2>&1 | grep -v amdgpu | tail -1; done
python profiles/model_ab.py cmp /tmp/pk0.pt /tmp/pk1.pt
for v in 0 1; do BSMS_AB_DTYPE=bf16 BSMS_PACK_FUSED=$v timeout 300 python profiles/model_ab.py save /tmp/pkb$v.pt 2>&1 | grep -v amdgpu | tail -1; done
python profiles/model_ab.py cmp /tmp/pkb0.pt /tmp/pkb1.pt
for r in 1 2; do for v in 0 1; do echo -n "BSMS_PACK_FUSED=$v "; BSMS_PACK_FUSED=$v timeout 300 python profiles/b1_rates.py airfoil 1 2>&1 | tail -1; rate BSMS_PACK_FUSED=$v; done; done
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_bf16.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r04_pack.txt
