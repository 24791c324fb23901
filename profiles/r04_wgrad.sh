#!/bin/bash
# same-box A/B of the weight-gradient kernels: bit identity, chunk-step timeline (experiment build), training-step rate
#   BSMS_WGRAD_PIPE: 0 = k_wgrad (round 3), 2 / 3 = k_wgrad_h2p, 4 = k_wgrad_h2w;  BSMS_WGRAD_WGS = workgroup target of a launch
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-r04w}; mkdir -p gpurun_out/$tag
rate() { env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
BSMS_WGRAD_PIPE=0 timeout 300 python profiles/wgrad_ab.py save /tmp/wg0.pt
BSMS_WGRAD_PIPE=4 timeout 300 python profiles/wgrad_ab.py save /tmp/wg1.pt
python profiles/wgrad_ab.py cmp /tmp/wg0.pt /tmp/wg1.pt
echo "--- timeline, old"; BSMS_WGRAD_PIPE=0 bash profiles/with_exp.sh timeout 300 python profiles/wgrad_timeline.py
echo "--- timeline, 8 waves"; BSMS_WGRAD_PIPE=4 bash profiles/with_exp.sh timeout 300 python profiles/wgrad_timeline.py
echo "--- timeline, 8 waves, 64 workgroups"; BSMS_WGRAD_WGS=64 BSMS_WGRAD_PIPE=4 bash profiles/with_exp.sh timeout 300 python profiles/wgrad_timeline.py
for r in 1 2; do
  rate BSMS_WGRAD_PIPE=0
  for w in 128 96 64 48; do rate BSMS_WGRAD_PIPE=4 BSMS_WGRAD_WGS=$w; done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/wgrad_ab.txt
