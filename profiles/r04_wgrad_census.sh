#!/bin/bash
# what does a chunk step of the pipelined weight-gradient kernel consist of?  Components switched off one at a time
# (experiment build, BSMS_WGRAD_DBG bits: 1 HBM loads, 2 split, 4 MFMAs, 8 fragment reads, 16 LDS stores, 32 barrier)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-r04wc}; mkdir -p gpurun_out/$tag
{
for pipe in ${PIPES:-2}; do
for d in 0 1 2 4 8 16 32 3 6 7 12 15 31 47 63; do
  echo "=== PIPE=$pipe DBG=$d"
  BSMS_WGRAD_DBG=$d BSMS_WGRAD_PIPE=$pipe bash profiles/with_exp.sh timeout 120 python profiles/wgrad_timeline.py 2>&1 | grep -v amdgpu.ids | grep "median cycles"
done; done
} 2>&1 | tee gpurun_out/$tag/wgrad_census.txt
