#!/bin/bash
# edge MLP forward of small launches through the feature-split kernel (BSMS_FS_EDGE_ROWS): bit identity, batch-1 rates by threshold
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
{
for v in 0 40000; do BSMS_FS_EDGE_ROWS=$v timeout 300 python profiles/model_ab.py save /tmp/fe1_$v.pt airfoil 1 2>&1 | grep -v amdgpu | tail -1; BSMS_FS_EDGE_ROWS=$v timeout 300 python profiles/model_ab.py save /tmp/fec_$v.pt cylinder 1 2>&1 | grep -v amdgpu | tail -1; done
python profiles/model_ab.py cmp /tmp/fe1_0.pt /tmp/fe1_40000.pt; python profiles/model_ab.py cmp /tmp/fec_0.pt /tmp/fec_40000.pt
for r in 1 2; do for v in 0 4096 12288 17000 22000 40000; do echo -n "BSMS_FS_EDGE_ROWS=$v "; BSMS_FS_EDGE_ROWS=$v timeout 300 python profiles/b1_rates.py airfoil 1 2>&1 | tail -1; done; done
for v in 0 12288; do echo -n "BSMS_FS_EDGE_ROWS=$v "; BSMS_FS_EDGE_ROWS=$v timeout 300 python profiles/b1_rates.py cylinder 1 2>&1 | tail -1; done
} 2>&1 | tee gpurun_out/r04_fs_edge.txt
