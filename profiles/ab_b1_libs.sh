#!/bin/bash
# bit identity (batch 8 and batch 1 training step) + batch-1 rates of two library builds:  bash profiles/ab_b1_libs.sh base cur
cd "$(dirname "$0")/../bsms-gnn_amd"; export TMPDIR=/tmp
cp libbsms_hip.so lib_cur.so.keep
use() { cp lib_$1.so.keep libbsms_hip.so; }
for v in "$@"; do use $v; (cd ..; timeout 300 python profiles/model_ab.py save /tmp/ab8_$v.pt 2>&1 | grep -v amdgpu | tail -1; timeout 300 python profiles/model_ab.py save /tmp/ab1_$v.pt airfoil 1 2>&1 | grep -v amdgpu | tail -1); done
(cd ..; python profiles/model_ab.py cmp /tmp/ab8_$1.pt /tmp/ab8_$2.pt; python profiles/model_ab.py cmp /tmp/ab1_$1.pt /tmp/ab1_$2.pt)
for r in 1 2 3; do for v in "$@"; do use $v; echo -n "$v "; (cd ..; timeout 300 python profiles/b1_rates.py airfoil 1 2>&1 | tail -1); done; done
use cur
