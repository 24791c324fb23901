#!/bin/bash
# same-box A/B on the batch-1 floor (training step + rollout).  Arguments are library builds (lib_<name>.so.keep) or, when they
# contain '=', environment settings of the current build:   bash profiles/ab_b1.sh A B     |     WL="airfoil 1" bash profiles/ab_b1.sh "X=0" "X=1"
cd "$(dirname "$0")/../bsms-gnn_amd"
cp libbsms_hip.so lib_cur.so.keep
for r in 1 2; do
  for v in "$@"; do
    printf "%-24s " "$v"
    case "$v" in
      *=*) (cd ..; env $v timeout 300 python profiles/b1_rates.py ${WL:-airfoil 1} 2>&1 | tail -1) ;;
      *) cp lib_$v.so.keep libbsms_hip.so; (cd ..; timeout 300 python profiles/b1_rates.py ${WL:-airfoil 1} 2>&1 | tail -1); cp lib_cur.so.keep libbsms_hip.so ;;
    esac
  done
done
