#!/bin/bash
# same-box A/B of library builds (bsms-gnn_amd/lib_<name>.so.keep; "cur" = the library in the tree): bit identity of one
# training step against the first build, then steps/s of bench.py, alternating, two rounds
#   bash profiles/ab_libs.sh base cur        BENCH="--dtype bf16" bash profiles/ab_libs.sh base cur
cd "$(dirname "$0")/../bsms-gnn_amd"; export TMPDIR=/tmp
cp libbsms_hip.so lib_cur.so.keep
use() { cp lib_$1.so.keep libbsms_hip.so; }
first=$1
for v in "$@"; do use $v; (cd ..; timeout 300 python profiles/model_ab.py save /tmp/ab_$v.pt 2>&1 | grep -v amdgpu.ids | tail -1); done
for v in "$@"; do [ $v != $first ] && (cd ..; echo -n "$first vs $v: "; python profiles/model_ab.py cmp /tmp/ab_$first.pt /tmp/ab_$v.pt); done
for r in 1 2; do for v in "$@"; do use $v
  (cd ..; timeout 300 python bench.py --steps 80 --warmup 20 --no-cpu-baseline --no-roofline $BENCH 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), 'steps/s', round(d['ms_per_step'],3), 'ms; rollout', round(d.get('rollout',{}).get('eager',0),1))")
done; done
use cur
