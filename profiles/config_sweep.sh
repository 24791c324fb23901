#!/bin/bash
# Every workload x precision x batch x layout once (3 steps): does it launch, and at what rate.  bash profiles/config_sweep.sh
cd "$(dirname "$0")/.."
run() { printf "%-62s " "$*"; timeout 300 python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import sys, json
s = sys.stdin.read()
try: d = json.loads(s); print(round(d['value'], 1), 'steps/s  loss', round(d['config']['loss'], 5))
except Exception: print('FAILED:', s[-160:].strip())"; }
for dt in f32 bf16; do
  for b in 1 2 8 16; do run --workload airfoil --batch $b --dtype $dt; done
  for b in 1 8; do run --workload cylinder --batch $b --dtype $dt; run --workload cylinder --batch $b --dtype $dt --layout blockdiag; done
  for b in 1 2 4; do run --workload surface --batch $b --dtype $dt; done
done
