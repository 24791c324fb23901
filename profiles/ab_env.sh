#!/bin/bash
# Same-box A/B of one build under different environments (development switches of the library):
#   gpurun -- 'bash profiles/ab_env.sh "BSMS_X=0" "BSMS_X=1"'      ("-" = no variable)
# alternates the settings three times, 100 timed steps each, and prints steps/s.
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for v in "$@"; do
    printf "%s " "$v"
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],2))"
  done
done
