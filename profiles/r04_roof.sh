#!/bin/bash
# roofline of the graded kernel: PMC traffic + rocprofv3 stats of `bench.py --roofline-only`, and bench.py's own numbers (fp32, bf16)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-r04}; mkdir -p gpurun_out/${tag}n
bash profiles/agg_pmc.sh $tag > gpurun_out/${tag}n/agg_pmc.log 2>&1; tail -n 1 gpurun_out/${tag}n/agg_pmc.log | cut -c1-1400
for dt in f32 bf16; do
  python bench.py --roofline-only --dtype $dt 2>/dev/null > gpurun_out/${tag}n/roofline_$dt.json
  python - "$dt" gpurun_out/${tag}n/roofline_$dt.json <<'PY'
import json, sys
r = json.load(open(sys.argv[2]))["roofline"]
keys = ("avg_us", "frac", "frac_warm", "avg_us_event_pair_per_launch", "frac_event_pair_per_launch")
print(sys.argv[1], {k: round(r[k], 4) for k in keys}, "copy", round(r["cold_device_copy"]["frac_of_peak"], 3))
PY
done
