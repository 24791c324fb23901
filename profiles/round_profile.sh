#!/bin/bash
# One GPU call that refreshes the tracked round summaries: default bench line, bf16 / surface / cylinder lines, and the
# kernel trace of the default step (summarised by profiles/summarize.py).   gpurun -- 'bash profiles/round_profile.sh r02'
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-rXX}
mkdir -p gpurun_out/$tag
timeout 600 python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/$tag/bench_bf16.json 2>/dev/null
timeout 300 python bench.py --workload surface --batch 2 --no-cpu-baseline > gpurun_out/$tag/bench_surface.json 2>/dev/null
timeout 300 python bench.py --workload surface --batch 2 --dtype bf16 --no-cpu-baseline > gpurun_out/$tag/bench_surface_bf16.json 2>/dev/null
timeout 300 python bench.py --dtype bf16_nodes --no-cpu-baseline > gpurun_out/$tag/bench_bf16_nodes.json 2>/dev/null
timeout 300 python bench.py --workload surface --batch 2 --dtype bf16_nodes --no-cpu-baseline > gpurun_out/$tag/bench_surface_bf16_nodes.json 2>/dev/null
timeout 300 python bench.py --workload cylinder --no-cpu-baseline > gpurun_out/$tag/bench_cyl.json 2>/dev/null
timeout 300 python bench.py --workload cylinder --layout blockdiag --no-cpu-baseline > gpurun_out/$tag/bench_cyl_blockdiag.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-lines > gpurun_out/$tag/prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_bf16 -o r -- python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-lines > gpurun_out/$tag/prof_bf16.log 2>&1
tail -n 1 gpurun_out/$tag/bench_*.json | cut -c1-400
