# kernel trace of the current build:  bash profiles/prof1.sh <tag>   -> gpurun_out/p_<tag>
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
v=${1:-cur}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_$v -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline ${BENCH_ARGS} > gpurun_out/p_$v.log 2>&1
python profiles/step_breakdown.py gpurun_out/p_$v/r_kernel_trace.csv
python profiles/gaps.py gpurun_out/p_$v/r_kernel_trace.csv 4 | head -12
