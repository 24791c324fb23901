# kernel trace of the bf16 step of the current build:  bash profiles/prof_bf16.sh <tag> [dtype]  -> gpurun_out/p_<tag>
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
v=${1:-cur}; dt=${2:-bf16}
rm -rf gpurun_out/p_$v
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_$v -o r -- python bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-lines > gpurun_out/p_$v.log 2>&1
f=$(find gpurun_out/p_$v -name "r_kernel_trace.csv" | head -1)
python profiles/step_breakdown.py $f
for k in "k_edge_fused_bwd" "k_ef_reduce" "k_chain_fwd<8, 3, 0" "k_rowsum_pair" "k_rowsum_bf16in" "k_chain_fwd<8, 1, 0" "k_chain_bwd<8, 0, 2" "k_wgrad"; do python profiles/level_trace.py $f "$k" 13; done
