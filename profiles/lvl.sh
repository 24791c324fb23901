# per-level times of the node-level chain kernels of a build:  bash profiles/lvl.sh <libtag>
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
cp bsms-gnn_amd/lib_$1.so.keep bsms-gnn_amd/libbsms_hip.so
rm -rf gpurun_out/lv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lv -o x -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/lv.log 2>&1
f=$(find gpurun_out/lv -name "x_kernel_trace.csv" | head -1)
echo "== $1"
for k in "k_chain_fwd<8, 1, 0" "k_chain_bwd<8, 0, 2" "k_chain_fwd<8, 1, 1" "k_chain_fwd<8, 0, 3" "k_edge_fwd" "k_edge_bwd"; do python profiles/level_trace.py $f "$k" 16; done
