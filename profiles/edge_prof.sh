# per-level times of the edge-MLP chain kernels under different switches:  bash profiles/edge_prof.sh "BSMS_EDGE32=0" "BSMS_EDGE32=1"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
k=0
for cfg in "$@"; do
  k=$((k+1))
  env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/edgep$k -o x -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/edgep$k.log 2>&1
  f=$(find gpurun_out/edgep$k -name "x_kernel_trace.csv" | head -1)
  echo "== $cfg"; python profiles/edge_trace.py $f 16
done
