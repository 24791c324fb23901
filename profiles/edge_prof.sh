cd /root/repo
export TMPDIR=/tmp
cp bsms-gnn_amd/lib_D.so.keep bsms-gnn_amd/libbsms_hip.so
for m in 0 1 2; do
  BSMS_EDGE_RB=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/edge$m -o x -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/edge$m.log 2>&1
  f=$(find gpurun_out/edge$m -name "x_kernel_trace.csv" | head -1)
  echo "== RB mode $m"; python profiles/edge_trace.py $f 16
done
