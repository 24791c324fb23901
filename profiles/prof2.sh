cd /root/repo; export TMPDIR=/tmp
for v in BASE H2; do
  cp bsms-gnn_amd/lib_$v.so.keep bsms-gnn_amd/libbsms_hip.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_$v -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/p_$v.log 2>&1
  f=$(find gpurun_out/p_$v -name "r_kernel_stats.csv" | head -1)
  echo "== $v"; head -16 $f | cut -c1-200
done
cp bsms-gnn_amd/lib_H2.so.keep bsms-gnn_amd/libbsms_hip.so
bash profiles/ab.sh BASE H2
