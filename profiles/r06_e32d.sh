# which half of the bias-gradient path fails the d128p2 golden case: all rows exact (dbexact) / all rows weighted (dbweights) / ...
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
cp bsms-gnn_amd/libbsms_hip.so bsms-gnn_amd/lib_cur.so.keep
for n in "$@"; do
  cp bsms-gnn_amd/lib_$n.so.keep bsms-gnn_amd/libbsms_hip.so
  echo "=== $n"
  for k in 1 2; do BSMS_EDGE_FUSED_F32=1 timeout 600 python -m pytest "tests/test_hip_parity.py::test_gmp_golden" -m gpu -q -x 2>&1 | grep "assert 0\.\|AssertionError: (\|passed\|failed"; done
  BSMS_EDGE_FUSED_F32=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-other-lines 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps/s', round(d['value'],2))"
done
cp bsms-gnn_amd/lib_cur.so.keep bsms-gnn_amd/libbsms_hip.so
