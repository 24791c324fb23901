# fused fp32 edge backward (round 6): census of variants -- chain-wave timeline at level 0 and the step rate for each library
#   gpurun -- 'bash profiles/r06_e32v.sh name1 name2 ...'   (libraries bsms-gnn_amd/lib_<name>.so.keep, built by profiles/build_efv.sh)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
cp bsms-gnn_amd/libbsms_hip.so bsms-gnn_amd/lib_cur.so.keep
for n in "$@"; do
  cp bsms-gnn_amd/lib_$n.so.keep bsms-gnn_amd/libbsms_hip.so
  echo "=== $n"
  BSMS_EDGE_FUSED_F32=1 timeout 300 python profiles/ef32_timeline.py 0 2>&1 | grep -v "Warning\|amdgpu.ids"
  BSMS_EDGE_FUSED_F32=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-other-lines 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps/s', round(d['value'],2))"
done
cp bsms-gnn_amd/lib_cur.so.keep bsms-gnn_amd/libbsms_hip.so
