#!/bin/bash
# round-4 second GPU call: dp / sanitizer tests, NT-load A/B of the step, cold roofline of both builds (fp32 + bf16)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
cp bsms-gnn_amd/libbsms_hip.so bsms-gnn_amd/lib_cur.so.keep
timeout 900 python -m pytest tests/test_hip_dp.py tests/test_host_sanitizers.py tests/test_hip_training.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r04b/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04b/tests.log
tail -n 4 gpurun_out/r04b/tests.log
bash profiles/ab.sh base nt > gpurun_out/r04b/ab_f32.txt 2>&1; cat gpurun_out/r04b/ab_f32.txt
BENCH_ARGS="--dtype bf16" bash profiles/ab.sh base nt > gpurun_out/r04b/ab_bf16.txt 2>&1; cat gpurun_out/r04b/ab_bf16.txt
for v in base nt; do
  cp bsms-gnn_amd/lib_$v.so.keep bsms-gnn_amd/libbsms_hip.so
  for dt in f32 bf16; do
    timeout 300 python bench.py --roofline-only --dtype $dt 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read())['roofline']
print('$v $dt', 'cold us', round(r['avg_us'],2), 'frac', round(r['frac'],3), 'warm', round(r['frac_warm'],3), 'copy', round(r['cold_device_copy']['frac_of_peak'],3))" | tee -a gpurun_out/r04b/roofline.txt
  done
done
cp bsms-gnn_amd/lib_cur.so.keep bsms-gnn_amd/libbsms_hip.so
