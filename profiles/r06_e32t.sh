# fused fp32 edge backward (round 6), parity record: the GPU parity suites with the experiment library and BSMS_EDGE_FUSED_F32=1
# (golden fixtures at 1e-5, full-size three-way criterion against fp64)   gpurun -- 'bash profiles/r06_e32t.sh'
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
BSMS_EDGE_FUSED_F32=1 bash profiles/with_exp.sh timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_fullsize.py -m gpu -q -s -x 2>&1 | grep -v amdgpu.ids > gpurun_out/e32/tests_fused32.txt
grep -n "ratios gpu\|grads vs fp64\|passed\|failed\|Error\|^\[" gpurun_out/e32/tests_fused32.txt | head -60
