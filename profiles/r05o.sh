mkdir -p gpurun_out/r05o; cd /root/repo
BENCH_ARGS="--workload cylinder --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_NODE_BF3=0" "BSMS_NODE_BF3=1" "BSMS_NODE_BF3=2" > gpurun_out/r05o/ab_cyl.txt 2>&1
BENCH_ARGS="--workload cylinder --layout blockdiag --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_NODE_BF3=0" "BSMS_NODE_BF3=1" "BSMS_NODE_BF3=2" > gpurun_out/r05o/ab_cyl_bd.txt 2>&1
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_NODE_BF3=0" "BSMS_NODE_BF3=1" > gpurun_out/r05o/ab_air.txt 2>&1
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "weight_gradient" 2>&1 | grep "wgrad\|passed\|failed" > gpurun_out/r05o/pytest_wgrad.txt
