mkdir -p gpurun_out/r05n; cd /root/repo
BENCH_ARGS="--workload cylinder --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_NODE_BF3=0" "BSMS_NODE_BF3=1" > gpurun_out/r05n/ab_cyl.txt 2>&1
BENCH_ARGS="--workload cylinder --layout blockdiag --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_NODE_BF3=0" "BSMS_NODE_BF3=1" > gpurun_out/r05n/ab_cyl_bd.txt 2>&1
timeout 1200 python -m pytest tests/test_hip_fullsize.py -m gpu -q -s 2>&1 | grep -A4 "^\[" > gpurun_out/r05n/fullsize.txt
