# node-level weight gradients range-free (BF3), fp16 x 2 window widened to 2^-18, efuse prologue by LDS-DMA: tests, digests, same-box A/Bs
mkdir -p gpurun_out/r05e; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05e/pytest_all.txt
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "weight_gradient" 2>&1 | grep "wgrad\|passed\|failed" > gpurun_out/r05e/pytest_wgrad.txt
BSMS_NODE_BF3=0 bash profiles/with_exp.sh python profiles/efwd_ab.py 2>&1 | grep "level\|digest" > gpurun_out/r05e/digest_nodebf3_0.txt
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efA exp > gpurun_out/r05e/ab_libs_bf16.txt 2>&1
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_NODE_BF3=0" "BSMS_NODE_BF3=1" > gpurun_out/r05e/ab_node_bf3_f32.txt 2>&1
BENCH_ARGS="--dtype bf16 --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_NODE_BF3=0" "BSMS_NODE_BF3=1" > gpurun_out/r05e/ab_node_bf3_bf16.txt 2>&1
bash profiles/prof_bf16.sh r05e bf16 > gpurun_out/r05e/prof_bf16.txt 2>&1
