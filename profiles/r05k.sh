mkdir -p gpurun_out/r05k; cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16" 2>&1 | tail -3 > gpurun_out/r05k/pytest_bf16.txt
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efE exp > gpurun_out/r05k/ab_libs.txt 2>&1
bash profiles/kernel_time.sh k_edge_fused_bwd efE exp > gpurun_out/r05k/kt_bwd.txt 2>&1
LIB=exp bash profiles/ef_timeline.sh 0 > gpurun_out/r05k/tl_exp.txt 2>&1
