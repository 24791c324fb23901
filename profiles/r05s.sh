mkdir -p gpurun_out/r05s; cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16 or rollout or dp" 2>&1 | tail -3 > gpurun_out/r05s/pytest_bf16.txt
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efH exp > gpurun_out/r05s/ab_libs.txt 2>&1
bash profiles/kernel_time.sh k_edge_fused_bwd efH exp > gpurun_out/r05s/kt_bwd.txt 2>&1
bash profiles/kernel_time.sh k_edge_fwd_res efH exp > gpurun_out/r05s/kt_fwd.txt 2>&1
LIB=exp bash profiles/ef_timeline.sh 0 > gpurun_out/r05s/tl_exp.txt 2>&1
