mkdir -p gpurun_out/r05j; cd /root/repo
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efE exp > gpurun_out/r05j/ab_libs.txt 2>&1
bash profiles/kernel_time.sh k_edge_fwd_res efE exp > gpurun_out/r05j/kt_fwd.txt 2>&1
bash profiles/kernel_time.sh k_edge_fused_bwd efE exp > gpurun_out/r05j/kt_bwd.txt 2>&1
