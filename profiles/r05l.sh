mkdir -p gpurun_out/r05l; cd /root/repo
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efF exp > gpurun_out/r05l/ab_libs.txt 2>&1
bash profiles/kernel_time.sh k_edge_fused_bwd efF exp > gpurun_out/r05l/kt_bwd.txt 2>&1
LIB=exp bash profiles/ef_timeline.sh 0 > gpurun_out/r05l/tl_exp.txt 2>&1
