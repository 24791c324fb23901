mkdir -p gpurun_out/r05u; cd /root/repo
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh exp fwrb1 fwrb2 > gpurun_out/r05u/ab_libs.txt 2>&1
bash profiles/kernel_time.sh k_edge_fwd_res exp fwrb1 fwrb2 > gpurun_out/r05u/kt_fwd.txt 2>&1
