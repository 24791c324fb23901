mkdir -p gpurun_out/r05i; cd /root/repo
bash profiles/kernel_time.sh k_edge_fwd_res exp fwst1 fwst2 fwst4 exp > gpurun_out/r05i/fw_stagger.txt 2>&1
