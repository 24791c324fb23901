# efwd.hip variants: waves per workgroup (8 / 12 / 16) and a one-tile-ahead prefetch of the endpoint rows; bf16 airfoil B=8, same box
mkdir -p gpurun_out/r05f; cd /root/repo
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh exp fww8 fww8p fww12 fww12p > gpurun_out/r05f/ab_fw.txt 2>&1
