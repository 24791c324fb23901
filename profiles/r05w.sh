mkdir -p gpurun_out/r05w; cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16" 2>&1 | tail -3 > gpurun_out/r05w/pytest_bf16.txt
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efI exp > gpurun_out/r05w/ab_libs.txt 2>&1
bash profiles/kernel_time.sh k_rowsum_pair_fiber efI exp > gpurun_out/r05w/kt.txt 2>&1
BENCH="--workload surface --batch 2 --dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efI exp > gpurun_out/r05w/ab_libs_surf.txt 2>&1
