mkdir -p gpurun_out/r05r; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05r/pytest_all.txt
BENCH="--no-other-lines" bash profiles/ab_libs.sh efG efH exp > gpurun_out/r05r/ab_libs_f32.txt 2>&1
bash profiles/b1_rates.sh > gpurun_out/r05r/b1.txt 2>&1
