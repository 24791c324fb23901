mkdir -p gpurun_out/r05m; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05m/pytest_all.txt
bash profiles/round_profile.sh r05m > gpurun_out/r05m/round.txt 2>&1
