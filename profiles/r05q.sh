mkdir -p gpurun_out/r05q; cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05q/pytest.txt
BENCH="--no-other-lines" bash profiles/ab_libs.sh efG exp > gpurun_out/r05q/ab_libs_f32.txt 2>&1
