mkdir -p gpurun_out/r05h; cd /root/repo
timeout 600 python -m pytest tests/test_hip_bf16.py -m gpu -x -q -s -k "ragged" 2>&1 | grep "surf200\|passed\|failed\|Error\|assert" > gpurun_out/r05h/pytest_ragged.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05h/pytest_all.txt
