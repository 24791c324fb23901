mkdir -p gpurun_out/r05t; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05t/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05t/smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/r05t/bench_default.json 2> gpurun_out/r05t/bench_default.err
