mkdir -p gpurun_out/r05a; cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16" 2>&1 | tail -5 > gpurun_out/r05a/pytest_bf16.txt
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r05a/bench_f32.json 2> gpurun_out/r05a/bench_f32.err
timeout 300 python bench.py --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-other-lines > gpurun_out/r05a/bench_bf16.json 2> gpurun_out/r05a/bench_bf16.err
bash profiles/prof_bf16.sh r05a bf16 > gpurun_out/r05a/prof_bf16.txt 2>&1
bash profiles/prof_bf16.sh r05a_f32 f32 > gpurun_out/r05a/prof_f32.txt 2>&1
bash profiles/ef_timeline.sh 0 2 4 > gpurun_out/r05a/ef_timeline.txt 2>&1
