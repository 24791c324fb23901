mkdir -p gpurun_out/r05p; cd /root/repo; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05p/prof_surf -o r -- python bench.py --workload surface --batch 2 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-lines > gpurun_out/r05p/prof.log 2>&1
python profiles/step_breakdown.py $(find gpurun_out/r05p/prof_surf -name "r_kernel_trace.csv" | head -1) > gpurun_out/r05p/breakdown.txt 2>&1
