# lane streams probed for their own hardware queue (plan.hip: create_lane_stream): block-diagonal cylinder in three process contexts +
# the standalone bench lines, production library
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05q2; export TMPDIR=/tmp
for m in alone dense_first twice pool_first; do python profiles/experiments/blockdiag_context.py $m 2>&1 | grep -v amdgpu; done > gpurun_out/r05q2/contexts.txt 2>&1
for a in "--workload cylinder --layout blockdiag" "--workload cylinder" "" "--dtype bf16" "--workload surface --batch 2 --dtype bf16"; do
  python bench.py $a --no-cpu-baseline --no-roofline --no-other-lines 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a:', round(d['value'],1), 'steps/s', round(d['ms_per_step'],3), 'ms')"
done > gpurun_out/r05q2/bench.txt 2>&1
python profiles/b1_rates.py airfoil 1 2>&1 | tail -1 >> gpurun_out/r05q2/bench.txt
python profiles/b1_rates.py cylinder 1 2>&1 | tail -1 >> gpurun_out/r05q2/bench.txt
cat gpurun_out/r05q2/contexts.txt gpurun_out/r05q2/bench.txt
