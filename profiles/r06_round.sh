#!/bin/bash
# round-6 evidence run: full GPU test suite, bench lines of every BASELINE configuration, kernel traces (fp32 + bf16), PMC traffic of the
# graded kernel and of both steps, batch-1 rates.    gpurun -- 'bash profiles/r06_round.sh r06'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-r06}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/$tag/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/$tag/tests.log
tail -n 3 gpurun_out/$tag/tests.log
bash profiles/round_profile.sh $tag
bash profiles/agg_pmc.sh $tag > gpurun_out/$tag/agg_pmc.log 2>&1; tail -n 1 gpurun_out/$tag/agg_pmc.log | cut -c1-300
bash profiles/step_pmc.sh > /dev/null 2>&1; cp gpurun_out/step_pmc/summary.txt gpurun_out/$tag/step_pmc_f32.txt
DTYPE=bf16 bash profiles/step_pmc.sh > /dev/null 2>&1; cp gpurun_out/step_pmc/summary.txt gpurun_out/$tag/step_pmc_bf16.txt
python profiles/b1_rates.py airfoil 1 2>&1 | tail -1 | tee gpurun_out/$tag/b1_rates.txt
python profiles/b1_rates.py cylinder 1 2>&1 | tail -1 | tee -a gpurun_out/$tag/b1_rates.txt
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-other-lines > gpurun_out/$tag/bench_gpus2.json 2> gpurun_out/$tag/bench_gpus2.err; tail -c 300 gpurun_out/$tag/bench_gpus2.json
timeout 600 python profiles/fresh_mesh.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/$tag/fresh_mesh.txt; tail -4 gpurun_out/$tag/fresh_mesh.txt
