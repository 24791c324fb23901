#!/bin/bash
# round-4 evidence run: full GPU test suite, bench lines of every BASELINE configuration, kernel traces (fp32 + bf16),
# PMC traffic of the graded kernel, batch-1 rates.    gpurun -- 'bash profiles/r04_round.sh r04'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-r04}
mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/$tag/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/$tag/tests.log
tail -n 3 gpurun_out/$tag/tests.log
bash profiles/round_profile.sh $tag
bash profiles/agg_pmc.sh $tag > gpurun_out/$tag/agg_pmc.log 2>&1; tail -n 1 gpurun_out/$tag/agg_pmc.log | cut -c1-300
python profiles/b1_rates.py airfoil 1 2>&1 | tail -1 | tee gpurun_out/$tag/b1_rates.txt
python profiles/b1_rates.py cylinder 1 2>&1 | tail -1 | tee -a gpurun_out/$tag/b1_rates.txt
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/$tag/bench_gpus2.json 2> gpurun_out/$tag/bench_gpus2.err; tail -c 400 gpurun_out/$tag/bench_gpus2.json
