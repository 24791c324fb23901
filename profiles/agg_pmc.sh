#!/bin/bash
# HBM traffic of the graded L0 aggregation kernel from the PMC counters (separate passes, MI355X_MICROARCH.md), current build:
#   gpurun -- 'bash profiles/agg_pmc.sh r04'   -> profiles/aggregation_traffic.json, profiles/<tag>_aggregation_pmc.csv
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-rXX}; out=gpurun_out/pmc_$tag; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- python bench.py --roofline-only > $out/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o r -- python bench.py --roofline-only > $out/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o r -- python bench.py --roofline-only > $out/write.log 2>&1
python profiles/pmc_traffic.py $out/fetch/r_counter_collection.csv $out/write/r_counter_collection.csv $out/stats/r_kernel_stats.csv $tag | tee $out/traffic.json
cp profiles/aggregation_traffic.json profiles/${tag}_aggregation_pmc.csv $out/
cp $out/stats/r_kernel_stats.csv $out/${tag}_roofline_only_kernel_stats.csv
