#!/bin/bash
# kernel traces of the surface (16384 nodes, D=256, B=2) training step, fp32 and bf16.   gpurun -- 'bash profiles/r06_surface_prof.sh'
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
out=gpurun_out/surf; mkdir -p $out
for dt in f32 bf16; do
  rm -rf $out/prof_$dt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$dt -o r -- python bench.py --workload surface --batch 2 --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-lines > $out/prof_$dt.log 2>&1
  tail -c 400 $out/prof_$dt.log | head -c 200; echo
  python profiles/step_breakdown.py $out/prof_$dt/r_kernel_trace.csv | tee $out/breakdown_$dt.txt | head -45
done
