# kernel summaries of the surface (16k nodes, 6 levels, D = 256) B = 2 steps, bf16 and fp32   (gpurun -- 'bash profiles/r05s2.sh')
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for cfg in "surf_bf16:--workload surface --batch 2 --dtype bf16" "surf_f32:--workload surface --batch 2"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  BENCH_ARGS="$args --no-other-lines" bash profiles/prof1.sh s_$tag > gpurun_out/s_$tag.txt 2>&1
  head -40 gpurun_out/p_s_$tag/r_kernel_stats.csv > gpurun_out/s_${tag}_stats.csv
  tail -2 gpurun_out/p_s_$tag.log >> gpurun_out/s_$tag.txt
  rm -rf gpurun_out/p_s_$tag/*.db
done
cat gpurun_out/s_surf_bf16.txt gpurun_out/s_surf_f32.txt
