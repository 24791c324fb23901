# main-stream breakdown of the bf16 B=8 step, the f32 / bf16 B=1 steps   (gpurun -- 'bash profiles/r05x.sh')
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for cfg in "bf16b8:--dtype bf16" "bf16b1:--dtype bf16 --batch 1" "f32b1:--batch 1" "f32b8:"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  BENCH_ARGS="$args --no-other-lines" bash profiles/prof1.sh x_$tag > gpurun_out/x_$tag.txt 2>&1
  python profiles/gaps.py gpurun_out/p_x_$tag/r_kernel_trace.csv 2 >> gpurun_out/x_$tag.txt 2>&1
  rm -rf gpurun_out/p_x_$tag/*.db
done
tail -n 60 gpurun_out/x_*.txt
