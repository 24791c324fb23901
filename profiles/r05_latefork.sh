# lane 1 (the block's D x D weight gradients) forked at the end of the block's tail instead of right behind the edge backward; experiment build
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05l
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_DEBUG_FLAGS=8192" > gpurun_out/r05l/f32.txt 2>&1
BENCH_ARGS="--dtype bf16 --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_DEBUG_FLAGS=8192" > gpurun_out/r05l/bf16.txt 2>&1
BENCH_ARGS="--workload cylinder --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_DEBUG_FLAGS=8192" > gpurun_out/r05l/cyl.txt 2>&1
tail -n 7 gpurun_out/r05l/*.txt
