# side lanes restricted to a subset of the CUs (hipExtStreamCreateWithCUMask), experiment build
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05m
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_LANE_CUMASK=1" "BSMS_LANE_CUMASK=2" "BSMS_LANE_CUMASK=3" "BSMS_LANE_CUMASK=4" > gpurun_out/r05m/f32.txt 2>&1
BENCH_ARGS="--dtype bf16 --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_LANE_CUMASK=1" "BSMS_LANE_CUMASK=4" > gpurun_out/r05m/bf16.txt 2>&1
cat gpurun_out/r05m/f32.txt gpurun_out/r05m/bf16.txt
