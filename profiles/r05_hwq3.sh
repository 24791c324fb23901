# same-box A/B of the hardware-queue probe for the lane streams: base = the library before (HEAD), cur = with create_lane_stream
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05q3
BENCH="--no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05q3/airfoil_f32.txt 2>&1
BENCH="--dtype bf16 --no-other-lines" BSMS_AB_DTYPE=bf16 bash profiles/ab_libs.sh base cur > gpurun_out/r05q3/airfoil_bf16.txt 2>&1
BENCH="--workload cylinder --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05q3/cyl.txt 2>&1
BENCH="--workload cylinder --layout blockdiag --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05q3/cyl_blockdiag.txt 2>&1
BENCH="--workload surface --batch 2 --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05q3/surface_f32.txt 2>&1
tail -n 5 gpurun_out/r05q3/*.txt
