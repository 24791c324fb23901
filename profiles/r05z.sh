# host order of the fork points (caller's stream first): bit identity + batch-1 rates + cylinder / airfoil batch-8 steps   (gpurun -- 'bash profiles/r05z.sh')
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05z
bash profiles/ab_b1_libs.sh base cur > gpurun_out/r05z/b1.txt 2>&1
BENCH="--workload cylinder --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05z/cyl.txt 2>&1
BENCH="--no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05z/f32.txt 2>&1
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05z/bf16.txt 2>&1
tail -n 12 gpurun_out/r05z/*.txt
