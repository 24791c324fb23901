# one scratch set per block of the backward (no waits for the lanes of the block before last): base = the library before   (gpurun -- 'bash profiles/r05_pbs.sh')
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05s
BENCH="--no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05s/airfoil_f32.txt 2>&1
BENCH="--dtype bf16 --no-other-lines" BSMS_AB_DTYPE=bf16 bash profiles/ab_libs.sh base cur > gpurun_out/r05s/airfoil_bf16.txt 2>&1
BENCH="--workload cylinder --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05s/cyl_f32.txt 2>&1
bash profiles/ab_b1_libs.sh base cur > gpurun_out/r05s/b1.txt 2>&1
tail -n 7 gpurun_out/r05s/*.txt
