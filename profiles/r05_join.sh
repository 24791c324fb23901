# two-stage join of the prepack lane (bsgmp.hip): surface B = 2 (D = 256) fp32 / bf16, airfoil fp32; base = the library before   (gpurun -- 'bash profiles/r05_join.sh')
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05j
BENCH="--workload surface --batch 2 --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05j/surface_f32.txt 2>&1
BENCH="--workload surface --batch 2 --dtype bf16 --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05j/surface_bf16.txt 2>&1
BENCH="--no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05j/airfoil_f32.txt 2>&1
BENCH="--workload cylinder --no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05j/cyl_f32.txt 2>&1
tail -n 5 gpurun_out/r05j/*.txt
