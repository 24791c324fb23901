# fp32 fused edge backward: step rates, experiment build, alternating   (gpurun -- 'bash profiles/r05_e32b.sh')
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
bash profiles/r05_e32.sh airfoil 8 > gpurun_out/e32/grads_airfoil8.txt 2>&1
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_EDGE_FUSED_F32=0" "BSMS_EDGE_FUSED_F32=1" > gpurun_out/e32/ab.txt 2>&1
cat gpurun_out/e32/grads_airfoil8.txt | head -12; cat gpurun_out/e32/ab.txt
