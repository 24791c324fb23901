# fp32 fused edge backward: gradients vs unfused, chain-wave timeline at level 0, step rates   (gpurun -- 'bash profiles/r05_e32e.sh')
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
bash profiles/r05_e32.sh airfoil 8 2>&1 | head -6 > gpurun_out/e32/grads_airfoil8.txt
BSMS_EDGE_FUSED_F32=1 bash profiles/with_exp.sh timeout 300 python profiles/ef32_timeline.py 0 2>&1 | grep -v "Warning\|amdgpu.ids" > gpurun_out/e32/timeline.txt
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_EDGE_FUSED_F32=0" "BSMS_EDGE_FUSED_F32=1" > gpurun_out/e32/ab.txt 2>&1
cat gpurun_out/e32/grads_airfoil8.txt gpurun_out/e32/timeline.txt gpurun_out/e32/ab.txt
