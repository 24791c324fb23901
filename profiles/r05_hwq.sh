# side lanes in their own hardware-queue pool (a non-default stream priority): experiment build, BSMS_LANE_PRIO = -1 (least) / 1 (greatest) / unset
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05q; export TMPDIR=/tmp
for v in "-" "BSMS_LANE_PRIO=-1" "BSMS_LANE_PRIO=1"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  echo "== $v"
  env $e bash profiles/with_exp.sh python profiles/experiments/blockdiag_context.py alone 2>&1 | grep -v amdgpu
  env $e bash profiles/with_exp.sh python profiles/experiments/blockdiag_context.py dense_first 2>&1 | grep -v amdgpu
done > gpurun_out/r05q/blockdiag.txt 2>&1
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_LANE_PRIO=-1" "BSMS_LANE_PRIO=1" > gpurun_out/r05q/airfoil_f32.txt 2>&1
BENCH_ARGS="--dtype bf16 --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_LANE_PRIO=-1" "BSMS_LANE_PRIO=1" > gpurun_out/r05q/airfoil_bf16.txt 2>&1
BENCH_ARGS="--workload cylinder --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_LANE_PRIO=-1" "BSMS_LANE_PRIO=1" > gpurun_out/r05q/cyl.txt 2>&1
BENCH_ARGS="--workload cylinder --layout blockdiag --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_LANE_PRIO=-1" "BSMS_LANE_PRIO=1" > gpurun_out/r05q/cyl_blockdiag.txt 2>&1
tail -n 12 gpurun_out/r05q/*.txt
