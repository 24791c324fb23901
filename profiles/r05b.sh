# resident-weights bf16 edge forward (efwd.hip): bit-identity against the ring kernel + same-box A/B (experiment build), then tests and bench
mkdir -p gpurun_out/r05b; cd /root/repo
for v in 0 1; do BSMS_EDGE_FWD_RES=$v bash profiles/with_exp.sh python profiles/efwd_ab.py 2>&1 | grep -v "Warning\|amdgpu.ids" > gpurun_out/r05b/digest_$v.txt; done
BENCH_ARGS="--dtype bf16 --no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_EDGE_FWD_RES=0" "BSMS_EDGE_FWD_RES=1" > gpurun_out/r05b/ab_bf16.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16 or rollout" 2>&1 | tail -5 > gpurun_out/r05b/pytest_bf16.txt
bash profiles/prof_bf16.sh r05b bf16 > gpurun_out/r05b/prof_bf16.txt 2>&1
