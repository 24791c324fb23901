# efuse.hip with the hand-over inside the gradient stages + deferred g0 stores, efwd.hip with the LDS-DMA prologue: digests, tests, same-box A/B, timeline
mkdir -p gpurun_out/r05c; cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16 or rollout" 2>&1 | tail -5 > gpurun_out/r05c/pytest_bf16.txt
for v in 0 1; do BSMS_EDGE_FWD_RES=$v bash profiles/with_exp.sh python profiles/efwd_ab.py 2>&1 | grep "level\|digest" > gpurun_out/r05c/digest_$v.txt; done
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efA exp > gpurun_out/r05c/ab_libs.txt 2>&1
bash profiles/ef_timeline.sh 0 4 > gpurun_out/r05c/ef_timeline.txt 2>&1
bash profiles/prof_bf16.sh r05c bf16 > gpurun_out/r05c/prof_bf16.txt 2>&1
