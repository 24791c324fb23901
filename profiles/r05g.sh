# efuse.hip: the chain waves' loads spread between the MFMA chunks.  bit identity + same-box A/B against the previous build (efE), timeline
mkdir -p gpurun_out/r05g; cd /root/repo
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh efE exp > gpurun_out/r05g/ab_libs.txt 2>&1
bash profiles/build_nothing.sh 2>/dev/null
LIB=exp bash profiles/ef_timeline.sh 0 > gpurun_out/r05g/tl_exp.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16" 2>&1 | tail -3 > gpurun_out/r05g/pytest_bf16.txt
