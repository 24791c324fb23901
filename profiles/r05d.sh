# efuse variants (stamps of the gradient wave off / prefetch after the first gradient stage), BF3 weight gradients A/B on the f32 step
mkdir -p gpurun_out/r05d; cd /root/repo
for v in exp efB efC efD; do LIB=$v bash profiles/ef_timeline.sh 0 > gpurun_out/r05d/tl_$v.txt 2>&1; done
BENCH="--dtype bf16 --no-other-lines" bash profiles/ab_libs.sh exp efC > gpurun_out/r05d/ab_efC.txt 2>&1
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_WGRAD_BF3=0" "BSMS_WGRAD_BF3=1" > gpurun_out/r05d/ab_bf3.txt 2>&1
