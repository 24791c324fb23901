# production library after the chain_dev.h split / efuse32 addition against the library of the commit before: bit identity + rates
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05p
BENCH="--no-other-lines" bash profiles/ab_libs.sh base cur > gpurun_out/r05p/f32.txt 2>&1
BENCH="--dtype bf16 --no-other-lines" BSMS_AB_DTYPE=bf16 bash profiles/ab_libs.sh base cur > gpurun_out/r05p/bf16.txt 2>&1
bash profiles/ab_b1_libs.sh base cur > gpurun_out/r05p/b1.txt 2>&1
tail -n 8 gpurun_out/r05p/*.txt
