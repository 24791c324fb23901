cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for cfg in "BSMS_CHAIN_NL=1" "BSMS_CHAIN_NL=2" "BSMS_CHAIN_NL=3"; do
  rm -rf gpurun_out/nl
  env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nl -o x -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/nl.log 2>&1
  f=$(find gpurun_out/nl -name "x_kernel_trace.csv" | head -1)
  echo "== $cfg"
  for k in "k_chain_fwd<8, 1, 0" "k_chain_bwd<8, 0, 2" "k_chain_fwd<8, 1, 1" "k_chain_fwd<8, 0, 3"; do python profiles/level_trace.py $f "$k" 16; done
done
