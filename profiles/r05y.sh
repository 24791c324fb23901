# upper bound of what a fused fp32 edge backward could give: the fp32 step WITHOUT the traffic such a kernel would not have
# (bit 0: the forward stores no a_0..a_2; bit 10: no edge-level weight-gradient jobs; bit 11: k_edge_bwd streams no gE[1..3])
# -- the gradients are wrong, the timing is the point.   gpurun -- 'bash profiles/r05y.sh'
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05y
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "-" "BSMS_DEBUG_FLAGS=1" "BSMS_DEBUG_FLAGS=1024" "BSMS_DEBUG_FLAGS=2048" "BSMS_DEBUG_FLAGS=3073" > gpurun_out/r05y/ab.txt 2>&1
cat gpurun_out/r05y/ab.txt
