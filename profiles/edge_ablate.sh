cd /root/repo
export TMPDIR=/tmp
for cfg in "BSMS_EDGE_RB=1 BSMS_EDGE_DELAY=0" "BSMS_EDGE_RB=1 BSMS_EDGE_DELAY=100" "BSMS_EDGE_RB=1 BSMS_EDGE_DELAY=200" "BSMS_EDGE_RB=1 BSMS_EDGE_DELAY=300" "BSMS_EDGE_RB=1 BSMS_EDGE_DELAY=400"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl_$tag -o x -- python profiles/edge_ablate.py > gpurun_out/abl_$tag.log 2>&1
  f=$(find gpurun_out/abl_$tag -name "x_kernel_trace.csv" | head -1)
  echo "== $cfg"; python profiles/edge_ablate_read.py $f || tail -5 gpurun_out/abl_$tag.log
done
