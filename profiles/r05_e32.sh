# fp32 fused edge backward (efuse32.hip) against the unfused path, experiment build: per-parameter gradients + step rate
#   gpurun -- 'bash profiles/r05_e32.sh [workload] [batch]'
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
wl=${1:-cylinder}; B=${2:-1}
run() { bash profiles/with_exp.sh env "$@"; }
run BSMS_EDGE_FUSED_F32=0 timeout 300 python profiles/efuse32_ab.py save /tmp/e32_a.pt $wl $B 2>&1 | grep -v amdgpu.ids | tail -2
run BSMS_EDGE_FUSED_F32=1 timeout 300 python profiles/efuse32_ab.py save /tmp/e32_b.pt $wl $B 2>&1 | grep -v amdgpu.ids | tail -5
python profiles/efuse32_ab.py cmp /tmp/e32_a.pt /tmp/e32_b.pt 20 $KEYS
