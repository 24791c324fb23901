# efuse32 variants (libraries lib_<name>.so.keep) under BSMS_EDGE_FUSED_F32=1: steps/s, alternating   (gpurun -- 'bash profiles/r05_e32f.sh exp prio3')
cd "$(dirname "$0")/../bsms-gnn_amd"; export TMPDIR=/tmp
cp libbsms_hip.so lib_cur.so.keep
for r in 1 2; do for v in "$@"; do cp lib_$v.so.keep libbsms_hip.so
  (cd ..; BSMS_EDGE_FUSED_F32=1 timeout 300 python bench.py --steps 80 --warmup 20 --no-cpu-baseline --no-roofline --no-other-lines 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), 'steps/s', round(d['ms_per_step'],3), 'ms')")
done; done
cp lib_cur.so.keep libbsms_hip.so
