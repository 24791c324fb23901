# cylinder batch-1 rates of library builds, alternating   (gpurun -- 'bash profiles/r05_cylb1.sh prejoin base cur')
cd "$(dirname "$0")/../bsms-gnn_amd"; export TMPDIR=/tmp
cp libbsms_hip.so lib_cur.so.keep
for r in 1 2 3; do for v in "$@"; do cp lib_$v.so.keep libbsms_hip.so; echo -n "$v "; (cd ..; timeout 300 python profiles/b1_rates.py cylinder 1 2>&1 | tail -1); done; done
cp lib_cur.so.keep libbsms_hip.so
