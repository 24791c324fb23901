# bash profiles/ef_timeline.sh [levels...]: stamps of the fused edge backward with the experiment build shipped as
# bsms-gnn_amd/lib_exp.so.keep (built locally: BSMS_EXPERIMENTS=1 python bsms-gnn_amd/build.py --force), production library restored after
cd "$(dirname "$0")/../bsms-gnn_amd"
cp libbsms_hip.so /tmp/lib_cur.so; cp lib_exp.so.keep libbsms_hip.so
(cd ..; timeout 300 python profiles/ef_timeline.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids")
cp /tmp/lib_cur.so libbsms_hip.so
