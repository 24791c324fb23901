# bash profiles/ef_timeline.sh [levels...]: stamps of the fused edge backward with an experiment build (LIB=exp by default:
# bsms-gnn_amd/lib_$LIB.so.keep, built by profiles/build_efv.sh / build_exp.sh), production library restored after
cd "$(dirname "$0")/../bsms-gnn_amd"
cp libbsms_hip.so /tmp/lib_cur.so; cp lib_${LIB:-exp}.so.keep libbsms_hip.so
(cd ..; timeout 300 python profiles/ef_timeline.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids")
cp /tmp/lib_cur.so libbsms_hip.so
