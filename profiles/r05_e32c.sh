# debugging: bias gradients of the fused fp32 kernel, fragment-based (lib exp) against staged-row based (lib ldsb)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
cp bsms-gnn_amd/libbsms_hip.so bsms-gnn_amd/lib_cur.so.keep
cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
BSMS_EDGE_FUSED_F32=0 python profiles/efuse32_ab.py save /tmp/a.pt airfoil 8 2>&1 | tail -1
BSMS_EDGE_FUSED_F32=1 python profiles/efuse32_ab.py save /tmp/b.pt airfoil 8 2>&1 | tail -1
cp bsms-gnn_amd/lib_ldsb.so.keep bsms-gnn_amd/libbsms_hip.so
BSMS_EDGE_FUSED_F32=1 python profiles/efuse32_ab.py save /tmp/c.pt airfoil 8 2>&1 | tail -1
cp bsms-gnn_amd/lib_cur.so.keep bsms-gnn_amd/libbsms_hip.so
echo "unfused vs fragments"; python profiles/efuse32_ab.py cmp /tmp/a.pt /tmp/b.pt 4
echo "unfused vs staged rows"; python profiles/efuse32_ab.py cmp /tmp/a.pt /tmp/c.pt 4
echo "fragments vs staged rows"; python profiles/efuse32_ab.py cmp /tmp/b.pt /tmp/c.pt 4
