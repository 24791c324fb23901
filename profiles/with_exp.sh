#!/bin/bash
# run a command with the experiment build (lib_exp.so.keep) installed as libbsms_hip.so, restore the production library after
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
cp bsms-gnn_amd/libbsms_hip.so bsms-gnn_amd/lib_cur.so.keep
cp bsms-gnn_amd/lib_exp.so.keep bsms-gnn_amd/libbsms_hip.so
"$@"
rc=$?
cp bsms-gnn_amd/lib_cur.so.keep bsms-gnn_amd/libbsms_hip.so
exit $rc
