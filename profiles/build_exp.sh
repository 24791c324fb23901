#!/bin/bash
# experiment build (in-kernel time stamps, BSMS_DEBUG_FLAGS) -> bsms-gnn_amd/lib_exp.so.keep, then the production build again
cd "$(dirname "$0")/.."
BSMS_EXPERIMENTS=1 python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
cp bsms-gnn_amd/libbsms_hip.so bsms-gnn_amd/lib_exp.so.keep
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
