cd "$(dirname "$0")/.."
python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
BSMS_EDGE_NL=2 BSMS_CHAIN_NL=2 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
BSMS_EDGE_RB=2 BSMS_EDGE_CW=7 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
bash profiles/ab_env.sh "-" "BSMS_EDGE_NL=2" "BSMS_EDGE_RB=2 BSMS_EDGE_NL=2" "BSMS_EDGE_RB=2 BSMS_EDGE_CW=6 BSMS_EDGE_NL=2" "BSMS_EDGE_RB=2 BSMS_EDGE_CW=7 BSMS_EDGE_NL=1" "BSMS_CHAIN_NL=2" "BSMS_EDGE_NL=2 BSMS_CHAIN_NL=2"
