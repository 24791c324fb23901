cd "$(dirname "$0")/.."
python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
BSMS_RING_DEEP=3 BSMS_LONE_NL=1 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
bash profiles/ab_env.sh "BSMS_RING_DEEP=3 BSMS_LONE_NL=1" "BSMS_RING_DEEP=6 BSMS_LONE_NL=2" "BSMS_RING_DEEP=5 BSMS_LONE_NL=1" "BSMS_RING_DEEP=4 BSMS_LONE_NL=2" "BSMS_RING_DEEP=6 BSMS_LONE_NL=3"
