cd /root/repo; export TMPDIR=/tmp
rate() { d=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --dtype $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp-build $d $*', round(d['value'],1), round(d['ms_per_step'],3))"; }
for r in 1 2; do
for cw in 4 5 6 7; do rate bf16 BSMS_BFEDGE_CW=$cw; done
rate bf16 BSMS_RING=4; rate f32 BSMS_RING=3; rate f32 BSMS_RING=4
for cfg in "BSMS_EDGE_CW=7 BSMS_EDGE_NL=1" "BSMS_EDGE_CW=6 BSMS_EDGE_NL=2" "BSMS_EDGE_CW=6 BSMS_EDGE_NL=1" "BSMS_EDGE_CW=5 BSMS_EDGE_NL=2"; do rate f32 $cfg; done
done
