cd /root/repo
for m in 1 2; do BSMS_EDGE_RB=$m timeout 300 python profiles/edge_timeline.py 2>&1 | grep -v Warning; done
