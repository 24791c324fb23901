cd "$(dirname "$0")/.."
for cfg in "$@"; do echo "=== $cfg"; env $cfg timeout 300 python profiles/edge_timeline.py 2>&1 | grep -v "Warning\|amdgpu.ids"; done
