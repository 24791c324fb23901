# bash profiles/kernel_time.sh <kernel-substring> <lib...>: per-step time of one kernel family in the bf16 step, for several experiment libraries
# (bsms-gnn_amd/lib_<name>.so.keep), from rocprofv3 kernel traces on the same box
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
k=$1; shift
cp bsms-gnn_amd/libbsms_hip.so /tmp/lib_cur.so
for v in "$@"; do
  cp bsms-gnn_amd/lib_$v.so.keep bsms-gnn_amd/libbsms_hip.so
  rm -rf gpurun_out/kt_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$v -o r -- python bench.py --dtype ${DTYPE:-bf16} --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-lines > /dev/null 2>&1
  f=$(find gpurun_out/kt_$v -name "r_kernel_trace.csv" | head -1)
  python - "$f" "$k" "$v" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows]
n = len(d) // 25 if len(d) >= 25 else 1      # launches per step (25 steps traced)
last = d[-20 * n:]
print(f"{sys.argv[3]:10s} {sys.argv[2]}: {sum(last) / 20:8.1f} us per step over {n} launches; largest launch {max(last):6.1f} us, smallest {min(last):6.1f} us")
PY
done
cp /tmp/lib_cur.so bsms-gnn_amd/libbsms_hip.so
