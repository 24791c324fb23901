# per-kernel time of the fp32 step with the forward saving fp32 rows (BSMS_EDGE_FUSED_F32=0) / fp16 x 2 pieces (=1; experiment build)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
O=gpurun_out/e32
for f in 0 1; do
  rm -rf $O/kt$f
  BSMS_EDGE_FUSED_F32=$f bash profiles/with_exp.sh timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$f -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-lines > /dev/null 2>&1
  python - "$(find $O/kt$f -name 'r_kernel_trace.csv' | head -1)" $f <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
t = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("bsms::", "")
    k = re.sub(r"\((ChainFwdArgs|ChainBwdArgs|WgradTable|EdgeFused32Args|RowSumArgs).*", "", k)[:60]
    t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000; n[k] += 1
print(f"BSMS_EDGE_FUSED_F32={sys.argv[2]}: kernel time per step (us), launches per step")
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v / 25:9.1f}  {n[k] / 25:6.1f}  {k}")
PY
  rm -rf $O/kt$f
done
