# fused fp32 edge backward (round 6): gradients vs unfused, census of variants, same-box A/B, per-kernel time of both steps
#   gpurun -- 'bash profiles/r06_e32w.sh'     (needs lib_exp + variant libraries: profiles/build_efv.sh)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
O=gpurun_out/e32
bash profiles/r05_e32.sh airfoil 8 2>&1 | head -8 > $O/grads_air8.txt
bash profiles/r05_e32.sh cylinder 1 2>&1 | head -6 > $O/grads_cyl1.txt
bash profiles/r06_e32v.sh ${VARIANTS:-exp nodw noa nohand nothing} > $O/census.txt 2>&1
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_EDGE_FUSED_F32=0" "BSMS_EDGE_FUSED_F32=1" > $O/ab.txt 2>&1
for f in 0 1; do
  rm -rf $O/kt$f
  BSMS_EDGE_FUSED_F32=$f bash profiles/with_exp.sh timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$f -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-lines > /dev/null 2>&1
  python - "$(find $O/kt$f -name 'r_kernel_trace.csv' | head -1)" $f <<'PY' > $O/kernels$f.txt
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
t = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    k = re.sub(r"\(.*", "", r["Kernel_Name"]); k = re.sub(r"^void (\(anonymous namespace\)::|bsms::)?", "", k)[:70]
    t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000; n[k] += 1
steps = 25
print(f"BSMS_EDGE_FUSED_F32={sys.argv[2]}: kernel time per step (us), launches per step")
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:22]:
    print(f"  {v / steps:9.1f}  {n[k] / steps:6.1f}  {k}")
PY
  rm -rf $O/kt$f
done
cat $O/grads_air8.txt $O/grads_cyl1.txt $O/census.txt $O/ab.txt $O/kernels0.txt $O/kernels1.txt
