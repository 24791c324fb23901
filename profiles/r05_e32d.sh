# fp32 fused edge backward: gradients vs unfused, step rates, kernel trace   (gpurun -- 'bash profiles/r05_e32d.sh')
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/e32; export TMPDIR=/tmp
bash profiles/r05_e32.sh airfoil 8 > gpurun_out/e32/grads_airfoil8.txt 2>&1
BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/ab_env.sh "BSMS_EDGE_FUSED_F32=0" "BSMS_EDGE_FUSED_F32=1" > gpurun_out/e32/ab.txt 2>&1
BSMS_EDGE_FUSED_F32=1 BENCH_ARGS="--no-other-lines" bash profiles/with_exp.sh bash profiles/prof1.sh e32 > gpurun_out/e32/prof.txt 2>&1
python /dev/stdin gpurun_out/p_e32/r_kernel_trace.csv > gpurun_out/e32/levels.txt 2>&1 <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_sim_prologue" in r["Kernel_Name"]]
seg = rows[idx[-3]:idx[-2]]
for r in seg:
    if "fused32" in r["Kernel_Name"] or "k_edge_fwd" in r["Kernel_Name"]:
        print(r["Kernel_Name"].split("(")[0][-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
rm -rf gpurun_out/p_e32/*.db
head -8 gpurun_out/e32/grads_airfoil8.txt; cat gpurun_out/e32/ab.txt; head -30 gpurun_out/e32/prof.txt; cat gpurun_out/e32/levels.txt
