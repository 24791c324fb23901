#!/bin/bash
# HBM bytes and matrix-pipe busy cycles of the training step by kernel family (PMC passes on their own, as the guide
# prescribes):  gpurun -- 'bash profiles/step_pmc.sh'   -> gpurun_out/step_pmc/summary.txt      (DTYPE=bf16 for the bf16 step)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/step_pmc
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$tag -o x -- python bench.py ${WL} --dtype ${DTYPE:-f32} --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-lines > $out/$tag.log 2>&1
done
python - <<'PY' | tee gpurun_out/step_pmc/summary.txt
import csv, glob, re, collections
def load(tag):
    f = glob.glob(f"gpurun_out/step_pmc/{tag}/**/x_counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
short = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n)).split("(")[0][:40]
def per_step(rows, counter):
    # the last 4 steps: dispatches after the (n-4)th k_sim_prologue
    rows = [r for r in rows if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if "k_sim_prologue" in r["Kernel_Name"]]
    a = starts[-4]
    fam = collections.defaultdict(float)
    for r in rows[a:]:
        fam[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    steps = len(starts[-4:])
    return {k: v / steps for k, v in fam.items()}
fetch, write = per_step(load("FETCH_SIZE"), "FETCH_SIZE"), per_step(load("WRITE_SIZE"), "WRITE_SIZE")
tot_r = sum(fetch.values()) * 2 * 1024; tot_w = sum(write.values()) * 1024
print(f"HBM bytes per training step (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024, MI355X_MICROARCH.md): read {tot_r / 1e9:.2f} GB + write {tot_w / 1e9:.2f} GB = {(tot_r + tot_w) / 1e9:.2f} GB")
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0)))[:14]:
    print(f"   {k:42s} read {2 * fetch.get(k, 0) * 1024 / 1e9:6.2f} GB  write {write.get(k, 0) * 1024 / 1e9:6.2f} GB")
sq = load("SQ_VALU_MFMA_BUSY_CYCLES")
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"):
    d = per_step(sq, c)
    if d: print(c, {k: round(v) for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:6]})
PY
