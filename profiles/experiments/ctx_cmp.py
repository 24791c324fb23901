"""per-kernel time of the last 25 steps of two kernel traces:  python ctx_cmp.py a/r_kernel_trace.csv b/r_kernel_trace.csv"""
import csv, re, sys, collections
def load(p):
    rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_sim_prologue" in r["Kernel_Name"]]
    seg = rows[idx[-26]:idx[-1]]
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        n = re.sub(r"^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0]
        out[n][0] += 1; out[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(rows[idx[-1]]["Start_Timestamp"]) - int(rows[idx[-26]]["Start_Timestamp"])) / 25e3
    return {k: (v[0] / 25, v[1] / 25) for k, v in out.items()}, span
a, sa = load(sys.argv[1]); b, sb = load(sys.argv[2])
print(f"step {sa:.1f} us vs {sb:.1f} us")
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    rows.append((ta - tb, k, ca, ta, cb, tb))
for d, k, ca, ta, cb, tb in sorted(rows, reverse=True)[:14] + sorted(rows)[:6]:
    print(f"{d:+8.1f} us  {k:56s} {ca:5.1f} x {ta:8.1f}   {cb:5.1f} x {tb:8.1f}")
