import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows
     if "k_edge_fwd" in r["Kernel_Name"] or "k_chain_fwd<8, 3" in r["Kernel_Name"]]
assert len(d) == 46, len(d)
lab = ["airfoil train", "airfoil infer", "local train", "local infer"]
for i, l in enumerate(lab):
    base = 23 * (i // 2) + 3 + 10 * (i % 2)
    v = d[base:base + 10]
    print(f"  {l:14s} avg {sum(v) / 10:7.1f} us  min {min(v):7.1f}")
