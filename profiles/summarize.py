#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` run into the small text summaries kept in
profiles/ (the raw traces stay in gpurun_out/, which is scratch).

  python profiles/summarize.py gpurun_out/prof2/r2 profiles/r01_step   [steps_profiled]
writes <out>_kernel_stats.csv (verbatim copy of rocprofv3's per-kernel stats) and <out>_summary.md
(per-kernel table in ms/step + the aggregation kernel split by launch geometry = mesh level)."""
import csv
import re
import shutil
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 100 else name[:97] + "..."


def main():
    prefix, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 13
    shutil.copy(prefix + "_kernel_stats.csv", out + "_kernel_stats.csv")
    rows = list(csv.DictReader(open(prefix + "_kernel_stats.csv")))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    lines = [f"# rocprofv3 kernel summary ({prefix.split('/')[-1]}, {steps} steps incl. warm-up)", "",
             f"total kernel time {total / 1e6:.2f} ms = {total / 1e6 / steps:.2f} ms/step", "",
             "| kernel | calls/step | avg us | ms/step | % |", "|---|---|---|---|---|"]
    for r in rows[:28]:
        t = int(r["TotalDurationNs"])
        lines.append(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                     f"{t / 1e6 / steps:.3f} | {100 * t / total:.1f} |")
    # per-geometry breakdown of the plan-order segment sum (edge aggregation)
    geo = defaultdict(list)
    for r in csv.DictReader(open(prefix + "_kernel_trace.csv")):
        if "k_rowsum_v4<32, false, false>" in r["Kernel_Name"]:
            geo[int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines += ["", "## `k_rowsum_v4<32,false,false>` by launch size (one group per mesh level; the largest is L0, "
              "B*N = 41864 output rows; used for the forward aggregation and the two backward gradient scatters)", "",
              "| grid threads | launches | avg us | min us |", "|---|---|---|---|"]
    for g in sorted(geo, reverse=True):
        v = geo[g]
        lines.append(f"| {g} | {len(v)} | {sum(v) / len(v) / 1e3:.1f} | {min(v) / 1e3:.1f} |")
    open(out + "_summary.md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
