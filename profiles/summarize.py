#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` run of `bench.py` (airfoil B=8 workload) into the small
text summaries kept in profiles/ (the raw traces stay in gpurun_out/, which is scratch).

  python profiles/summarize.py gpurun_out/prof2/r2 profiles/r02_step   [steps_profiled]
writes <out>_kernel_stats.csv (verbatim copy of rocprofv3's per-kernel stats) and <out>_summary.md:
  * per-kernel table in ms/step,
  * launches per step and kernel time per step,
  * the HBM-bound and matrix-bound kernel families split by launch geometry (= mesh level), with the L0 rows priced
    against their roofline (algorithmic bytes / flops of SURVEY.md section 8(d), airfoil B=8 D=128).
Kernel families are matched by NAME PREFIX (template arguments change between rounds)."""
import csv
import re
import shutil
import sys
from collections import defaultdict

# airfoil B=8, D=128 (bench.py default workload): level sizes N/E, SURVEY.md section 8
LEVELS = [(5233, 31354), (2609, 25362), (1263, 20896), (591, 16076), (236, 10078), (70, 3878)]
B, D, S = 8, 128, 4
HBM_PEAK, SPLIT_PEAK = 8.0e12, 2516.6e12 / 3   # fp32 products = three f16 MFMA products (chain.h)


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
    calls = sum(int(r["Calls"]) for r in rows)
    lines = [f"# rocprofv3 kernel summary ({prefix.split('/')[-1]}, {steps} steps incl. warm-up)", "",
             f"total kernel time {total / 1e6:.2f} ms = {total / 1e6 / steps:.2f} ms/step (kernels on side streams overlap: "
             f"the wall step is shorter), {calls / steps:.0f} launches/step", "",
             "| kernel | calls/step | avg us | ms/step | % |", "|---|---|---|---|---|"]
    for r in rows[:30]:
        t = int(r["TotalDurationNs"])
        lines.append(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                     f"{t / 1e6 / steps:.3f} | {100 * t / total:.1f} |")

    trace = sorted(csv.DictReader(open(prefix + "_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
    # The chain kernels are persistent (grid = what the chip holds), so the launch geometry does not identify the mesh
    # level; the POSITION of a launch within the step does: the U-Net visits L0..L4, L5 (bottom), L4..L0 in the forward
    # and -- up blocks first -- the same level sequence in the backward (csrc/bsgmp.hip).
    unet = [0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0]
    wg = ["decoder"] + [f"L{l} {w}" for l in unet for w in ("D x D layers", "projections")] + ["encoder"]
    families = {"k_rowsum_v4<32, false, false": ("edge aggregation (unweighted plan-order row sum)", [f"L{l}" for l in unet]),
                "k_rowsum_pair": ("both scatters of gE[0] (by source + by target) in one launch", [f"L{l}" for l in unet]),
                "k_edge_fwd<8": ("edge MLP forward chain (pipelined kernel; <8, RB, save>: RB row blocks per wave)", [f"L{l}" for l in unet]),
                "k_edge_bwd<8": ("edge MLP backward chain (pipelined kernel)", [f"L{l}" for l in unet]),
                "k_edge_fused_bwd": ("fused edge backward of the bf16 precisions (efuse.hip: recompute + dgrad + dW on chip)", [f"L{l}" for l in unet]),
                "k_edge_fwd_res": ("edge MLP forward of the bf16 precisions with LDS-resident weights (efwd.hip)", [f"L{l}" for l in unet]),
                "k_chain_fwd<8, 3, 0": ("edge MLP forward chain (IN_EDGE, OUT_LN; builds before the pipelined kernels)", [f"L{l}" for l in unet]),
                "k_chain_bwd<8, 1, 0": ("edge MLP backward chain (G_EDGE_LN; builds before the pipelined kernels)", [f"L{l}" for l in unet]),
                "k_chain_fwd<8, 1, 0": ("node MLP forward chain (IN_ROWS2, OUT_LN)", [f"L{l}" for l in unet]),
                "k_chain_bwd<8, 0, 2": ("node MLP backward chain", [f"L{l}" for l in unet]),
                "k_wgrad<": ("batched split-K weight gradients", wg)}
    fam = defaultdict(list)
    for r in trace:
        nm = short(r["Kernel_Name"])
        for key in families:
            if nm.startswith(key):
                fam[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

    def priced(key, level, avg_ns):
        if not level.startswith("L") or not level[1].isdigit():
            return ""
        n, e = LEVELS[int(level[1])]
        t = avg_ns * 1e-9
        if key.startswith("k_rowsum_v4"):
            w = B * e * D * S + B * n * D * S + 4 * (n + 1) + 4 * e
            return f"{w / 1e6:.1f} MB -> {w / t / 1e12:.2f} TB/s = {w / t / HBM_PEAK:.2f} of HBM peak"
        if key.startswith("k_rowsum_pair"):
            w = 2 * B * e * D * S + 2 * B * n * D * S
            return f"{w / 1e6:.1f} MB -> {w / t / 1e12:.2f} TB/s = {w / t / HBM_PEAK:.2f} of HBM peak"
        if key.startswith("k_edge_fwd_res") or key.startswith("k_edge_fused_bwd"):   # bf16 precisions: ONE product per fragment pair
            units = 3 if "fwd" in key else 8                      # Linears: forward 3; backward 2 recomputed + 3 dgrad + 3 dW
            fl = 2 * B * e * units * D * D
            by = B * e * (D * 2 + 20) * (1 if "fwd" in key else 2)  # y (bf16) + fiber + rstd written / read, g0 (bf16) written
            return (f"{fl / 1e9:.1f} GFLOP -> {fl / t / 1e12:.0f} TF/s = {fl / t / (3 * SPLIT_PEAK):.3f} of the dense bf16 peak; "
                    f"{by / 1e6:.0f} MB of edge rows -> {by / t / 1e12:.2f} TB/s = {by / t / HBM_PEAK:.2f} of HBM peak")
        if key.startswith("k_chain_fwd<8, 3") or key.startswith("k_chain_bwd<8, 1") or key.startswith("k_edge_"):
            fl = 2 * B * e * 3 * D * D
            by = B * e * D * S * (4 if "fwd" in key else 5)   # fwd: 3 saved activations + messages; bwd: 4 layer gradients + y
            return (f"{fl / 1e9:.1f} GFLOP -> {fl / t / 1e12:.0f} TF/s = {fl / t / SPLIT_PEAK:.2f} of split-f16 peak; "
                    f"{by / 1e6:.0f} MB -> {by / t / 1e12:.2f} TB/s = {by / t / HBM_PEAK:.2f} of HBM peak")
        if key.startswith("k_wgrad") and "D x D" in level:
            by = 2 * 3 * B * (e + n) * D * S + 2 * B * n * D * S       # G and A of every D x D Linear, read once
            fl = 2 * (3 * B * (e + n) + 2 * B * n) * D * D
            return (f"{by / 1e6:.0f} MB -> {by / t / 1e12:.2f} TB/s = {by / t / HBM_PEAK:.2f} of HBM peak; "
                    f"{fl / 1e9:.1f} GFLOP -> {fl / t / 1e12:.0f} TF/s = {fl / t / SPLIT_PEAK:.2f} of split-f16 peak")
        return ""

    for key, (title, labels) in families.items():
        v = fam[key]
        if not v:
            continue
        per = len(v) // steps
        lines += ["", f"## `{key}...>`: {title}", ""]
        if per * steps != len(v) or per != len(labels):
            lines.append(f"{len(v)} launches over {steps} steps do not match the expected {len(labels)} per step; "
                         f"avg {sum(v) / len(v) / 1e3:.1f} us")
            continue
        lines += ["| position in step | level | avg us | min us | priced against the roofline |", "|---|---|---|---|---|"]
        for pos, lab in enumerate(labels):
            d = v[pos::per]
            avg = sum(d) / len(d)
            lines.append(f"| {pos} | {lab} | {avg / 1e3:.1f} | {min(d) / 1e3:.1f} | {priced(key, lab, avg)} |")
    # the graded kernel inside the real step (cold data: the messages were just streamed out by the edge chain);
    # bench.py reports this next to its own cold / warm timings as roofline.frac_in_step
    agg = fam["k_rowsum_v4<32, false, false"]
    if agg and len(agg) == 11 * steps:
        import json
        import os
        l0 = agg[0::11] + agg[10::11]
        e0, n0 = LEVELS[0][1], LEVELS[0][0]
        w = B * e0 * D * S + B * n0 * D * S + 4 * (n0 + 1) + 4 * e0
        avg = sum(l0) / len(l0)
        json.dump({"kernel": "k_rowsum_v4<32,false,false,...> at L0 inside the training step", "avg_us": avg / 1e3,
                   "min_us": min(l0) / 1e3, "launches": len(l0), "algorithmic_bytes": w, "GBps": w / avg,
                   "frac": w / (avg * 1e-9) / HBM_PEAK, "source": os.path.basename(out) + "_summary.md"},
                  open(os.path.join(os.path.dirname(os.path.abspath(out)), "aggregation_in_step.json"), "w"), indent=1)
    open(out + "_summary.md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
