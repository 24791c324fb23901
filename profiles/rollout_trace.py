"""Kernel timeline of one eager rollout step (B = 1, airfoil):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ro -o x -- python profiles/rollout_trace.py run
  python profiles/rollout_trace.py show gpurun_out/ro/.../x_kernel_trace.csv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch, time
    import bsms_gnn_amd as eng
    from bench import build_workload, make_cfg, data_tuple
    wl = build_workload("airfoil", 8, "cuda")
    torch.manual_seed(0)
    sim = eng.BSMS_Simulator(make_cfg(wl["cfg"])).cuda()
    sim(data_tuple(wl), True, True)
    ic, mask = wl["node_in"][:1].contiguous(), wl["mask"][:1].contiguous()
    g1, i1 = [g[:1] for g in wl["m_gs"]], [i[:1] for i in wl["m_ids"]]
    res = torch.zeros(40, ic.shape[1], 3, device="cuda")
    eng.rollout_one_traj(sim, ic, res[:3], mask, g1, i1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.rollout_one_traj(sim, ic, res, mask, g1, i1)
    torch.cuda.synchronize(); print("rollout steps/s", 39 / (time.perf_counter() - t0))
else:
    import csv, re
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_sim_prologue" in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    seg = rows[a:b]
    short = lambda n: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n)).split("(")[0][:44]
    t0 = int(seg[0]["Start_Timestamp"]); prev = t0; busy = 0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:8.1f}  gap {(s - prev) / 1e3:5.1f}  dur {(e - s) / 1e3:6.1f}  grid {r['Grid_Size_X']:>8}  {short(r['Kernel_Name'])}")
        prev = e; busy += e - s
    print("step", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us; busy", busy / 1e3, "us;", len(seg), "launches")
