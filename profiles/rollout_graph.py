"""Rollout step rate: eager vs HIP-graph replay, capture excluded.  python profiles/rollout_graph.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bsms_gnn_amd.rollout import _Stepper
from bench import build_workload, make_cfg, data_tuple
wl = build_workload("airfoil", 8, "cuda")
torch.manual_seed(0)
sim = eng.BSMS_Simulator(make_cfg(wl["cfg"])).cuda()
sim(data_tuple(wl), True, True)
for B in (1, 4):
    ic, mask = wl["node_in"][:B].contiguous(), wl["mask"][:B].contiguous()
    g1, i1 = [g[:B] for g in wl["m_gs"]], [i[:B] for i in wl["m_ids"]]
    with torch.no_grad():
        for use_graph in (False, True):
            t0 = time.perf_counter()
            st = _Stepper(sim, ic, mask, g1, i1, 3, use_graph)
            for _ in range(5): st.step()
            torch.cuda.synchronize(); setup = time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(200): st.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 200
            print(f"B={B} {'graph' if use_graph else 'eager'}: {1 / dt:8.1f} steps/s ({dt * 1e3:.3f} ms), setup {setup * 1e3:.1f} ms")
