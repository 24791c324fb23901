"""Ablations of the L0 edge forward chain without touching the kernel: (a) training forward (saves 3 activations + sign
bits + fiber) vs no-grad forward (stores only the messages); (b) the airfoil graph vs a synthetic graph of the same size
whose sources are the 6 rows next to the target (perfectly local gathers).  Run under rocprofv3 --kernel-trace and read
the k_edge_fwd / k_chain_fwd<8, 3 durations with profiles/edge_ablate_read.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import build_workload

wl = build_workload("airfoil", 8, "cuda")
n0, e0 = wl["levels"][0]
g_air = wl["m_gs"][0][0]
q = torch.arange(e0, device="cuda")
dst = (q * n0 // e0).clamp(max=n0 - 1)
src = (dst + 1 + q % 6) % n0
g_loc = torch.stack([src, dst]).to(g_air.dtype)
torch.manual_seed(0)
gmp = eng.GMP(128, 3, 2).cuda()
pos = torch.rand(8, n0, 2, device="cuda")
for name, g in (("airfoil", g_air), ("local", g_loc)):
    plan = eng.plan_for(g, n0)
    x = torch.randn(8, n0, 128, device="cuda", requires_grad=True)
    for _ in range(3):
        gmp(x, g, pos, plan=plan)
    torch.cuda.synchronize()
    for _ in range(10):          # launches 0..9 of this graph: training forward
        gmp(x, g, pos, plan=plan)
    torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(10):      # launches 10..19: inference forward
            gmp(x, g, pos, plan=plan)
    torch.cuda.synchronize()
print("done")
