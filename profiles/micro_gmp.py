"""Micro-benchmark used for PMC runs: GMP forward+backward at airfoil L0 (B=8, D=128), a few iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
lvl = int(os.environ.get("LEVEL", "0"))
n0, e0 = wl["levels"][lvl]
g0 = wl["m_gs"][lvl][0]
plan = eng.plan_for(g0, n0)
gmp = eng.GMP(128, 3, 2).cuda()
x = torch.randn(8, n0, 128, device="cuda", requires_grad=True)
pos = torch.rand(8, n0, 2, device="cuda")
for _ in range(int(os.environ.get("ITERS", "3"))):
    y = gmp(x, g0, pos, plan=plan)
    y.square().sum().backward()
torch.cuda.synchronize()
print("done", n0, e0)
