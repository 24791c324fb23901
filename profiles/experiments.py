"""A/B micro-experiments on the L0 GMP forward (airfoil, B=8, D=128): per-kernel time from HIP events around
the whole GMP forward under different debug flags.  Usage: python profiles/experiments.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import build_workload, time_kernel
wl = build_workload("airfoil", 8, "cuda")
L = eng._abi.lib()
raw = ctypes.CDLL(eng._abi.LIB_PATH)
for lvl in (0, 3):
    n0, e0 = wl["levels"][lvl]
    g0 = wl["m_gs"][lvl][0]
    plan = eng.plan_for(g0, n0)
    gmp = eng.GMP(128, 3, 2).cuda()
    x = torch.randn(8, n0, 128, device="cuda")
    pos = torch.rand(8, n0, 2, device="cuda")
    for flags in (0, 1, 2, 4):
        raw.bsms_debug_set_flags(flags)
        with torch.no_grad():
            ms = time_kernel(lambda: gmp(x, g0, pos, plan=plan), iters=20, warm=3)
        print(f"level {lvl} flags {flags}: GMP forward {ms * 1e3:.1f} us")
raw.bsms_debug_set_flags(0)
