"""GMP forward / backward GPU time per mesh level (HIP events around many back-to-back calls) next to two floors:
HBM (bytes the block must move at 5.5 TB/s) and matrix cores (fp32 flops at the 419 TF/s split-bf16 peak)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
B, D = 8, 128
tot_f = tot_b = 0.0
for lvl, (n, e) in enumerate(wl["levels"]):
    K = 4 if lvl == 0 else 20   # K saved-for-backward sets are alive at once (1.2 GB each at L0)
    g = wl["m_gs"][lvl][0]
    plan = eng.plan_for(g, n)
    gmp = eng.GMP(D, 3, 2).cuda()
    x = torch.randn(B, n, D, device="cuda", requires_grad=True)
    pos = torch.rand(B, n, 2, device="cuda")
    def fwd():
        return gmp(x, g, pos, plan=plan)
    for _ in range(3):
        fwd().sum().backward()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ys = []
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(K):
        ys.append(fwd())
    ev[1].record()
    torch.cuda.synchronize()
    gy = torch.ones_like(ys[0])
    ev[2].record()
    for y in ys:
        y.backward(gy)
    ev[3].record()
    torch.cuda.synchronize()
    tf, tb = ev[0].elapsed_time(ev[1]) / K * 1e3, ev[2].elapsed_time(ev[3]) / K * 1e3
    re, rn = B * e, B * n
    bytes_f = (4 * re + re) * D * 4 + 8 * rn * D * 4          # 4 saved edge streams + aggregation read + node side
    bytes_b = (1 + 4 + 6 + 3) * re * D * 4 + 16 * rn * D * 4  # y, 4 gradients, wgrad 6, g0 three times + node side
    fl_f = 2 * (re * 3 + rn * 7) * D * D
    fl_b = 2 * fl_f
    calls = 2 if lvl < len(wl["levels"]) - 1 else 1
    tot_f += calls * tf; tot_b += calls * tb
    print(f"level {lvl}: N={n:5d} E={e:6d}  fwd {tf:7.1f} us (HBM floor {bytes_f / 5.5e6:6.1f}, matrix floor {fl_f / 419e6:6.1f})   "
          f"bwd {tb:7.1f} us (HBM floor {bytes_b / 5.5e6:6.1f}, matrix floor {fl_b / 419e6:6.1f})")
print(f"11 blocks of the U-Net: fwd {tot_f / 1e3:.2f} ms, bwd {tot_b / 1e3:.2f} ms")
