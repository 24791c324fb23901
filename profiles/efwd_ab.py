"""Digest of a bf16 airfoil step (prediction, loss, every gradient): run with BSMS_EDGE_FWD_RES=0 / 1 against the experiment build;
equal digests = the resident-weights edge forward (efwd.hip) is bit-identical to the ring kernel.   python profiles/efwd_ab.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import build_workload
wl = build_workload("airfoil", 8, "cuda")
h = hashlib.sha256()
for lvl, p in ((0, 2), (3, 2), (5, 2)):
    n0, e0 = wl["levels"][lvl]
    g0 = wl["m_gs"][lvl][0]
    torch.manual_seed(lvl)
    net = eng.BSGMP(0, 128, 3, p).cuda()
    net.precision = "bf16"
    pos = torch.rand(8, n0, p, device="cuda")
    x = torch.randn(8, n0, 128, device="cuda", requires_grad=True)
    with torch.no_grad():
        yi = net(x, [], [g0], pos)
    y = net(x, [], [g0], pos)
    y.square().mean().backward()
    torch.cuda.synchronize()
    for t in [yi, y, x.grad] + [q.grad for q in net.parameters()]:
        h.update(t.detach().cpu().numpy().tobytes())
    print(f"level {lvl}: |y| {float(y.abs().mean()):.6f} inference==training {bool((yi == y).all())}")
print("digest", h.hexdigest())
