#!/usr/bin/env python3
"""k_wgrad (BSMS_WGRAD_PIPE=0) against the pipelined k_wgrad_h2p (=1, default): the weight gradients of one 3-Linear MLP
backward over airfoil-L0-sized rows must agree bit for bit; prints the time of the backward.
   python profiles/wgrad_ab.py save out.pt     (run once per setting)      python profiles/wgrad_ab.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    worst = max(float((a[k] - b[k]).abs().max()) for k in a)
    print(f"{len(a)} tensors, {len(bad)} differ bitwise, worst abs difference {worst:.3e}")
    sys.exit(1 if bad else 0)
import bsms_gnn_amd as eng
out = {}
for R in (250880, 250880 - 37, 5000, 131):
    torch.manual_seed(R)
    mlp = eng.MLP(128, 128, 128, 3, True).cuda()
    x = (torch.randn(R, 128, device="cuda") * torch.logspace(-3, 2, 128, device="cuda")).requires_grad_(True)
    for it in range(3):
        mlp.zero_grad(set_to_none=True); x.grad = None
        y = mlp(x)
        g = torch.randn_like(y) * 1e-3
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); y.backward(g); t1.record(); torch.cuda.synchronize()
    print(f"R={R}: backward {t0.elapsed_time(t1) * 1e3:.0f} us")
    for n, p in mlp.named_parameters():
        out[f"{R}.{n}"] = p.grad.detach().cpu()
    out[f"{R}.x"] = x.grad.detach().cpu()
torch.save(out, sys.argv[2])
