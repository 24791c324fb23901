#!/usr/bin/env python3
"""Per-parameter A/B of a training step under two settings of the experiment build (e.g. BSMS_EDGE_FUSED_F32=0 / 1):
   python profiles/efuse32_ab.py save out.pt [workload batch]      python profiles/efuse32_ab.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    rows = []
    for k in a:
        x, y = a[k].double(), b[k].double()
        rel = float((x - y).norm() / (x.norm() + 1e-300))
        rows.append((rel, k, bool(torch.equal(a[k], b[k])), float(x.norm()), bool(torch.isfinite(y).all())))
    rows.sort(reverse=True)
    nbit = sum(1 for r in rows if r[2])
    print(f"{len(rows)} tensors, {nbit} bit-identical; worst relative L2 differences:")
    for rel, k, same, n, fin in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 14]:
        print(f"  {rel:10.3e}  {k:60s} |a| {n:10.3e} {'' if fin else 'NON-FINITE'}")
    for k in sys.argv[5:]:
        x, y = a[k].double().flatten(), b[k].double().flatten()
        d = (y - x)
        print(k, "max |diff| at", int(d.abs().argmax()), "of", x.numel())
        for i in range(0, min(x.numel(), 128), 8):
            print("  ", " ".join(f"{float(v):+.2e}" for v in (d[i:i + 8] / (x.abs().max() + 1e-300))))
    sys.exit(0)
import bench
import bsms_gnn_amd as eng
kind = sys.argv[3] if len(sys.argv) > 3 else "airfoil"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
wl = bench.build_workload(kind, B, "cuda")
torch.manual_seed(0)
sim = eng.BSMS_Simulator(bench.make_cfg(wl["cfg"])).cuda()
data = bench.data_tuple(wl)
sim(data, True, True)
dp = eng.DataParallel(sim)
for _ in range(2):
    loss = dp.step_loss_backward(data, True)
torch.cuda.synchronize()
out = {"loss": torch.as_tensor(loss).detach().cpu().reshape(-1)}
for name, prm in sim.named_parameters():
    if prm not in dp.grads._slot:
        continue
    off, n = dp.grads._slot[prm]
    out[name] = dp.grads.flat[off:off + n].detach().cpu().clone()
torch.save(out, sys.argv[2])
print("saved", sys.argv[2], float(out["loss"][0]), len(out))
