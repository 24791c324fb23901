#!/usr/bin/env python3
"""Same-box A/B of two library builds on the whole training step: loss and every parameter gradient of one fused step
(airfoil, batch 8, fp32 unless BSMS_AB_DTYPE=bf16) saved per build, then compared bit for bit.
   python profiles/model_ab.py save out.pt [workload batch]      python profiles/model_ab.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    rel = {k: float((a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-300)) for k in a}
    print(f"{len(a)} tensors, {len(bad)} differ bitwise; worst relative L2 difference {max(rel.values()):.3e}")
    sys.exit(1 if bad else 0)
import bench
import bsms_gnn_amd as eng
kind = sys.argv[3] if len(sys.argv) > 3 else "airfoil"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
wl = bench.build_workload(kind, B, "cuda")
torch.manual_seed(0)
cfg = bench.make_cfg(wl["cfg"])
sim = eng.BSMS_Simulator(cfg).cuda()
if os.environ.get("BSMS_AB_DTYPE") == "bf16":
    sim.process.precision = "bf16"
data = bench.data_tuple(wl)
sim(data, True, True)
dp = eng.DataParallel(sim)
for _ in range(3):
    loss = dp.step_loss_backward(data, True)
torch.cuda.synchronize()
out = {"loss": torch.as_tensor(loss).detach().cpu().reshape(-1), "flat": dp.grads.flat.detach().cpu().clone()}
torch.save(out, sys.argv[2])
print("saved", sys.argv[2], float(out["loss"][0]), float(out["flat"].norm()))
