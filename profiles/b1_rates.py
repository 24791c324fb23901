#!/usr/bin/env python3
"""The batch-independent floor in two numbers: airfoil B=1 training steps/s (fused step) and B=1 rollout steps/s (eager).
   python profiles/b1_rates.py [workload] [batch] [precision: f32 | bf16 | bf16_nodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import bsms_gnn_amd as eng

kind = sys.argv[1] if len(sys.argv) > 1 else "airfoil"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
wl = bench.build_workload(kind, B, "cuda")
torch.manual_seed(0)
sim = eng.BSMS_Simulator(bench.make_cfg(wl["cfg"])).cuda()
data = bench.data_tuple(wl)
sim(data, True, True)
sim.process.precision = prec
dp = eng.DataParallel(sim)
for _ in range(20):
    dp.step_loss_backward(data, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 300
for _ in range(n):
    dp.step_loss_backward(data, True)
torch.cuda.synchronize()
train = n / (time.perf_counter() - t0)
r = bench.rollout_rate(sim, wl, steps=400)
print(f"{kind} B={B} {prec}: train {train:8.1f} steps/s ({1e3 / train:.3f} ms)   rollout eager {r['eager']:8.1f}  graph {r['hip_graph']:8.1f} steps/s")
