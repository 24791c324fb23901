#!/usr/bin/env python3
"""Is the host ahead of the GPU in the steady-state training loop?  Per-step host time of dp.step_loss_backward in a
loop of 60 unsynchronised steps (airfoil B=8), and where inside the step the host waits if it does.
   python profiles/host_ahead.py [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import bsms_gnn_amd as eng
from bsms_gnn_amd import _abi

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
wl = bench.build_workload("airfoil", 8, "cuda")
torch.manual_seed(0)
sim = eng.BSMS_Simulator(bench.make_cfg(wl["cfg"])).cuda()
data = bench.data_tuple(wl)
sim(data, True, True)
sim.process.precision = prec
dp = eng.DataParallel(sim)
for _ in range(20):
    dp.step_loss_backward(data, True)
torch.cuda.synchronize()
n = 60
t = np.zeros(n + 1)
t0 = time.perf_counter()
for i in range(n):
    t[i] = time.perf_counter()
    dp.step_loss_backward(data, True)
t[n] = time.perf_counter()
torch.cuda.synchronize()
t1 = time.perf_counter()
d = np.diff(t) * 1e3
print(f"{prec}: host loop {1e3 * (t[n] - t0):.1f} ms for {n} steps, GPU done after {1e3 * (t1 - t0):.1f} ms  ->  {1e3 * (t1 - t0) / n:.3f} ms/step")
print("host ms per step call:", " ".join(f"{x:.2f}" for x in d))
# which library call waits: wrap every bsms_* entry of the loaded library with a timer for 20 more steps
L = _abi.lib()
acc = {}
def wrap(name, fn):
    def f(*a):
        s = time.perf_counter(); r = fn(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - s; return r
    return f
names = ["bsms_sim_prologue", "bsms_mlp_fwd", "bsms_bsgmp_fwd_p", "bsms_sim_epilogue", "bsms_sim_loss_bwd", "bsms_mlp_bwd_ex",
         "bsms_bsgmp_bwd_ev", "bsms_mlp_bwd", "bsms_side_lanes_join"]
orig = {k: getattr(L, k) for k in names}
for k in names:
    setattr(L, k, wrap(k, orig[k]))
torch.cuda.synchronize()
s = time.perf_counter()
for _ in range(20):
    dp.step_loss_backward(data, True)
tot = time.perf_counter() - s
torch.cuda.synchronize()
for k in names:
    setattr(L, k, orig[k])
print(f"20 more steps: host {tot * 1e3 / 20:.3f} ms per step; inside the library calls (ms per step):")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"   {k:24s} {v * 1e3 / 20:.3f}")
