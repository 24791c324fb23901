"""How long does the host need to ENQUEUE one training step (no sync) vs. the GPU time of the step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import build_workload, make_cfg, data_tuple
wl = build_workload(os.environ.get("WORKLOAD", "airfoil"), 8, "cuda")
torch.manual_seed(0)
sim = eng.BSMS_Simulator(make_cfg(wl["cfg"])).cuda()
data = data_tuple(wl)
sim(data, True, True)
dp = eng.DataParallel(sim)
for _ in range(5):
    dp.step_loss_backward(data, True)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    dp.step_loss_backward(data, True)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {t_enq / K * 1e3:.2f} ms/step, total {t_all / K * 1e3:.2f} ms/step, cpus {os.cpu_count()}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    dp.step_loss_backward(data, True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
