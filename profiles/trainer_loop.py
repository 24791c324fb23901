"""End-to-end Trainer.iter on HOST batches, as a DataLoader delivers them (fresh CPU tensors every step; consistent mesh:
edge lists and kept ids repeated along the batch axis), against the device-resident step bench.py times.
  python profiles/trainer_loop.py [workload] [batch]"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import build_workload, make_cfg, data_tuple, usable_cpus

torch.set_num_threads(max(1, min(usable_cpus() // 2, 8)))
kind = sys.argv[1] if len(sys.argv) > 1 else "airfoil"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = build_workload(kind, B, "cuda")
cfg = make_cfg(wl["cfg"])
torch.manual_seed(0)
sim = eng.BSMS_Simulator(cfg).cuda()
opt = SimpleNamespace(peak_lr=1e-4, weight_decay=1e-4, warmup_steps=10, decay_steps=1000, gnorm_clip=1.0)
tr = eng.Trainer(sim, cfg, opt)
dev = data_tuple(wl)
host = tuple([x.cpu() for x in t] if isinstance(t, (list, tuple)) else t.cpu() for t in dev)
host = (host[0], host[1], host[2], [g.contiguous() for g in host[3]], [i.contiguous() for i in host[4]])   # [B,2,E] / [B,Nk] materialised
fresh = lambda: tuple([x.clone() for x in t] if isinstance(t, list) else t.clone() for t in host)
sim(dev, True, True)
for _ in range(5):
    tr.iter(fresh())
torch.cuda.synchronize()
N = 50
batches = [fresh() for _ in range(N)]
t0 = time.perf_counter()
for b in batches:
    tr.iter(b)
torch.cuda.synchronize()
t_host = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    tr.dp.step_loss_backward(dev, True); tr.optimizer.step(1e-4)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / N
print(f"{kind} B={B}: Trainer.iter on host batches {t_host * 1e3:.2f} ms/step ({1 / t_host:.1f} steps/s); "
      f"device-resident step + optimizer {t_dev * 1e3:.2f} ms/step ({1 / t_dev:.1f} steps/s)")
if os.environ.get("LOOP_PROFILE"):
    import cProfile, pstats
    batches = [fresh() for _ in range(20)]
    pr = cProfile.Profile(); pr.enable()
    for b in batches:
        tr.iter(b)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
