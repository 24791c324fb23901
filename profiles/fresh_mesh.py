"""Variable-mesh training (the reference's cylinder_flow path, consistent_mesh: false) when EVERY batch is a new
combination of meshes -- what a shuffled DataLoader delivers: the block-diagonal plans of the batch are not in the
cache.  Times (a) a step on a cached batch, (b) a step on a fresh batch (new collate, plans, edge weights).
  python profiles/fresh_mesh.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsms_gnn_amd as eng
from bench import WORKLOADS, build_mesh, make_cfg

from bench import usable_cpus
torch.set_num_threads(max(1, min(usable_cpus() // 2, 8)))   # a 256-thread default on a 16-CPU cgroup quota gets the process throttled (80 ms stalls)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kind = "cylinder"
w = WORKLOADS[kind]
n, c = w["nodes"], w["out_dim"]
gen = torch.Generator().manual_seed(0)
pool = []
for seed in range(2 * B):                      # a "dataset" of 2B meshes
    pts, m_es, m_ids = build_mesh(kind, seed=seed)
    state, target = torch.randn(n, c, generator=gen), torch.randn(n, c, generator=gen)
    x = torch.cat([state, torch.tensor(pts, dtype=torch.float32), torch.zeros(n, 1)], -1)
    sizes = [n] + [len(i) for i in m_ids]
    pool.append([eng.LevelData(torch.tensor(m_es[l]), sizes[l], face=torch.tensor(m_ids[l]) if l < w["levels"] else None,
                               x=x if l == 0 else None, y=target if l == 0 else None,
                               mask=torch.ones(n, 1) if l == 0 else None) for l in range(w["levels"] + 1)])
torch.manual_seed(0)
sim = eng.BSMS_Simulator(make_cfg(w)).cuda()
dp = eng.DataParallel(sim)
perm = torch.Generator().manual_seed(1)

def batch():
    idx = torch.randperm(len(pool), generator=perm)[:B].tolist()
    return eng.collate_variable_meshes([pool[i] for i in idx])

def to_dev(b):
    mover = getattr(eng, "move_to_device", None)
    return [d.to("cuda", intern=True) for d in b]

data = to_dev(batch())
sim(data, False, True)
for _ in range(5):
    dp.step_loss_backward(data, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    dp.step_loss_backward(data, False)
torch.cuda.synchronize()
print(f"cached batch : {(time.perf_counter() - t0) / 30 * 1e3:7.2f} ms/step")
ts = {"collate": 0.0, "to_dev": 0.0, "step": 0.0}
N, WARM = 60, 40
for it in range(WARM + N):
    t0 = time.perf_counter(); b = batch(); t1 = time.perf_counter(); d = to_dev(b); torch.cuda.synchronize(); t2 = time.perf_counter()
    dp.step_loss_backward(d, False); torch.cuda.synchronize(); t3 = time.perf_counter()
    if it >= WARM:
        ts["collate"] += t1 - t0; ts["to_dev"] += t2 - t1; ts["step"] += t3 - t2
print("fresh batch  : " + "  ".join(f"{k} {v / N * 1e3:7.2f} ms" for k, v in ts.items()), f"  plans built {eng.graph.LevelPlan.constructed}")
batches = [batch() for _ in range(N)]            # pipelined: no synchronisation between steps (a loader thread would collate)
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in batches:
    dp.step_loss_backward(to_dev(b), False)
torch.cuda.synchronize()
print(f"fresh, pipelined (upload + plans + step, no sync): {(time.perf_counter() - t0) / N * 1e3:7.2f} ms/step")
# ---- round 6: the same stream of fresh batches assembled ON THE DEVICE from per-mesh plans that stay in HBM (graph.MeshBank:
# bsms_plan_concat per level, cached edge weights concatenated; no index collate, no index upload, no CSR build)
bank = eng.MeshBank(sim.process, "cuda")
def samples():
    return [pool[i] for i in torch.randperm(len(pool), generator=perm)[:B].tolist()]
for _ in range(WARM):
    dp.step_loss_backward(bank.collate(samples()), False)
torch.cuda.synchronize()
tc = tstep = 0.0
for _ in range(N):
    t0 = time.perf_counter(); d = bank.collate(samples()); torch.cuda.synchronize(); t1 = time.perf_counter()
    dp.step_loss_backward(d, False); torch.cuda.synchronize(); t2 = time.perf_counter()
    tc += t1 - t0; tstep += t2 - t1
print(f"fresh batch, MeshBank (synchronised): collate {tc / N * 1e3:7.2f} ms  step {tstep / N * 1e3:7.2f} ms   plans built {eng.graph.LevelPlan.constructed}")
todo = [samples() for _ in range(N)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for sm in todo:
    dp.step_loss_backward(bank.collate(sm), False)
torch.cuda.synchronize()
print(f"fresh, pipelined, MeshBank (device collate + step, no sync): {(time.perf_counter() - t0) / N * 1e3:7.2f} ms/step")
if os.environ.get("FRESH_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    b = batch()
    pr.enable()
    d = to_dev(b)
    dp.step_loss_backward(d, False); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)

# ---- the same stream of fresh batches through Trainer.iter, alone and behind the prefetch thread
from types import SimpleNamespace
cfg = make_cfg(w); cfg.consistent_mesh = False
torch.manual_seed(0)
tr = eng.Trainer(eng.BSMS_Simulator(cfg).cuda(), cfg, SimpleNamespace(peak_lr=1e-4, weight_decay=1e-4, warmup_steps=10, decay_steps=10000, gnorm_clip=1.0))
tr.model(to_dev(batch()), False, True)
for name, wrap in (("Trainer.iter", lambda it: it), ("Trainer.iter behind DevicePrefetcher", lambda it: eng.DevicePrefetcher(it, tr))):
    for b in wrap([batch() for _ in range(20)]):
        tr.iter(b)
    batches = [batch() for _ in range(N)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in wrap(batches):
        tr.iter(b)
    torch.cuda.synchronize()
    print(f"fresh batches, {name}: {(time.perf_counter() - t0) / N * 1e3:7.2f} ms/step")
for _ in range(20):
    tr.iter(tr.collate(samples()))
todo = [samples() for _ in range(N)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for sm in todo:
    tr.iter(tr.collate(sm))
torch.cuda.synchronize()
print(f"fresh batches, Trainer.iter(Trainer.collate(samples)) [MeshBank]: {(time.perf_counter() - t0) / N * 1e3:7.2f} ms/step")
