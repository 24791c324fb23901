import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BSMS_PLAN_CACHE"] = "24"
import torch
torch.set_num_threads(4)
import bsms_gnn_amd as eng
from types import SimpleNamespace
from bench import WORKLOADS, build_mesh, make_cfg
w = WORKLOADS["cylinder"]; c = w["out_dim"]
gen = torch.Generator().manual_seed(0); pool = []
for seed in range(12):
    # meshes of DIFFERENT sizes (the real dataset): vary the node count
    import bench
    n = w["nodes"] - 37 * seed
    old = bench.WORKLOADS["cylinder"]["nodes"]; bench.WORKLOADS["cylinder"]["nodes"] = n
    pts, m_es, m_ids = build_mesh("cylinder", seed=seed)
    bench.WORKLOADS["cylinder"]["nodes"] = old
    n = pts.shape[0]
    state, target = torch.randn(n, c, generator=gen), torch.randn(n, c, generator=gen)
    x = torch.cat([state, torch.tensor(pts, dtype=torch.float32), torch.zeros(n, 1)], -1)
    sizes = [n] + [len(i) for i in m_ids]
    pool.append([eng.LevelData(torch.tensor(m_es[l]), sizes[l], face=torch.tensor(m_ids[l]) if l < w["levels"] else None,
                               x=x if l == 0 else None, y=target if l == 0 else None, mask=torch.ones(n, 1) if l == 0 else None)
                 for l in range(w["levels"] + 1)])
cfg = make_cfg(w); cfg.consistent_mesh = False
torch.manual_seed(0)
tr = eng.Trainer(eng.BSMS_Simulator(cfg).cuda(), cfg, SimpleNamespace(peak_lr=1e-4, weight_decay=1e-4, warmup_steps=10, decay_steps=10000, gnorm_clip=1.0))
perm = torch.Generator().manual_seed(1)
def batches(k):
    for _ in range(k):
        idx = torch.randperm(len(pool), generator=perm)[:8].tolist()
        yield eng.collate_variable_meshes([pool[i] for i in idx])
tr.model([d.to("cuda", intern=True) for d in next(batches(1))], False, True)
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter(); losses = []
for i, b in enumerate(eng.DevicePrefetcher(batches(400), tr)):
    l = tr.iter(b)
    if i % 100 == 99:
        torch.cuda.synchronize()
        print(i + 1, "steps", f"{(time.perf_counter() - t0) / (i + 1) * 1e3:.2f} ms/step  loss {float(l):.4f}  allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB (start {m0 / 2**20:.0f})  reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB  plans {eng.graph.LevelPlan.constructed}  grave {len(eng.graph._GRAVE)}")

# ---- round 6: the same stream of fresh batches through Trainer.collate (graph.MeshBank: meshes of DIFFERENT sizes resident in HBM,
# every batch assembled by bsms_plan_concat), a small plan cache so that the unions are retired and their blocks recycled all the time;
# checks that the loss of a batch equals the host-collated route's bit for bit every 100 steps and that memory stays flat
def sample_sets(k):
    for _ in range(k):
        yield [pool[i] for i in torch.randperm(len(pool), generator=perm)[:8].tolist()]
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); t0 = time.perf_counter()
for i, sm in enumerate(sample_sets(1200)):
    l = tr.iter(tr.collate(sm))
    if i % 100 == 99:
        a = float(tr.get_loss(tr.collate(sm))); b = float(tr.get_loss(eng.collate_variable_meshes(sm)))
        torch.cuda.synchronize()
        print("MeshBank", i + 1, "steps", f"{(time.perf_counter() - t0) / (i + 1) * 1e3:.2f} ms/step  loss {float(l):.4f}  device-collated == host-collated loss: {a == b} ({a:.6f})  "
              f"allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB (start {m0 / 2**20:.0f})  reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB  plans {eng.graph.LevelPlan.constructed}  grave {len(eng.graph._GRAVE)}")
