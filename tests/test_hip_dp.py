"""GPU: the HIP engine under data parallelism.  Two ranks (gloo, both on cuda:0 -- RCCL needs one GPU per rank,
the 8-GPU run is the driver's) each take half of a batch; the all-reduced gradients and the exact global loss
must equal one process running the whole batch, and a Trainer step must leave identical parameters on both ranks."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from conftest import Golden
    from oracle import bsms_oracle as ro
    import bsms_gnn_amd as eng

    z, graphs = Golden("sim"), Golden("graphs")
    es, ids = graphs.levels("del300")
    cfg = ro.make_cfg(2, 32, 3, 3, 2)
    B = 2                                               # golden batch of 2 -> one sample per rank
    torch.manual_seed(50 + rank)                        # different init per rank: the broadcast must fix it
    sim = eng.BSMS_Simulator(cfg)
    if rank == 0:
        sim.load_state_dict(z.state_dict())             # golden weights + warmed normalisers on rank 0 only
    sim = sim.cuda()
    c = lambda t: t.cuda()
    sl = slice(rank, rank + 1)
    data = (c(z.t("node_in")[sl]), c(z.t("tar")[sl]), c(z.t("mask")[sl]),
            [c(e.unsqueeze(0)) for e in es], [c(i.unsqueeze(0)) for i in ids])
    engine = eng.DataParallel(sim, bucket_bytes=64 << 10)
    engine.fused.force_overlap = True                   # the per-bucket overlapped all-reduce (default on nccl only), on gloo here
    assert len(engine.grads.buckets) > 3 and any(e is not None for e in engine.fused._bucket_schedule(3))
    loss = engine.step_loss_backward(data, True)
    flat = engine.grads.flat.clone()
    # one fused optimizer step on the reduced gradients: parameters must stay identical across ranks
    opt = eng.FusedAdamW(engine.grads, lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0)
    opt.step()
    torch.cuda.synchronize()
    names = [k for k, p in sim.named_parameters() if p.requires_grad]
    order = {p: k for k, p in sim.named_parameters()}
    torch.save({"loss": loss.detach().cpu(), "flat": flat.cpu(), "params": opt.flat_p.cpu(),
                "slot_names": [order[p] for p in reversed([q for q in sim.parameters() if q.requires_grad])],
                "n_buckets": len(engine.grads.buckets), "names": names}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_single_process_large_batch(tmp_path):
    from conftest import load_golden, rel_err
    port = 29600 + os.getpid() % 2000
    out = str(tmp_path / "res")
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    z = load_golden("sim")
    assert r0["n_buckets"] > 1
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["loss"], r1["loss"])     # bit-identical across ranks
    assert torch.equal(r0["params"], r1["params"])                                         # ... also after the update
    assert abs(float(r0["loss"]) - float(z.t("loss"))) < 1e-5 * abs(float(z.t("loss")))    # exact GLOBAL masked RMSE
    # summed per-rank gradients == gradient of the single-process batch-2 step recorded in the golden file
    want = torch.cat([z.t("g/" + k).reshape(-1) for k in r0["slot_names"]])
    assert rel_err(r0["flat"], want) < 2e-5


@pytest.mark.timeout(600)
def test_bench_n2_path_with_gloo_on_one_gpu(tmp_path):
    """bench.py's own N > 1 path (one process per rank under torch.distributed.run, barrier + max-over-ranks timing, the
    two collectives of FusedStep) with gloo on one GPU -- RCCL needs a GPU per rank, the multi-GPU run is the driver's.
    Checks the JSON contract and prints the host enqueue time per step of rank 0 (8 ranks share one host)."""
    import json
    import subprocess
    env = dict(os.environ, BSMS_DIST_BACKEND="gloo", BSMS_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--workload", "cylinder"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    assert "roofline" not in line and "cpu_baseline" not in line            # N = 1 only
    print(f"\nbench --gpus 2 (gloo, one GPU): {line['value']:.1f} steps/s aggregate, host enqueue "
          f"{line['host_enqueue_ms_per_step']:.2f} ms/step on rank 0, loss {line['config']['loss']:.6f}")
    # (with gloo the figure includes two blocking host-staged collectives per step; the N = 1 bench line has the pure
    #  enqueue cost: 1.2 ms per step)


@pytest.mark.timeout(600)
def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell -- the form the driver uses for --gpus 1 -- starts its own two ranks
    (replaces nn.DataParallel, trainer/trainer.py:15-18): RCCL (backend nccl, one GPU per rank) when the box has two
    GPUs, otherwise both ranks on the one GPU over gloo, labelled as a functional run.  Exit code 0, one JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BSMS_DIST_BACKEND",
                                                              "BSMS_FORCE_DEVICE")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "cylinder"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2 and line["value"] > 0
    d = line["distributed"]
    assert d["launcher"] == "self"
    if torch.cuda.device_count() >= 2:
        assert d["backend"] == "nccl" and d["rccl_ranks"] == 2
    else:
        assert d["backend"] == "gloo" and d["rccl_ranks"] == 0 and "functional" in d["note"]
    assert line["step_ms_hipevent"]["median"] > 0
    print(f"\nbench --gpus 2 self-launched ({d['backend']}, {d['devices']} device(s)): {line['value']:.1f} steps/s aggregate")


@pytest.mark.timeout(300)
def test_bench_refuses_a_mismatched_world():
    """--gpus N under a launcher that started another number of ranks is a usage error with a message, not an assert."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=250, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def _rollout_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from conftest import Golden
    from oracle import bsms_oracle as ro
    import bsms_gnn_amd as eng
    z, graphs = Golden("sim"), Golden("graphs")
    es, ids = graphs.levels("del300")
    sim = eng.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2))
    sim.load_state_dict(z.state_dict())
    sim = sim.cuda()
    c = lambda t: t.cuda()
    ic, rmask = c(z.t("rollout_ic")), c(z.t("rollout_mask"))
    g1, i1 = [c(e.unsqueeze(0)) for e in es], [c(i.unsqueeze(0)) for i in ids]
    ics = torch.cat([ic, ic * 0.5, ic * 0.25], 0)                        # three trajectories of one mesh, two ranks: 2 + 1
    m3, gs3, is3 = rmask.repeat(3, 1, 1), [g.repeat(3, 1, 1) for g in g1], [i.repeat(3, 1) for i in i1]
    part = eng.rollout_batch(sim, ics, torch.full((4, 3, 300, 2), -7.0, device="cuda"), m3, gs3, is3, shard=True)
    full = eng.rollout_batch(sim, ics, torch.zeros(4, 3, 300, 2, device="cuda"), m3, gs3, is3, shard=True, gather=True)
    # the driver loop, trajectories dealt round-robin; targets = the (unsharded) rollouts shifted a little
    with torch.no_grad():
        loader = []
        for k in range(3):
            truth = eng.rollout_one_traj(sim, ics[k:k + 1], torch.zeros(4, 300, 2, device="cuda"), rmask, g1, i1)
            inp = ics[k:k + 1].repeat(4, 1, 1).unsqueeze(0)            # [1, T-1, N, C+p+1]: only frame 0 is read
            loader.append((inp, (truth + 0.01 * (k + 1)).unsqueeze(0), rmask.repeat(4, 1, 1).unsqueeze(0), g1, i1))
    errs = eng.rollout_dataset(sim, loader)
    torch.save({"part": part.cpu(), "full": full.cpu(), "slice": eng.rank_slice(3), "summary": errs.summary(),
                "count": float(errs.all._num_accumulations)}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rollout_trajectories_shard_across_ranks(tmp_path):
    """trajectories of a rollout are independent -- each rank advances its slice
    (`rollout_batch(shard=True)`), nothing is exchanged on the data path; `gather=True` reassembles the frames and
    `rollout_dataset` merges only the three error accumulators (src/rollout.py:86-112).  Two ranks on one GPU (gloo)."""
    from conftest import load_golden
    from oracle import bsms_oracle as ro
    import bsms_gnn_amd as eng
    port = 29650 + os.getpid() % 40
    out = str(tmp_path / "roll")
    mp.start_processes(_rollout_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["slice"] == (0, 2) and r1["slice"] == (2, 3)
    # single-process reference of the same three trajectories
    z, graphs = load_golden("sim"), load_golden("graphs")
    es, ids = graphs.levels("del300")
    sim = eng.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2))
    sim.load_state_dict(z.state_dict())
    sim = sim.cuda()
    ic, rmask = z.t("rollout_ic").cuda(), z.t("rollout_mask").cuda()
    g1, i1 = [e.unsqueeze(0).cuda() for e in es], [i.unsqueeze(0).cuda() for i in ids]
    ics = torch.cat([ic, ic * 0.5, ic * 0.25], 0)
    want = eng.rollout_batch(sim, ics, torch.zeros(4, 3, 300, 2, device="cuda"), rmask.repeat(3, 1, 1),
                             [g.repeat(3, 1, 1) for g in g1], [i.repeat(3, 1) for i in i1]).cpu()
    assert torch.equal(r0["full"], want) and torch.equal(r1["full"], want)                 # gathered: bit-identical frames on every rank
    assert torch.equal(r0["part"][:, :2], want[:, :2]) and bool((r0["part"][:, 2] == -7).all())   # un-gathered: own columns only
    assert torch.equal(r1["part"][:, 2], want[:, 2]) and bool((r1["part"][:, :2] == -7).all())
    assert r0["count"] == 3 and r1["count"] == 3
    errs = eng.RolloutErrors()
    for k in range(3):
        truth = want[:, k] + 0.01 * (k + 1)
        errs.add(want[:, k], truth, rmask.cpu().repeat(4, 1, 1))
    for name, (mean, std) in errs.summary().items():
        for r in (r0, r1):
            assert torch.allclose(r["summary"][name][0].cpu(), mean, rtol=1e-5, atol=1e-9), name
            assert torch.equal(r["summary"][name][0], r0["summary"][name][0]), name


def _probe_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from conftest import Golden
    from oracle import bsms_oracle as ro
    import bsms_gnn_amd as eng
    z, graphs = Golden("sim"), Golden("graphs")
    es, ids = graphs.levels("del300")
    sim = eng.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2))
    sim.load_state_dict(z.state_dict())
    sim = sim.cuda()
    c = lambda t: t.cuda()
    sl = slice(rank, rank + 1)
    data = (c(z.t("node_in")[sl]), c(z.t("tar")[sl]), c(z.t("mask")[sl]), [c(e.unsqueeze(0)) for e in es], [c(i.unsqueeze(0)) for i in ids])
    engine = eng.DataParallel(sim, bucket_bytes=64 << 10)
    engine.fused.probe_any_backend = True               # the nccl-only self-check of the overlapped all-reduce, exercised on gloo
    losses, flats = [], []
    for _ in range(6):
        losses.append(float(engine.step_loss_backward(data, True)))
        flats.append(engine.grads.flat.clone().cpu())
    st = engine.fused._ov_probe
    torch.save({"losses": losses, "flats": flats, "use": st["use"], "n": st["n"], "ms": st.get("measured_ms")}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_overlap_self_check_decides_once_and_identically(tmp_path):
    """step.FusedStep._overlap_now: the first four data-parallel steps measure the plain and the per-bucket overlapped
    gradient all-reduce (two each), the ranks agree on ONE decision, and every step -- whichever form it took -- produces
    the same reduced gradients (the model does not change between these steps)."""
    port = 29750 + os.getpid() % 40
    out = str(tmp_path / "probe")
    mp.start_processes(_probe_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["n"] == 4 and r0["use"] is not None and r0["use"] == r1["use"] and r0["ms"] == r1["ms"]
    for k in range(6):
        assert torch.equal(r0["flats"][k], r1["flats"][k]) and r0["losses"][k] == r1["losses"][k]      # ranks agree bit for bit
        assert torch.equal(r0["flats"][k], r0["flats"][0])                                             # plain == overlapped form
    print(f"\nself-check on gloo: plain {r0['ms']['plain']:.2f} ms, overlapped {r0['ms']['overlapped']:.2f} ms per step -> overlap {'kept' if r0['use'] else 'dropped'}")


def _rccl_world1_worker(rank, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from conftest import Golden
    from oracle import bsms_oracle as ro
    import bsms_gnn_amd as eng
    from bsms_gnn_amd import _abi

    z, graphs = Golden("sim"), Golden("graphs")
    es, ids = graphs.levels("del300")
    cfg = ro.make_cfg(2, 32, 3, 3, 2)
    c = lambda t: t.cuda()
    data = (c(z.t("node_in")), c(z.t("tar")), c(z.t("mask")), [c(e.unsqueeze(0)) for e in es], [c(i.unsqueeze(0)) for i in ids])

    def run(distributed, overlapped):
        sim = eng.BSMS_Simulator(cfg)
        sim.load_state_dict(z.state_dict())
        sim = sim.cuda()
        if not distributed:                                   # the non-distributed step: no process group in sight of FusedStep
            grads = eng.GradBuckets(list(sim.parameters()), 64 << 10)
            fused = eng.FusedStep(sim, grads, None)
            fused._world = lambda: 1
        else:
            engine = eng.DataParallel(sim, bucket_bytes=64 << 10)
            fused, grads = engine.fused, engine.grads
            fused.collectives_at_world_one = True             # a one-rank group: every collective is issued, every sum is the identity
            fused.force_overlap = overlapped
            fused.overlap_allreduce = overlapped
        losses, flats = [], []
        for _ in range(3):
            losses.append(fused(data, True).detach().cpu())
            flats.append(grads.flat.clone().cpu())
        torch.cuda.synchronize()
        return losses, flats, fused

    base_l, base_f, _ = run(False, False)
    plain_l, plain_f, _ = run(True, False)
    over_l, over_f, fused = run(True, True)
    assert fused._comm is not None and fused._overlap is not None            # the overlapped path ran: comm stream, per-block events
    assert len(fused.grads.buckets) > 3 and any(e is not None for e in fused._overlap["sched"])
    ovl = _abi.lib().bsms_streams_overlap(fused._comm.cuda_stream, torch.cuda.current_stream().cuda_stream)
    torch.save({"backend": dist.get_backend(), "base_l": base_l, "base_f": base_f, "plain_l": plain_l, "plain_f": plain_f,
                "over_l": over_l, "over_f": over_f, "streams_overlap": int(ovl)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_world_one_smoke(tmp_path):
    """The RCCL code path on the one GPU a test box has: `init_process_group("nccl", world_size=1)`,
    DataParallel's broadcast-free construction, then three FusedStep steps in the plain form (two-scalar loss all-reduce + ONE
    all-reduce of the flat gradient buffer) and three in the overlapped form (`bsms_bsgmp_bwd_ev` events -> `comm.wait_event` ->
    `all_reduce(async_op=True)` per bucket on the communication stream -> `work.wait()` -> `main.wait_stream(comm)`), plus
    `bsms_streams_overlap(comm, main)`.  A sum over one rank is the identity, so all three must equal the non-distributed step
    BIT FOR BIT.  This proves that the RCCL library loads, a communicator is created and the stream / event choreography is
    legal on gfx950 -- it says NOTHING about scaling (replaces the DataParallel wrap of trainer/trainer.py:15-18)."""
    port = 29800 + os.getpid() % 1500
    out = str(tmp_path / "rccl1.pt")
    mp.start_processes(_rccl_world1_worker, args=(port, out), nprocs=1, join=True, start_method="spawn")
    r = torch.load(out)
    assert r["backend"] == "nccl"
    for k in range(3):
        assert torch.equal(r["plain_l"][k], r["base_l"][k]) and torch.equal(r["over_l"][k], r["base_l"][k]), k
        assert torch.equal(r["plain_f"][k], r["base_f"][k]), ("plain form", k)
        assert torch.equal(r["over_f"][k], r["base_f"][k]), ("overlapped form", k)
    assert r["streams_overlap"] in (0, 1)
    print(f"\n[rccl world 1] backend {r['backend']}, comm stream overtakes the caller's stream: {bool(r['streams_overlap'])}; "
          f"loss {float(r['base_l'][0]):.7f}")
