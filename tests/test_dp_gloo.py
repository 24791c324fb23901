"""CPU, world_size 2 over gloo: the data-parallel wrapper (bsms_gnn_amd.dp) -- bucketed gradient all-reduce,
exact global masked RMSE, normaliser synchronisation -- reproduces the single-process large-batch result.
The replica here is the CPU oracle model (the HIP engine needs a GPU); dp.py is model-agnostic."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from conftest import Golden
    from oracle import bsms_oracle as ro
    import bsms_gnn_amd.dp as dp                       # imports without touching the GPU
    from bsms_gnn_amd.model import Normalizer

    graphs = Golden("graphs")
    es, ids = graphs.levels("del64")
    n, B = 64, 4                                        # global batch 4 -> 2 per rank
    cfg = ro.make_cfg(2, 32, 2, 2, 2)
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    model = ro.BSMS_Simulator(cfg)
    gen = torch.Generator().manual_seed(7)
    pos = torch.tensor(graphs.np("del64/pos"), dtype=torch.float32)
    state = torch.randn(B, n, 2, generator=gen)
    ntype = (torch.rand(B, n, 1, generator=gen) < 0.2).float()
    node_in = torch.cat([state, pos.expand(B, n, 2), ntype], -1)
    tar = state + 0.1 * torch.randn(B, n, 2, generator=gen)
    mask = (ntype == 0).float()
    gs = lambda b: [e.unsqueeze(0).repeat(b, 1, 1) for e in es[:3]]
    iis = lambda b: [i.unsqueeze(0).repeat(b, 1) for i in ids[:2]]
    sl = slice(rank * B // world, (rank + 1) * B // world)
    local = (node_in[sl], tar[sl], mask[sl], gs(B // world), iis(B // world))

    engine = dp.DataParallel(model, bucket_bytes=16 << 10)      # several buckets
    assert len(engine.grads.buckets) > 2
    model(local, True, True)                                    # per-rank normaliser accumulation ...
    # ... merged with the product's Normalizer.synchronize arithmetic (run on the oracle's identical fields)
    for nm in (model._inputNormalizer, model._targetNormalizer):
        Normalizer.synchronize(nm, None)
    loss = engine.step_loss_backward(local, True)
    total = engine.grads.clip_(1e9)
    res = {"loss": loss.detach(), "flat": engine.grads.flat.clone(), "norm": total,
           "stats": torch.cat([model._inputNormalizer._E_data, model._targetNormalizer._E_data_squared]),
           "p0": next(model.parameters()).detach().clone()}
    # every parameter's .grad is a view of the flat buffer
    for p in engine.grads.params:
        off, cnt = engine.grads._slot[p]
        assert p.grad.data_ptr() == engine.grads.flat.data_ptr() + 4 * off
    # second step after zero(): same result (hooks re-arm)
    loss2 = engine.step_loss_backward(local, True)
    assert torch.allclose(loss2, loss) and torch.allclose(engine.grads.flat, res["flat"], rtol=1e-6, atol=1e-8)
    if rank == 0:
        # reference: one process, whole batch, same (rank-0) initial weights
        torch.manual_seed(100)
        ref = ro.BSMS_Simulator(cfg)
        whole = (node_in, tar, mask, gs(B), iis(B))
        # the merged statistics equal sequential accumulation of the two half batches
        ref((node_in[:2], tar[:2], mask[:2], gs(2), iis(2)), True, True)
        ref((node_in[2:], tar[2:], mask[2:], gs(2), iis(2)), True, True)
        l = ro.masked_rmse(ref(whole, True, False), tar, mask)
        l.backward()
        res["ref_loss"] = l.detach()
        res["ref_flat"] = torch.cat([p.grad.reshape(-1) for p in reversed([q for q in ref.parameters() if q.requires_grad])])
        res["ref_stats"] = torch.cat([ref._inputNormalizer._E_data, ref._targetNormalizer._E_data_squared])
        res["ref_p0"] = next(ref.parameters()).detach().clone()
    # Restore-then-warm-up (Trainer.restore with a checkpoint taken during warm-up): statistics every rank already
    # shares (`base`) plus per-rank additions -> the merge counts the shared part ONCE.
    nm = Normalizer(3)
    gen2 = torch.Generator().manual_seed(11)
    chunks = [torch.randn(50, 3, generator=gen2, dtype=torch.float64) * (k + 1) + k for k in range(3)]
    nm._accumulate(chunks[0])                                   # "restored": identical on both ranks
    base = nm.snapshot()
    nm._accumulate(chunks[1 + rank])                            # warm-up continues on this rank's shard
    nm.synchronize(None, base=base)
    seq = Normalizer(3)
    for c in chunks:
        seq._accumulate(c)
    res["base_merge"] = torch.cat([nm._acc_weight.data, nm._num_accumulations.data, nm._E_data.data, nm._E_data_squared.data])
    res["base_seq"] = torch.cat([seq._acc_weight.data, seq._num_accumulations.data, seq._E_data.data, seq._E_data_squared.data])
    torch.save(res, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradients_match_single_process(tmp_path):
    port = 29500 + os.getpid() % 2000
    out = str(tmp_path / "res")
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["p0"], r1["p0"]) and torch.equal(r0["p0"], r0["ref_p0"])       # broadcast from rank 0
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["loss"], r1["loss"])   # ranks agree bit-for-bit
    torch.testing.assert_close(r0["stats"], r0["ref_stats"], rtol=1e-12, atol=1e-15)     # fp64 normaliser merge
    torch.testing.assert_close(r0["stats"], r1["stats"], rtol=0, atol=0)
    torch.testing.assert_close(r0["loss"], r0["ref_loss"], rtol=1e-6, atol=0)            # exact global RMSE
    err = (r0["flat"] - r0["ref_flat"]).abs().max() / r0["ref_flat"].abs().max()
    assert err < 1e-5, err                                                                 # summed grads == big batch
    assert abs(float(r0["norm"]) - float(r0["ref_flat"].norm())) < 1e-4 * float(r0["ref_flat"].norm())
    for r in (r0, r1):                                                                     # shared base counted once
        torch.testing.assert_close(r["base_merge"], r["base_seq"], rtol=1e-12, atol=1e-15)


def test_bucket_schedule_follows_the_backward_execution_order():
    """step.FusedStep._bucket_schedule: every gradient bucket is released by the bsms_bsgmp_bwd_ev event of the LAST U-Net
    block (in execution order: up_gmps[L-1] .. up_gmps[0], bottom_gmp, down_gmps[L-1] .. down_gmps[0]) any of its
    parameters belongs to; buckets holding encoder parameters wait for the final join.  Host logic only (no launch)."""
    from types import SimpleNamespace
    import bsms_gnn_amd as eng
    L = 3
    cfg = SimpleNamespace(out_dim=2, latent_dim=32, hidden_layer=2, unet_depth=L, pos_dim=2)
    sim = eng.BSMS_Simulator(cfg)
    grads = eng.GradBuckets(list(sim.parameters()), bucket_bytes=16 << 10)
    step = eng.FusedStep(sim, grads)
    sched = step._bucket_schedule(L)
    assert len(sched) == len(grads.buckets) > 4
    name = {p: k for k, p in sim.named_parameters()}
    exec_order = [f"process.up_gmps.{i}" for i in range(L - 1, -1, -1)] + ["process.bottom_gmp"] + \
                 [f"process.down_gmps.{i}" for i in range(L - 1, -1, -1)]
    for bk, e in zip(grads.buckets, sched):
        names = [name[p] for p in bk["params"]]
        if any(n.startswith("encode.") for n in names):
            assert e is None
            continue
        want = max(0 if n.startswith("decode.") else next(i for i, pre in enumerate(exec_order) if n.startswith(pre + ".")) for n in names)
        assert e == want, (names[0], e, want)
    assert sched[0] is not None and sched[-1] is None          # the decoder's bucket goes first, the encoder's last


class _Recorder:
    """Stand-in for torch.distributed inside step.FusedStep._backward_overlapped: records the collectives in issue order."""

    class ReduceOp:
        SUM = "sum"

    class _Work:
        def wait(self):
            pass

    def __init__(self):
        self.calls = []

    def all_reduce(self, t, op=None, group=None, async_op=False):
        self.calls.append((t.data_ptr(), t.numel()))
        return self._Work()


@pytest.mark.parametrize("depth", [4, 5, 6])
def test_overlapped_allreduce_issue_order_is_rank_independent(depth, monkeypatch):
    """the scheduling of the per-bucket gradient all-reduce (step.FusedStep) with a fake
    process group that records the call order -- two 'ranks' (same architecture, different weights) issue the same
    collectives in the same order, every bucket exactly once, event-released buckets in non-decreasing event order
    (= the order the backward completes them) and the join-released ones last.  No GPU: the backward itself is stubbed."""
    import bsms_gnn_amd as eng
    import bsms_gnn_amd.dp as dp
    import bsms_gnn_amd.step as step
    from oracle import bsms_oracle as ro

    class _Ev:
        cuda_event = 0
        def record(self):
            pass

    class _Stream:
        def __init__(self, device=None):
            pass
        def wait_event(self, e):
            pass
        def wait_stream(self, s):
            pass

    class _Ctx:
        def __init__(self, s):
            pass
        def __enter__(self):
            return self
        def __exit__(self, *a):
            return False

    orders = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        sim = eng.BSMS_Simulator(ro.make_cfg(3, 128, 3, depth, 2))
        grads = dp.GradBuckets(list(sim.parameters()), 2 << 20, None)
        fs = step.FusedStep(sim, grads)
        rec = _Recorder()
        monkeypatch.setattr(step, "dist", rec)
        monkeypatch.setattr(step.torch.cuda, "Event", _Ev)
        monkeypatch.setattr(step.torch.cuda, "Stream", _Stream)
        monkeypatch.setattr(step.torch.cuda, "stream", _Ctx)
        monkeypatch.setattr(step.torch.cuda, "current_stream", lambda: _Stream())
        monkeypatch.setattr(step._abi, "ptr_array", lambda ps: (ps, None))
        monkeypatch.setattr(fs, "_backward", lambda *a, **k: None)
        fs._backward_overlapped(dict(depth=depth, h0=torch.zeros(1)), None, None, None, 8, 100)
        sched = fs._overlap["sched"]
        assert len(sched) == len(grads.buckets) >= 3
        ev = [e for e in sched if e is not None]
        assert ev == sorted(ev) and all(0 <= e <= 2 * depth for e in ev)           # buckets complete in backward order
        assert sched[-1] is None or sched[-1] == 2 * depth                           # the encoder's bucket: only the final join
        base = grads.flat.data_ptr()
        order = [((ptr - base) // 4, n) for ptr, n in rec.calls]
        assert sorted(order) == sorted(((bk["view"].data_ptr() - base) // 4, bk["view"].numel()) for bk in grads.buckets)   # each bucket once
        assert [k for k, e in fs._issue_order(sched)] == [k for k, e in enumerate(sched) if e is not None] + [k for k, e in enumerate(sched) if e is None]
        orders.append(order)
    assert orders[0] == orders[1]
