"""CPU: the error statistics of the rollout driver (reference src/rollout.py:64-68, 86-112, 118-143) -- oracle and product
against tests/golden/rollout_err.npz (generated with the reference's own Normalizer, tests/golden/make_golden_r4.py) -- and
the data-parallel split of a rollout: trajectories are independent, every rank takes its own and ONLY the accumulators
are merged (2 gloo ranks == one process)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden
from oracle import bsms_oracle as ro


def _trajs(z):
    n = int(z.np("shape")[0])
    return [(z.t(f"t{k}/results"), z.t(f"t{k}/target"), z.t(f"t{k}/mask")) for k in range(n)]


def test_oracle_error_statistics_match_the_reference():
    z = load_golden("rollout_err")
    stats = ro.RolloutErrorStats()
    for k, (res, tar, mask) in enumerate(_trajs(z)):
        rmse, rmse_c, rmse_t = stats.add(res, tar, mask)
        assert torch.equal(rmse, z.t(f"t{k}/rmse")) and torch.equal(rmse_c, z.t(f"t{k}/rmse_c")) and torch.equal(rmse_t, z.t(f"t{k}/rmse_t"))
        assert rmse.shape == (1, 1) and rmse_c.shape == (res.shape[0], res.shape[-1]) and rmse_t.shape == rmse_c.shape[::-1]
    for name, acc in (("all", stats.all), ("channel", stats.channel), ("time", stats.time)):
        assert torch.equal(acc._E_data.data, z.t(f"{name}/mean")), name                   # fp64, same operation order: bit-exact
        assert torch.equal(acc.std_with_epsilon(), z.t(f"{name}/std")), name
        assert torch.equal(acc._acc_weight.data, z.t(f"{name}/weight")) and torch.equal(acc._num_accumulations.data, z.t(f"{name}/count"))


def test_product_error_statistics_match_the_reference():
    import bsms_gnn_amd as eng
    z = load_golden("rollout_err")
    errs = eng.RolloutErrors()
    for k, (res, tar, mask) in enumerate(_trajs(z)):
        rmse, rmse_c, rmse_t = errs.add(res, tar, mask)
        assert torch.equal(rmse, z.t(f"t{k}/rmse")) and torch.equal(rmse_c, z.t(f"t{k}/rmse_c")) and torch.equal(rmse_t, z.t(f"t{k}/rmse_t"))
    summ = errs.summary()
    for name in ("all", "channel", "time"):
        assert torch.equal(summ[name][0], z.t(f"{name}/mean")) and torch.equal(summ[name][1], z.t(f"{name}/std")), name
    # per-channel = mean over time of the per-(time, channel) RMSE -- NOT the RMSE pooled over time (what rounds 1-3 computed)
    res, tar, mask = _trajs(z)[0]
    pooled = torch.sqrt((((res - tar) ** 2) * mask).sum(dim=(0, 1)) / mask.sum())
    assert not torch.allclose(pooled.double(), eng.rollout_errors(res, tar, mask)[1].double().mean(0))


def test_rank_slice_covers_everything_once():
    import bsms_gnn_amd as eng
    assert eng.rank_slice(7) == (0, 7)                         # no process group: one rank owns everything


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bsms_gnn_amd as eng
    from conftest import Golden
    z = Golden("rollout_err")
    n = int(z.np("shape")[0])
    errs = eng.RolloutErrors("cpu")
    mine = [k for k in range(n) if k % world == rank] if rank < 2 else []       # rank 2 (world 3) gets nothing
    for k in mine:
        errs.add(z.t(f"t{k}/results"), z.t(f"t{k}/target"), z.t(f"t{k}/mask"))
    shape = tuple(int(v) for v in z.np("shape")[[1, 3]])
    errs.synchronize(shape=shape, device="cpu")
    slices = [eng.rank_slice(m) for m in (1, 2, 5, 8)]
    torch.save({"summary": errs.summary(), "count": errs.all._num_accumulations.data.clone(), "slices": slices}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_rollout_statistics_equal_single_process(tmp_path, world):
    """Every rank accumulates ITS trajectories; one merge of the accumulators; every rank then holds the statistics of the
    single-process run (mean exact up to fp64 re-association of the weighted sum; a rank without trajectories joins with
    weight zero)."""
    port = 29950 + os.getpid() % 40 + world
    out = str(tmp_path / "res")
    mp.start_processes(_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    z = load_golden("rollout_err")
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    n = int(z.np("shape")[0])
    used = n if world == 2 else len([k for k in range(n) if k % world < 2])
    for r in res:
        assert float(r["count"]) == used
        for name in ("all", "channel", "time"):
            assert torch.equal(r["summary"][name][0], res[0]["summary"][name][0])          # identical on every rank
            if world == 2:                                                                    # all trajectories used: the golden statistics
                assert torch.allclose(r["summary"][name][0], z.t(f"{name}/mean"), rtol=1e-12, atol=0)
                assert torch.allclose(r["summary"][name][1], z.t(f"{name}/std"), rtol=1e-7, atol=1e-12)
    for m_i, m in enumerate((1, 2, 5, 8)):                     # contiguous cover of [0, m), sizes differ by at most one
        sl = [r["slices"][m_i] for r in res]
        assert sl[0][0] == 0 and sl[-1][1] == m and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        sizes = [hi - lo for lo, hi in sl]
        assert max(sizes) - min(sizes) <= 1


def _gather_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import bsms_gnn_amd as eng
    from conftest import Golden
    from oracle import bsms_oracle as ro
    graphs = Golden("graphs")
    es, ids = graphs.levels("del64")
    n, C, T = 64, 2, 3
    torch.manual_seed(0)
    model = ro.BSMS_Simulator(ro.make_cfg(C, 16, 2, 2, 2))
    pos = torch.tensor(graphs.np("del64/pos"), dtype=torch.float32)
    res = {}
    for B in (1, 2, 4):                                   # B < world, B == world - 1, B > world
        gen = torch.Generator().manual_seed(B)
        ic = torch.cat([torch.randn(B, n, C, generator=gen), pos.expand(B, n, 2), torch.zeros(B, n, 1)], -1)
        mask = torch.ones(B, n, 1)
        gs = [e.unsqueeze(0).repeat(B, 1, 1) for e in es[:3]]
        iis = [i.unsqueeze(0).repeat(B, 1) for i in ids[:2]]
        model((ic, ic[..., :C], mask, gs, iis), True, True)   # normaliser statistics (same on every rank)
        full = eng.rollout_batch(model, ic, torch.zeros(T, B, n, C), mask, gs, iis)                 # every trajectory on this rank
        got = eng.rollout_batch(model, ic, torch.zeros(T, B, n, C), mask, gs, iis, shard=True, gather=True)
        res[B] = (full, got)
    torch.save(res, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_rollout_gather_with_fewer_trajectories_than_ranks(tmp_path):
    """`rollout_batch(shard=True, gather=True)` with B < world -- rank 0's slice is the whole batch, the
    others' slices are empty; every rank must still enter the all_gather (the old per-rank guard deadlocked) and every
    rank must return all frames."""
    world = 3
    port = 29990 + os.getpid() % 40
    out = str(tmp_path / "g")
    mp.start_processes(_gather_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        res = torch.load(f"{out}.{r}")
        for B, (full, got) in res.items():
            assert torch.equal(full, got), (r, B)


def test_rollout_errors_rejects_an_unrepeated_mask():
    import bsms_gnn_amd as eng
    res, tar = torch.randn(4, 10, 2), torch.randn(4, 10, 2)
    with pytest.raises(ValueError):
        eng.rollout_errors(res, tar, torch.ones(10, 1))
    with pytest.warns(DeprecationWarning):
        eng.rollout_rmse(res, tar, torch.ones(4, 10, 1))
