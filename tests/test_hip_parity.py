"""GPU parity: the HIP path (through the C ABI of libbsms_hip.so) against the CPU oracle and against
the golden vectors generated from the reference.

Tolerances: integer / index results bit-exact; segment sums, edge weights and transitions are summed in
the reference's edge order and must be BIT-IDENTICAL to the CPU result; dense fp32 results within
1e-5 of the tensor scale (max|a-b| / max|b|, BASELINE.json north_star), gradients likewise 1e-5 (data kept away from
ReLU kinks, conftest.KinkMargin; at full size the three-way fp64 criterion of tests/test_hip_fullsize.py applies)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pick_seed, rel_err
from oracle import bsms_oracle as ro

pytestmark = pytest.mark.gpu
FWD_TOL, BWD_TOL = 1e-5, 1e-5


@pytest.fixture(scope="module")
def eng():
    import bsms_gnn_amd as eng
    return eng


def dev(t):
    return t.cuda()


def load_sd(module, sd):
    module.load_state_dict({k: v for k, v in sd.items()}, strict=True)
    return module.cuda()


def random_graph(n, e, seed, hub=None):
    """Directed multigraph with isolated targets (degree-0 rows) and optionally one very high degree row."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e)
    dst = rng.integers(0, max(1, n - n // 8), e)  # the last n/8 nodes receive nothing
    if hub is not None:
        dst[: e // 4] = hub                     # a row with degree >= 64
    return torch.tensor(np.stack([src, dst]), dtype=torch.int64)


# ------------------------------------------------------------------------------------ plan / indices
def test_plan_matches_stable_sort(eng):
    g = random_graph(97, 1000, 0, hub=5)
    plan = eng.LevelPlan(g.cuda(), 97)
    rp, src, perm, trp = plan.export()
    order = np.argsort(g[1].numpy(), kind="stable")
    assert np.array_equal(perm, order.astype(np.int32))              # bit-exact indexing
    assert np.array_equal(src, g[0].numpy()[order].astype(np.int32))
    assert np.array_equal(rp, np.concatenate([[0], np.cumsum(np.bincount(g[1].numpy(), minlength=97))]).astype(np.int32))
    assert np.array_equal(trp, np.concatenate([[0], np.cumsum(np.bincount(g[0].numpy(), minlength=97))]).astype(np.int32))
    with pytest.raises(eng._abi.BsmsError):
        eng.LevelPlan(torch.tensor([[0, 5], [1, 2]]).cuda(), 3)      # out-of-range endpoint


def _plan_arrays_ok(plan, g, n):
    rp, src, perm, trp = plan.export()
    order = np.argsort(g[1].numpy(), kind="stable")
    return (np.array_equal(perm, order.astype(np.int32)) and np.array_equal(src, g[0].numpy()[order].astype(np.int32))
            and np.array_equal(rp, np.concatenate([[0], np.cumsum(np.bincount(g[1].numpy(), minlength=n))]).astype(np.int32))
            and np.array_equal(trp, np.concatenate([[0], np.cumsum(np.bincount(g[0].numpy(), minlength=n))]).astype(np.int32)))


def test_plans_of_new_meshes_recycled_blocks_and_concurrent_builds(eng):
    """Variable meshes bring new plans every batch: device blocks of destroyed plans are recycled (no hipFree), the
    levels of a batch are built concurrently, interned index tensors hand their host original to the builder.  Every
    plan must still be the stable sort of ITS edge list, and the kernels must read the recycled block's new content."""
    import gc
    from bsms_gnn_amd import graph
    eng.clear_plan_cache()
    rng = np.random.default_rng(3)
    for rnd in range(6):                                   # sizes wander by +-20 %: recycled blocks of other capacities
        sizes = [(int(97 * f), int(1000 * f)) for f in rng.uniform(0.8, 1.2, 5)]
        gs = [random_graph(n, e, 100 * rnd + k, hub=3) for k, (n, e) in enumerate(sizes)]
        ids = [torch.arange(0, n, 2) for n, _ in sizes]
        dev_g = [graph.intern_index(g, "cuda") for g in gs]
        dev_i = [graph.intern_index(i, "cuda") for i in ids]
        assert all(getattr(t, "_bsms_host", None) is not None for t in dev_g + dev_i)
        before = eng.LevelPlan.constructed
        plans = graph.plans_for([(g, n, i) for g, (n, _), i in zip(dev_g, sizes, dev_i)])
        assert eng.LevelPlan.constructed == before + 5
        assert graph.plans_for([(g, n, i) for g, (n, _), i in zip(dev_g, sizes, dev_i)]) == plans   # cached now
        for plan, g, (n, e), i in zip(plans, gs, sizes, ids):
            assert _plan_arrays_ok(plan, g, n) and plan.Nk == i.numel()
            src, out = dev(torch.randn(2, e, 8)), torch.empty(2, n, 8, device="cuda")
            eng._abi.check(eng._abi.lib().bsms_segment_sum_fwd(plan.handle, src.data_ptr(), 2, 8, 0, out.data_ptr(),
                                                               torch.cuda.current_stream().cuda_stream), "segment_sum")
            assert torch.equal(out.cpu(), ro.scatter_sum(src.cpu(), g[1], -2, n))       # the kernels read the recycled block's new content
        del plans, plan
        eng.clear_plan_cache()
        gc.collect()
        torch.cuda.synchronize()
        graph._reap()
        assert not graph._GRAVE                              # retired after their event: blocks are back in the pool
    assert eng._abi.lib().bsms_plan_pool_trim() == 0


# ------------------------------------------------------------------------------------ A1 segment sum
@pytest.mark.parametrize("B,D", [(1, 1), (2, 3), (3, 8), (2, 32), (2, 128), (1, 256), (2, 36)])
def test_segment_sum_bit_exact(eng, B, D):
    n, e = 203, 3000
    g = random_graph(n, e, B * 100 + D, hub=7)
    torch.manual_seed(D)
    src = torch.randn(B, e, D)
    want = ro.scatter_sum(src, g[1], -2, n)
    x = dev(src).requires_grad_(True)
    got = eng.scatter_sum(x, dev(g[1]), dim=-2, dim_size=n)
    assert torch.equal(got.cpu(), want)                              # same summation order -> same bits
    cot = torch.randn(B, n, D)
    got.backward(dev(cot))
    assert torch.equal(x.grad.cpu(), cot[:, g[1]])                   # backward = gather by target
    # empty rows really are zero, 1-D and 2-D layouts
    assert float(got[:, n - n // 8:].abs().sum()) == 0.0
    assert torch.equal(eng.scatter_sum(dev(src[0]), dev(g[1]), dim=-2, dim_size=n).cpu(), want[0])
    assert torch.equal(eng.scatter_sum(dev(src[0, :, 0].contiguous()), dev(g[1]), dim=-1, dim_size=n).cpu(), want[0, :, 0])


def test_segment_sum_golden_and_degree(eng, graphs):
    z = load_golden("prims")
    for name in ("del64", "del300"):
        es, _ = graphs.levels(name)
        n0 = graphs.np(f"{name}/pos").shape[0]
        got = eng.scatter_sum(dev(z.t(f"{name}/scatter_src")), dev(es[0][1]), dim=-2, dim_size=n0)
        assert torch.equal(got.cpu(), z.t(f"{name}/scatter_out"))
        assert torch.equal(eng.degree(dev(es[0][0]), dtype=torch.float).cpu(), z.t(f"{name}/degree"))
    assert eng.degree(torch.tensor([0, 1, 0, 2, 0]).cuda(), dtype=torch.long).tolist() == [3, 1, 1]


def test_empty_and_ragged(eng):
    g = torch.zeros(2, 0, dtype=torch.int64)
    out = eng.scatter_sum(torch.zeros(2, 0, 32).cuda(), g[1].cuda(), dim=-2, dim_size=5)
    assert out.shape == (2, 5, 32) and float(out.abs().sum()) == 0.0
    g1 = torch.tensor([[0], [0]])
    assert torch.equal(eng.scatter_sum(torch.ones(1, 1, 4).cuda(), g1[1].cuda(), dim=-2, dim_size=1).cpu(), torch.ones(1, 1, 4))


# ------------------------------------------------------------------------------------ A2,A5-A8
@pytest.mark.parametrize("name", ["del64", "del300"])
def test_transitions_golden_bit_exact(eng, graphs, name):
    z = load_golden("prims")
    es, ids = graphs.levels(name)
    n0 = graphs.np(f"{name}/pos").shape[0]
    conv = eng.WeightedEdgeConv()
    w = torch.ones(n0, 1).cuda()
    for l in range(len(ids)):                                        # cal_ew chain exactly as BSGMP runs it
        ew, aw = conv.cal_ew(w, dev(es[l]))
        assert torch.equal(ew.cpu(), z.t(f"{name}/ew{l}")), l
        assert torch.equal(aw.cpu(), z.t(f"{name}/aggr_w{l}")), l
        w = aw[dev(ids[l])]
    g0, ew = dev(es[0]), dev(z.t(f"{name}/ew0"))
    x3, x2 = dev(z.t(f"{name}/x3")), dev(z.t(f"{name}/x2"))
    assert torch.equal(conv(x3, g0, ew).cpu(), z.t(f"{name}/conv_down3"))
    assert torch.equal(conv(x3, g0, ew, aggragating=False).cpu(), z.t(f"{name}/conv_up3"))
    assert torch.equal(conv(x2, g0, ew).cpu(), z.t(f"{name}/conv_down2"))
    assert torch.equal(conv(x2, g0, ew, aggragating=False).cpu(), z.t(f"{name}/conv_up2"))
    un = eng.Unpool()(dev(z.t(f"{name}/coarse3")), n0, dev(ids[0]))
    assert torch.equal(un.cpu(), z.t(f"{name}/unpool3"))
    # fused restrict / prolong == the unfused sequences
    from bsms_gnn_amd.ops import _edge_conv
    plan = eng.plan_for(g0, n0, dev(ids[0]))
    assert torch.equal(_edge_conv(x3, ew, plan, True, True).cpu(), z.t(f"{name}/restrict3"))
    assert torch.equal(_edge_conv(dev(z.t(f"{name}/coarse3")), ew, plan, False, True).cpu(), z.t(f"{name}/prolong3"))
    # positions (D = 2) through the scalar path
    pos = torch.tensor(graphs.np(f"{name}/pos")[:, :2], dtype=torch.float32)
    assert torch.equal(conv(dev(pos), g0, ew).cpu(), ro.edge_conv(pos, es[0], z.t(f"{name}/ew0")))


def test_transition_gradients_and_adjoint(eng, graphs):
    es, ids = graphs.levels("del300")
    n0, nk = 300, ids[0].numel()
    from bsms_gnn_amd.ops import _edge_conv
    ew_cpu, _ = ro.cal_ew(torch.ones(n0, 1), es[0])
    plan = eng.plan_for(dev(es[0]), n0, dev(ids[0]))
    torch.manual_seed(3)
    h = torch.randn(2, n0, 32, requires_grad=True)
    c = torch.randn(2, nk, 32, requires_grad=True)
    hd, cd = dev(h.detach()).requires_grad_(True), dev(c.detach()).requires_grad_(True)
    r_gpu = _edge_conv(hd, dev(ew_cpu), plan, True, True)
    p_gpu = _edge_conv(cd, dev(ew_cpu), plan, False, True)
    r_cpu = ro.edge_conv(h, es[0], ew_cpu)[:, ids[0]]
    p_cpu = ro.edge_conv(ro.unpool(c, n0, ids[0]), es[0], ew_cpu, False)
    assert torch.equal(r_gpu.cpu(), r_cpu.detach()) and torch.equal(p_gpu.cpu(), p_cpu.detach())
    cot_r, cot_p = torch.randn_like(r_cpu), torch.randn_like(p_cpu)
    (r_gpu * dev(cot_r)).sum().backward()
    (p_gpu * dev(cot_p)).sum().backward()
    (r_cpu * cot_r).sum().backward()
    (p_cpu * cot_p).sum().backward()
    assert rel_err(hd.grad.cpu(), h.grad) < 1e-6 and rel_err(cd.grad.cpu(), c.grad) < 1e-6
    lhs = float((r_gpu.double() * cd.detach().double()).sum())          # <R h, c> == <h, P c>
    rhs = float((hd.detach().double() * p_gpu.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)


@pytest.mark.parametrize("name", ["del64", "del300", "surf200"])
def test_bound_edge_weights_transitions_bit_identical(eng, graphs, name):
    """bsms_plan_bind_edge_weights: the pooled transitions with weights gathered once into compact slot order (kept-row CSR
    for restrict, by-source CSR without the dropped targets for prolong) perform the same additions in the same order as
    the index-chasing path -- restrict, prolong and both adjoint shapes, features (D = 128, 32) and positions (D = 2, 3:
    scalar kernel), with a fused addend; a NEW weight tensor falls back to the generic path and leaves the binding alone, a
    re-pooled plan unbinds."""
    from bsms_gnn_amd import _abi
    from bsms_gnn_amd.ops import _stream
    es, ids = graphs.levels(name)
    n0, nk = graphs.np(f"{name}/pos").shape[0], ids[0].numel()
    L = _abi.lib()
    plan = eng.LevelPlan(dev(es[0]), n0, ids=dev(ids[0]))
    ew = dev(ro.cal_ew(torch.ones(n0, 1), es[0])[0])
    torch.manual_seed(7)

    def run(w, D, B=2):
        fine, coarse = torch.randn(B, n0, D, device="cuda"), torch.randn(B, nk, D, device="cuda")
        out = []
        for agg, x, n_out in ((1, fine, nk), (0, coarse, n0)):
            y = torch.empty(B, n_out, D, device="cuda")
            _abi.check(L.bsms_edge_conv(plan.handle, x.data_ptr(), B, D, w.data_ptr(), agg, 1, y.data_ptr(), _stream()), "edge_conv")
            out.append(y)
        return fine, coarse, out

    for D in (128, 32, 2, 3):
        torch.manual_seed(D)
        _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, None, _stream()), "unbind")
        _, _, slow = run(ew, D)
        torch.manual_seed(D)
        _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, ew.data_ptr(), _stream()), "bind")
        _, _, fast = run(ew, D)
        assert torch.equal(slow[0], fast[0]) and torch.equal(slow[1], fast[1]), D
        other = ew * 0.5                                   # another tensor: not the bound pointer -> generic path, its own values
        torch.manual_seed(D)
        _, _, half = run(other, D)
        torch.manual_seed(D)
        _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, other.data_ptr(), _stream()), "rebind")
        _, _, half_fast = run(other, D)
        assert torch.equal(half[0], half_fast[0]) and torch.equal(half[1], half_fast[1]), D
        torch.manual_seed(D)   # ... and a plan bound to `ew` KEEPS that binding (captured graphs have its copies baked in)
        _, _, again = run(ew, D)
        assert torch.equal(again[0], fast[0]) and torch.equal(again[1], fast[1]), D
    # the whole U-Net with bound weights (BSGMP.prepare binds them) == the golden output is covered by test_bsgmp_golden;
    # re-pooling unbinds: results follow the new pool
    keep = ids[0][: max(2, nk // 2)]
    plan.set_pool(dev(keep))
    x = torch.randn(1, n0, 32, device="cuda")
    y = torch.empty(1, keep.numel(), 32, device="cuda")
    _abi.check(L.bsms_edge_conv(plan.handle, x.data_ptr(), 1, 32, ew.data_ptr(), 1, 1, y.data_ptr(), _stream()), "edge_conv")
    want = ro.edge_conv(x.cpu(), es[0], ew.cpu())[:, keep]
    assert torch.equal(y.cpu(), want)
    _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, ew.data_ptr(), _stream()), "bind")
    y2 = torch.empty_like(y)
    _abi.check(L.bsms_edge_conv(plan.handle, x.data_ptr(), 1, 32, ew.data_ptr(), 1, 1, y2.data_ptr(), _stream()), "edge_conv")
    assert torch.equal(y2.cpu(), want)


def test_bound_edge_weights_lifetime_and_refresh(eng, graphs):
    """the binding is by ADDRESS.  (1) bsms_plan_bound_edge_weights reports what a plan is bound to; (2) the
    Python wrapper (ops._bind_edge_weights) keeps exactly the BOUND tensor alive -- a bind call the library declined (the plan is
    bound elsewhere) must not replace it, or the allocator could recycle the bound address for other weights; (3) a rebuilt
    chain with the SAME content gets the bound tensor back and stays on the compact lists; (4) binding the bound pointer again
    is a no-op, also after an in-place change -- bind NULL, then bind again to refresh (include/bsms_hip.h)."""
    from bsms_gnn_amd import _abi
    from bsms_gnn_amd.ops import _bind_edge_weights, _stream
    es, ids = graphs.levels("del300")
    n0, nk = graphs.np("del300/pos").shape[0], ids[0].numel()
    L = _abi.lib()
    plan = eng.LevelPlan(dev(es[0]), n0, ids=dev(ids[0]))
    assert not L.bsms_plan_bound_edge_weights(plan.handle)
    ew = dev(ro.cal_ew(torch.ones(n0, 1), es[0])[0])
    assert _bind_edge_weights(plan, ew) is ew and L.bsms_plan_bound_edge_weights(plan.handle) == ew.data_ptr() and plan._ew_bound is ew
    other = ew * 0.5                                        # different content: declined, the held tensor stays
    assert _bind_edge_weights(plan, other) is other
    assert L.bsms_plan_bound_edge_weights(plan.handle) == ew.data_ptr() and plan._ew_bound is ew
    same = ew.clone()                                       # a rebuilt chain: new tensor, same content -> the bound one is handed back
    assert _bind_edge_weights(plan, same) is ew and plan._ew_bound is ew

    x_cpu = torch.arange(n0 * 4, dtype=torch.float32).reshape(1, n0, 4).sin()   # ONE input for both sides (sin differs in the last bit between devices)
    x = x_cpu.cuda()

    def restrict(w):
        y = torch.empty(1, nk, 4, device="cuda")
        _abi.check(L.bsms_edge_conv(plan.handle, x.data_ptr(), 1, 4, w.data_ptr(), 1, 1, y.data_ptr(), _stream()), "edge_conv")
        return y

    y0 = restrict(ew)
    ew.mul_(2.0)                                            # in-place change of the bound tensor
    _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, ew.data_ptr(), _stream()), "rebind same pointer")
    assert torch.equal(restrict(ew), y0)                    # documented: a no-op, the gathered copies are the old values
    _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, None, _stream()), "unbind")
    assert not L.bsms_plan_bound_edge_weights(plan.handle)
    y_slow = restrict(ew)                                   # unbound: index chasing, the new values
    _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, ew.data_ptr(), _stream()), "bind")
    y_fast = restrict(ew)
    assert torch.equal(y_fast, y_slow) and not torch.equal(y_fast, y0)
    want = ro.edge_conv(x_cpu, es[0], ew.cpu())[:, ids[0]]
    assert torch.equal(y_fast.cpu(), want)


def test_inference_work_size_excludes_backward_sets(eng, graphs):
    """bsms_bsgmp_infer_work_bytes: the forward's part of the scratch layout only; the rollout path sizes its
    session buffer with it (covered end to end by the rollout tests)."""
    from bsms_gnn_amd import _abi
    es, ids = graphs.levels("del300")
    sizes = [graphs.np("del300/pos").shape[0]] + [int(i.numel()) for i in ids]
    plans = [eng.LevelPlan(dev(es[l]), sizes[l], ids=dev(ids[l]) if l < len(ids) else None) for l in range(len(es))]
    pl, keep = _abi.ptr_array([q.handle.value for q in plans])
    L = _abi.lib()
    full = L.bsms_bsgmp_work_bytes(pl, len(plans) - 1, 8, 128, 2, 3)
    infer = L.bsms_bsgmp_infer_work_bytes(pl, len(plans) - 1, 8, 128, 2, 3)
    assert 0 < infer < full and infer * 2 < full, (infer, full)


def test_cal_ew_degree_quirk(eng):
    """degree() has length max(g[0])+1: a trailing node without out-edges breaks w/deg."""
    g = torch.tensor([[0, 1], [1, 2]])  # node 2 never sends
    with pytest.raises(RuntimeError):
        eng.WeightedEdgeConv().cal_ew(torch.ones(3, 1).cuda(), g.cuda())
    with pytest.raises(RuntimeError):
        ro.cal_ew(torch.ones(3, 1), g)


def test_rank_and_device_errors(eng):
    g = torch.tensor([[0, 1], [1, 0]]).cuda()
    with pytest.raises(NotImplementedError):
        eng.GMP(32, 1, 2).cuda()(torch.zeros(1, 1, 2, 32).cuda(), g, torch.zeros(2, 2).cuda())
    with pytest.raises(NotImplementedError):
        eng.WeightedEdgeConv()(torch.zeros(1, 1, 2, 8).cuda(), g, torch.ones(2).cuda())
    with pytest.raises(eng._abi.BsmsError):                             # no CPU fallback
        eng.GMP(32, 1, 2)(torch.zeros(2, 32), g.cpu(), torch.zeros(2, 2))
    with pytest.raises(eng._abi.BsmsError):                             # unsupported latent width
        eng.GMP(48, 1, 2).cuda()(torch.zeros(2, 48).cuda(), g, torch.zeros(2, 2).cuda())


# ------------------------------------------------------------------------------------ A3 MLP
@pytest.mark.parametrize("in_dim,D,out_dim,H,ln,rows", [
    (3, 32, 32, 3, True, 300),      # encoder shape (C+1 -> D, LayerNorm)
    (4, 128, 128, 3, True, 1000),
    (128, 128, 3, 3, False, 1000),  # decoder shape (D -> C, no LayerNorm)
    (32, 32, 2, 2, False, 77),
    (64, 64, 64, 1, True, 129),     # generic D -> D
    (128, 128, 128, 3, True, 513),
    (256, 256, 256, 2, True, 200),
])
def test_mlp_against_oracle(eng, in_dim, D, out_dim, H, ln, rows):
    def build(seed):
        torch.manual_seed(seed)
        ref = ro.MLP(in_dim, D, out_dim, H, ln)
        x = torch.randn(2, rows, in_dim, requires_grad=True)
        return ref, (lambda: ref(x)), x

    seed = pick_seed(lambda s: build(s)[:2], first=in_dim + D + rows)   # stay away from ReLU kinks (conftest.KinkMargin)
    ref, _, x = build(seed)
    mine = load_sd(eng.MLP(in_dim, D, out_dim, H, ln), ref.state_dict())
    cot = torch.randn(2, rows, out_dim)
    y = ref(x)
    (y * cot).sum().backward()
    xd = dev(x.detach()).requires_grad_(in_dim == D)
    yd = mine(xd)
    (yd * dev(cot)).sum().backward()
    assert rel_err(yd.cpu(), y) < FWD_TOL
    if in_dim == D:
        assert rel_err(xd.grad.cpu(), x.grad) < BWD_TOL
    for (k, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        assert rel_err(pm.grad.cpu(), pr.grad) < BWD_TOL, k


# ------------------------------------------------------------------------------------ A4 GMP
@pytest.mark.parametrize("tag,D,p", [("d32p2", 32, 2), ("d128p2", 128, 2), ("d32p3", 32, 3)])
def test_gmp_golden(eng, graphs, tag, D, p):
    z = load_golden(f"gmp_{tag}")
    es, _ = graphs.levels(str(z.np("graph")))
    g = dev(es[0])
    gmp = load_sd(eng.GMP(D, 3, p), z.state_dict())
    for xk, pk, yk, dk, gk in (("x3", "pos3", "y33", "dx33", "g33/"), ("x3", "pos2", "y32", "dx32", None),
                               ("x2", "pos2", "y22", "dx22", "g22/")):
        gmp.zero_grad()
        x = dev(z.t(xk)).requires_grad_(True)
        cot = dev(z.t("cot3")) if xk == "x3" else dev(z.t("cot3")[0])
        y = gmp(x, g, dev(z.t(pk)))
        (y * cot).sum().backward()
        assert rel_err(y.cpu(), z.t(yk)) < FWD_TOL, (xk, pk)
        assert rel_err(x.grad.cpu(), z.t(dk)) < BWD_TOL, (xk, pk)
        if gk:
            for k, prm in gmp.named_parameters():
                assert rel_err(prm.grad.cpu(), z.t(gk + k)) < BWD_TOL, (k, xk, pk)


def test_gmp_degenerate_rows(eng):
    """Targets with no incoming edge (aggr = 0) and one with degree >= 64, ragged tile tails."""
    n, e, D = 150, 777, 64
    g = random_graph(n, e, 9, hub=3)
    def build(seed):
        torch.manual_seed(seed)
        ref = ro.GMP(D, 2, 2)
        x, pos = torch.randn(3, n, D, requires_grad=True), torch.rand(3, n, 2)
        return ref, (lambda: ref(x, g, pos)), x, pos

    seed = pick_seed(lambda s: build(s)[:2], first=2)
    ref, _, x, pos = build(seed)
    mine = load_sd(eng.GMP(D, 2, 2), ref.state_dict())
    y = ref(x, g, pos)
    y.square().sum().backward()
    xd = dev(x.detach()).requires_grad_(True)
    yd = mine(xd, dev(g), dev(pos))
    yd.square().sum().backward()
    assert rel_err(yd.cpu(), y) < FWD_TOL and rel_err(xd.grad.cpu(), x.grad) < BWD_TOL
    for (k, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        assert rel_err(pm.grad.cpu(), pr.grad) < BWD_TOL, k


@pytest.mark.parametrize("D,p,H", [(256, 3, 2), (64, 3, 3)])
def test_gmp_other_widths(eng, D, p, H):
    """Every latent width instantiates its own kernels (D = 256: 16 accumulator blocks, 155 KB of LDS per workgroup;
    the surface config of BASELINE.json): forward, input and parameter gradients against the oracle."""
    n, e = 210, 1500
    g = random_graph(n, e, 4)
    def build(seed):
        torch.manual_seed(seed)
        ref = ro.GMP(D, H, p)
        x, pos = torch.randn(2, n, D, requires_grad=True), torch.rand(2, n, p)
        return ref, (lambda: ref(x, g, pos)), x, pos

    seed = pick_seed(lambda s: build(s)[:2], first=5)
    ref, _, x, pos = build(seed)
    mine = load_sd(eng.GMP(D, H, p), ref.state_dict())
    y = ref(x, g, pos)
    y.square().sum().backward()
    xd = dev(x.detach()).requires_grad_(True)
    yd = mine(xd, dev(g), dev(pos))
    yd.square().sum().backward()
    assert rel_err(yd.cpu(), y) < FWD_TOL and rel_err(xd.grad.cpu(), x.grad) < BWD_TOL
    for (k, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        assert rel_err(pm.grad.cpu(), pr.grad) < BWD_TOL, k


def test_gmp_d256_wide_weight_gradient_tiles(eng):
    """D = 256 with enough rows (3 x 90 000 edge rows >= 262 144) for the weight gradients to take the 128 x 256 tiles of
    k_wgrad_wide (round 6: A read once per slab instead of twice).  7e7 ReLU inputs: dozens lie within 1e-8 of zero, so any two
    fp32 summation orders disagree on some masks and every edge-MLP gradient moves by ~2e-3 of its scale (measured; see
    test_gmp_launch_shapes).  Criterion = the three-way one of test_hip_fullsize.py: the engine's distance to an fp64 run is at
    most 1.5 x the fp32 oracle's (the larger of its all-threads and one-thread runs), worst tensor and median; a wrong tile
    would corrupt whole 16 x 16 blocks of dW (error ~1).  Forward three-way; input gradient within 1e-5 on all but a few rows."""
    D, p, H, n, e, B = 256, 3, 3, 6000, 45000, 2
    g = random_graph(n, e, 11)
    torch.manual_seed(23)
    ref = ro.GMP(D, H, p)
    x, pos, r = torch.randn(B, n, D), torch.rand(B, n, p), torch.randn(B, n, D)
    def run(dt):
        m = ro.GMP(D, H, p).to(dt)
        m.load_state_dict({k: v.to(dt) for k, v in ref.state_dict().items()})
        xx = x.to(dt).clone().requires_grad_(True)
        y = m(xx, g, pos.to(dt))
        (y * r.to(dt)).sum().backward()
        return y.detach(), xx.grad, {k: q.grad for k, q in m.named_parameters()}
    y64, gx64, gw64 = run(torch.float64)
    y32, _, gw32 = run(torch.float32)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        _, _, gw32_one = run(torch.float32)
    finally:
        torch.set_num_threads(nthreads)
    mine = load_sd(eng.GMP(D, H, p), ref.state_dict())
    xd = dev(x).requires_grad_(True)
    yd = mine(xd, dev(g), dev(pos))
    (yd * dev(r)).sum().backward()
    err = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    assert err(yd.detach(), y64) <= 3.0 * err(y32, y64) + 2e-6
    row_err = (xd.grad.double().cpu() - gx64).abs().amax(-1).reshape(-1) / float(gx64.abs().max())
    outliers = int((row_err > 1e-5).sum())
    assert outliers <= max(4, B * n // 500), (outliers, float(row_err.max()))
    errs = {k: err(q.grad, gw64[k]) for k, q in mine.named_parameters()}
    med = lambda d: sorted(d.values())[len(d) // 2]
    ref_n = {k: err(gw32[k], gw64[k]) for k in errs}
    ref_1 = {k: err(gw32_one[k], gw64[k]) for k in errs}
    print(f"\n[d256 wide] dW vs fp64: gpu worst {max(errs.values()):.2e} median {med(errs):.2e} | cpu32 all threads worst {max(ref_n.values()):.2e} "
          f"median {med(ref_n):.2e} | cpu32 one thread worst {max(ref_1.values()):.2e} median {med(ref_1):.2e}")
    assert max(errs.values()) <= 1.5 * max(max(ref_n.values()), max(ref_1.values())) + 1e-6, errs
    assert med(errs) <= 1.5 * max(med(ref_n), med(ref_1)) + 1e-6, errs


def test_split_products_are_fp32_accurate(eng):
    """The matrix cores multiply fp16 x 2 pieces of power-of-two-scaled fp32 operands, three partial products (chain.h;
    profiles/census/f16split.hip).  Against an fp64 computation the engine must be at least as accurate as plain fp32
    arithmetic on the same data (not merely within the 1e-5 parity tolerance): forward of a Linear-ReLU-Linear-LayerNorm
    stack, its input gradient and its weight gradients (the split-K kernel scales per tensor, wgrad.hip) -- also for
    inputs and upstream gradients whose magnitudes vary by orders of magnitude between rows."""
    torch.manual_seed(11)
    R, D = 4096, 128
    ref = ro.MLP(D, D, D, 1, True)
    sd = ref.state_dict()
    for scale_rows in (False, True):
        x = torch.randn(R, D)
        r = torch.randn(R, D)
        if scale_rows:   # rows spanning 1e-4 .. 1e3 in the input, 1e-8 .. 1 in the upstream gradient
            x = x * torch.logspace(-4, 3, R).unsqueeze(1)
            r = r * torch.logspace(-8, 0, R)[torch.randperm(R)].unsqueeze(1)
        def run(dt):
            ws = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
            xx = x.to(dt).clone().requires_grad_(True)
            h = torch.relu(xx @ ws["seq.0.weight"].T + ws["seq.0.bias"])
            z = h @ ws["seq.2.weight"].T + ws["seq.2.bias"]
            y = torch.nn.functional.layer_norm(z, (D,))
            (y * r.to(dt)).sum().backward()
            return y.detach(), xx.grad, {k: v.grad for k, v in ws.items()}
        y64, gx64, gw64 = run(torch.float64)
        y32, gx32, gw32 = run(torch.float32)
        mine = load_sd(eng.MLP(D, D, D, 1, True), sd)
        xd = dev(x).requires_grad_(True)
        yd = mine(xd)
        (yd * dev(r)).sum().backward()
        got_w = dict(mine.named_parameters())
        def err(a, b):   # error relative to the tensor scale
            return float((a.double().cpu() - b).abs().max() / b.abs().max())
        assert err(yd.detach(), y64) <= 2.0 * err(y32, y64) + 1e-7
        assert err(xd.grad, gx64) <= 2.0 * err(gx32, gx64) + 1e-7, (scale_rows, err(xd.grad, gx64), err(gx32, gx64))
        for k in gw64:
            e_mine, e_32 = err(got_w[k].grad, gw64[k]), err(gw32[k], gw64[k])
            assert e_mine <= 2.0 * e_32 + 1e-7, (scale_rows, k, e_mine, e_32)
        assert rel_err(yd.detach().cpu(), y64.float()) < 1e-6


def test_weight_gradient_precision_per_column(eng):
    """Weight gradients reduce over the rows, so the chain kernels' per-row scales cannot be used.  NODE-level jobs (bsms_mlp_bwd,
    the node MLP and the projections of a GMP block: operands may be caller tensors) run the range-free three-way bf16 split
    (wgrad.hip: BF3, include/bsms_hip.h): a feature of the Linear's input at 1e-3, 1e-5 and 1e-7 of the tensor maximum must be as
    accurate as fp32 arithmetic.  Errors are measured PER COLUMN of dW (a test that normalises by the whole tensor cannot see a
    degraded small column -- ADVICE round 3); VERDICT round 4 item 2: <= 2x the fp32 error down to 1e-7."""
    torch.manual_seed(12)
    R, D = 8192, 128
    ref = ro.MLP(D, D, D, 1, True)
    sd = ref.state_dict()
    for ratio in (1e-3, 1e-5, 1e-7):
        x = torch.randn(R, D)
        small = [3, 64, 127]
        x[:, small] *= ratio
        r = torch.randn(R, D)
        def run(dt):
            ws = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
            h = torch.relu(x.to(dt) @ ws["seq.0.weight"].T + ws["seq.0.bias"])
            y = torch.nn.functional.layer_norm(h @ ws["seq.2.weight"].T + ws["seq.2.bias"], (D,))
            (y * r.to(dt)).sum().backward()
            return ws["seq.0.weight"].grad
        g64, g32 = run(torch.float64), run(torch.float32)
        mine = load_sd(eng.MLP(D, D, D, 1, True), sd)
        (mine(dev(x)) * dev(r)).sum().backward()
        got = dict(mine.named_parameters())["seq.0.weight"].grad.double().cpu()
        col_err = lambda a: ((a.double() - g64).abs().max(dim=0).values / g64.abs().max(dim=0).values)
        e_mine, e_32 = col_err(got), col_err(g32)
        big = [c for c in range(D) if c not in small]
        assert float(e_mine[big].max()) <= 2.0 * float(e_32[big].max()) + 1e-7, (ratio, float(e_mine[big].max()), float(e_32[big].max()))
        assert float(e_mine[small].max()) <= 2.0 * float(e_32[small].max()) + 1e-7, (ratio, e_mine[small], e_32[small])
        print(f"[wgrad per column, small features at {ratio:g} of the maximum] worst column error: engine {float(e_mine[small].max()):.2e} "
              f"(fp32 {float(e_32[small].max()):.2e}); other columns {float(e_mine[big].max()):.2e} (fp32 {float(e_32[big].max()):.2e})")


def test_edge_weight_gradient_envelope(eng):
    """EDGE-level weight gradients of the fp32 GMP keep fp16 x 2 pieces with one scale per operand tensor (include/bsms_hip.h: a
    value within 2^-18 of its tensor's maximum keeps 22 bits).  Their operands are the edge MLP's own activations / layer
    gradients; a degraded column needs an edge unit whose activation is uniformly tiny.  Built here on purpose: row 5 of the
    second edge Linear scaled by 1e-4 (inside the window: as accurate as fp32) and by 1e-7 (outside: the documented loss, still
    <= 1e-3 of that column's own scale); the other columns stay at fp32 accuracy in both cases."""
    torch.manual_seed(3)
    n, D, B = 300, 128, 2
    pts = np.random.default_rng(5).random((n, 2))
    from scipy.spatial import Delaunay
    flat = torch.tensor(eng.to_flat_edge(Delaunay(pts).simplices.astype(np.int64), "tri"))
    pos = torch.tensor(pts, dtype=torch.float32).unsqueeze(0).repeat(B, 1, 1)
    x = torch.randn(B, n, D)
    r = torch.randn(B, n, D)
    col_err = lambda a, g64: ((a.double() - g64).abs().max(dim=0).values / g64.abs().max(dim=0).values)
    for ratio, bound in ((1e-4, None), (1e-7, 1e-3)):
        for seed in range(40):   # a model whose fp32 and fp64 runs take the same side of every ReLU (else a flipped unit is the error)
            torch.manual_seed(100 + seed)
            ref = ro.GMP(D, 3, 2)
            with torch.no_grad():
                ref.mlp_edge.seq[2].weight[5] *= ratio
                ref.mlp_edge.seq[2].bias[5] *= ratio
            sd = ref.state_dict()
            def run(dt):
                m = ro.GMP(D, 3, 2).to(dt)
                m.load_state_dict({k: v.to(dt) for k, v in sd.items()})
                (m(x.to(dt), flat, pos.to(dt)) * r.to(dt)).sum().backward()
                return m.mlp_edge.seq[4].weight.grad     # dW of the third edge Linear: its INPUT column 5 is the tiny activation
            g64, g32 = run(torch.float64), run(torch.float32)
            if float(col_err(g32, g64).max()) < 3e-6:
                break
        else:
            pytest.skip("no kink-free seed")
        mine = load_sd(eng.GMP(D, 3, 2), sd)
        (mine(dev(x), dev(flat), dev(pos)) * dev(r)).sum().backward()
        got = dict(mine.named_parameters())["mlp_edge.seq.4.weight"].grad.double().cpu()
        e_mine, e_32 = col_err(got, g64), col_err(g32, g64)
        big = [c for c in range(D) if c != 5]
        assert float(e_mine[big].max()) <= 2.0 * float(e_32[big].max()) + 1e-6, (ratio, float(e_mine[big].max()), float(e_32[big].max()))
        if bound is None:
            assert float(e_mine[5]) <= 2.0 * float(e_32[5]) + 1e-6, (ratio, float(e_mine[5]), float(e_32[5]))
        else:
            assert float(e_mine[5]) <= bound, (ratio, float(e_mine[5]), float(e_32[5]))
        print(f"[edge wgrad envelope, activation column at {ratio:g}] that column: engine {float(e_mine[5]):.2e} (fp32 {float(e_32[5]):.2e}); "
              f"others {float(e_mine[big].max()):.2e} (fp32 {float(e_32[big].max()):.2e})")


def test_feature_split_kernels_equal_the_ring_kernels(eng):
    """Launches of at most a few thousand rows take the feature-split chain kernels (csrc/chain.hip: k_fs_fwd: the four waves
    of a workgroup share a 16-row tile by OUTPUT FEATURES, weights straight from L2), larger ones the persistent ring
    kernels.  Every row is processed independently with the same arithmetic in the same order, so the first rows of a large
    launch must equal a small launch of the same rows BIT FOR BIT: forward (training and inference), input gradient; the
    weight gradients of the two paths agree to round-off.  MLP shapes of the path (encoder, decoder, D -> D with
    LayerNorm) and a whole GMP block at batch 1 against the same sample inside a batch of 6."""
    torch.manual_seed(21)
    D, small, big = 128, 3000, 30000      # 3000 rows: feature-split forward (<= 6144) AND backward (<= 3072)
    for in_dim, out_dim, ln in ((3, D, True), (D, D, True), (D, 3, False)):
        mlp = eng.MLP(in_dim, D, out_dim, 3, ln).cuda()
        x = torch.randn(big, in_dim, device="cuda")
        with torch.no_grad():
            y_big, y_small = mlp(x), mlp(x[:small].contiguous())
        assert torch.equal(y_big[:small], y_small), (in_dim, out_dim)
        xs = x[:small].clone().requires_grad_(in_dim == D)
        xb = x.clone().requires_grad_(in_dim == D)
        ys, yb = mlp(xs), mlp(xb)
        assert torch.equal(ys.detach(), y_small) and torch.equal(yb.detach()[:small], y_small)     # training forward == inference forward
        r = torch.randn(big, out_dim, device="cuda")
        (ys * r[:small]).sum().backward()
        gs = {k: q.grad.clone() for k, q in mlp.named_parameters()}
        mlp.zero_grad(set_to_none=True)
        (yb * torch.cat([r[:small], torch.zeros(big - small, out_dim, device="cuda")])).sum().backward()   # only the shared rows carry gradient
        if in_dim == D:
            assert torch.equal(xs.grad, xb.grad[:small]), (in_dim, out_dim)       # k_fs_bwd == k_chain_bwd bit for bit
        for k, q in mlp.named_parameters():
            assert rel_err(q.grad, gs[k]) < 2e-6, (in_dim, out_dim, k)            # weight gradients: other split-K slabs, round-off only
    # a GMP block: node-level chains of B = 1 (3000 rows) are feature-split, of B = 6 (18000 rows) ring kernels
    n, e = 3000, 18000
    g = random_graph(n, e, 5)
    gmp = eng.GMP(D, 3, 2).cuda()
    x6, pos6 = torch.randn(6, n, D, device="cuda"), torch.rand(6, n, 2, device="cuda")
    with torch.no_grad():
        y6 = gmp(x6, dev(g), pos6)
        y1 = gmp(x6[2:3].contiguous(), dev(g), pos6[2:3].contiguous())
    assert torch.equal(y6[2:3], y1)
    # training: forward equal, input gradient equal bit for bit (row-independent chains), weight gradients to round-off
    def step(xx, pp):
        gmp.zero_grad(set_to_none=True)
        xx = xx.clone().requires_grad_(True)
        y = gmp(xx, dev(g), pp)
        (y * y).sum().backward()
        return y.detach(), xx.grad, {k: q.grad.clone() for k, q in gmp.named_parameters()}
    ya, ga, wa = step(x6[2:3].contiguous(), pos6[2:3].contiguous())
    x_rep, pos_rep = x6[2:3].repeat(6, 1, 1).contiguous(), pos6[2:3].repeat(6, 1, 1).contiguous()
    yb, gb, wb = step(x_rep, pos_rep)
    assert torch.equal(ya, y1) and torch.equal(yb[4:5], ya)
    assert torch.equal(gb[4:5], ga)
    for k in wa:
        assert rel_err(wb[k] / 6, wa[k]) < 2e-6, k


def test_gmp_magnitude_range_zero_input_and_many_rows(eng):
    """The fp16 x 2 arithmetic scales every activation row, weight matrix and (in the weight gradients) operand tensor by a
    power of two taken from its magnitude (chain.h).  (1) A GMP block whose samples differ by five orders of magnitude and
    whose weights were rescaled layer by layer must stay as accurate as plain fp32 (three-way against fp64: input
    gradient and every weight gradient <= 2x the fp32 oracle's distance).  (2) All-zero inputs (every bound is 0) give
    finite results equal to the oracle's.  (3) A narrow MLP over 70 000 rows uses the persistent grid with four
    workgroups per CU: every compute wave's entry of the bound slots must be seen (chain.h: kBoundWidth)."""
    n, e, D, H, p = 180, 1300, 128, 3, 2
    g = random_graph(n, e, 21)
    torch.manual_seed(5)
    ref = ro.GMP(D, H, p)
    with torch.no_grad():   # weight matrices of very different scale (the per-matrix scale must follow)
        for k, (name, prm) in enumerate(ref.named_parameters()):
            if name.endswith("weight"):
                prm.mul_(10.0 ** ((k % 5) - 2))
    x = torch.randn(3, n, D) * torch.tensor([1e-3, 1.0, 50.0]).view(3, 1, 1)
    pos = torch.rand(3, n, p)
    r = torch.randn(3, n, D) * torch.tensor([1.0, 1e-4, 1e-2]).view(3, 1, 1)
    def run(dt):
        m = ro.GMP(D, H, p).to(dt)
        m.load_state_dict({k: v.to(dt) for k, v in ref.state_dict().items()})
        xx = x.to(dt).clone().requires_grad_(True)
        y = m(xx, g, pos.to(dt))
        (y * r.to(dt)).sum().backward()
        return y.detach(), xx.grad, {k: q.grad for k, q in m.named_parameters()}
    y64, gx64, gw64 = run(torch.float64)
    y32, gx32, gw32 = run(torch.float32)
    mine = load_sd(eng.GMP(D, H, p), ref.state_dict())
    xd = dev(x).requires_grad_(True)
    yd = mine(xd, dev(g), dev(pos))
    (yd * dev(r)).sum().backward()
    err = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    assert err(yd.detach(), y64) <= 2.0 * err(y32, y64) + 1e-7
    assert err(xd.grad, gx64) <= 2.0 * err(gx32, gx64) + 1e-7
    for k, q in mine.named_parameters():
        assert err(q.grad, gw64[k]) <= 2.0 * err(gw32[k], gw64[k]) + 1e-7, (k, err(q.grad, gw64[k]), err(gw32[k], gw64[k]))
    # (2) zero input (every input bound is 0; with the loss sum(y^2) behind a LayerNorm the gradients are differences of
    # nearly equal numbers, ~1e-4 from fp64 in ANY fp32 arithmetic): finite, forward equal to the oracle, gradients
    # three-way like above
    x = torch.zeros(2, n, D)
    r = None
    def run0(dt):
        m = ro.GMP(D, H, p).to(dt)
        m.load_state_dict({k: v.to(dt) for k, v in ref.state_dict().items()})
        xx = x.to(dt).clone().requires_grad_(True)
        y = m(xx, g, pos[:2].to(dt))
        y.square().sum().backward()
        return y.detach(), xx.grad, {k: q.grad for k, q in m.named_parameters()}
    y64, gx64, gw64 = run0(torch.float64)
    y32, gx32, gw32 = run0(torch.float32)
    mine0 = load_sd(eng.GMP(D, H, p), ref.state_dict())
    x0d = dev(x).requires_grad_(True)
    y0d = mine0(x0d, dev(g), dev(pos[:2]))
    y0d.square().sum().backward()
    assert torch.isfinite(y0d).all() and torch.isfinite(x0d.grad).all()
    assert err(y0d.detach(), y64) <= 2.0 * err(y32, y64) + 1e-7 and err(x0d.grad, gx64) <= 2.0 * err(gx32, gx64) + 1e-7
    for k, q in mine0.named_parameters():
        assert torch.isfinite(q.grad).all() and err(q.grad, gw64[k]) <= 2.0 * err(gw32[k], gw64[k]) + 1e-6, (k, err(q.grad, gw64[k]), err(gw32[k], gw64[k]))
    # (3) many rows at a narrow width
    torch.manual_seed(7)
    R, Dn = 70000, 32
    refm = ro.MLP(Dn, Dn, Dn, 2, True)
    xm = torch.randn(R, Dn) * torch.logspace(-3, 2, R).unsqueeze(1)
    rm = torch.randn(R, Dn)
    def runm(dt):
        m = ro.MLP(Dn, Dn, Dn, 2, True).to(dt)
        m.load_state_dict({k: v.to(dt) for k, v in refm.state_dict().items()})
        xx = xm.to(dt).clone().requires_grad_(True)
        (m(xx) * rm.to(dt)).sum().backward()
        return xx.grad, {k: q.grad for k, q in m.named_parameters()}
    gx64, gw64 = runm(torch.float64)
    gx32, gw32 = runm(torch.float32)
    minem = load_sd(eng.MLP(Dn, Dn, Dn, 2, True), refm.state_dict())
    xmd = dev(xm).requires_grad_(True)
    (minem(xmd) * dev(rm)).sum().backward()
    assert err(xmd.grad, gx64) <= 2.0 * err(gx32, gx64) + 1e-7
    for k, q in minem.named_parameters():   # 3e-6: 70 000-term fp32 sums in slab order (torch sums pairwise); far inside the 1e-5 parity bar
        assert err(q.grad, gw64[k]) <= 2.0 * err(gw32[k], gw64[k]) + 3e-6, (k, err(q.grad, gw64[k]), err(gw32[k], gw64[k]))


@pytest.mark.parametrize("n,e,B", [(40, 90, 1), (300, 2300, 3), (900, 7000, 5), (2000, 16000, 4), (1200, 9000, 7)])
def test_gmp_launch_shapes(eng, n, e, B):
    """The chain launchers pick tile shapes by row count: 16-row blocks per wave (one or two), 4-7 compute waves, one or
    several rounds of workgroups, the single-round kernel variants, 3-6 ring slots, 1-2 loader waves (chain.hip:
    pick_stream, launch_edge_fwd / _bwd, chain_compute_waves).  Edge-row counts from 90 to 64 000 at D = 128 walk through
    all of them (35 000 rows: two row blocks per wave in the forward; 63 000 / 64 000: several rounds).
    Forward: three-way against fp64 (<= 3x the fp32 oracle's distance).  Gradients: at these sizes a few of the 10^6-10^7
    ReLU inputs lie within 1e-8 of zero (checked: 1.1e-8 for the second case), so any two fp32 arithmetics disagree on a
    handful of ReLU masks -- one flipped mask moves two node rows of the input gradient by ~1e-2 and the edge-MLP weight
    gradients by ~1e-3 (conftest.KinkMargin) -- whereas a wrong tile would corrupt whole 16-row blocks.  Hence: all but
    at most max(4, 0.2 %) node rows of the input gradient within 1e-5 of the gradient scale, every weight gradient within
    5e-3, the median over the weight tensors within 3e-5."""
    D, H, p = 128, 3, 2
    g = random_graph(n, e, 100 + n)
    torch.manual_seed(n)
    ref = ro.GMP(D, H, p)
    x, pos, r = torch.randn(B, n, D), torch.rand(B, n, p), torch.randn(B, n, D)
    def run(dt):
        m = ro.GMP(D, H, p).to(dt)
        m.load_state_dict({k: v.to(dt) for k, v in ref.state_dict().items()})
        xx = x.to(dt).clone().requires_grad_(True)
        y = m(xx, g, pos.to(dt))
        (y * r.to(dt)).sum().backward()
        return y.detach(), xx.grad, {k: q.grad for k, q in m.named_parameters()}
    y64, gx64, gw64 = run(torch.float64)
    y32, _, _ = run(torch.float32)
    mine = load_sd(eng.GMP(D, H, p), ref.state_dict())
    xd = dev(x).requires_grad_(True)
    yd = mine(xd, dev(g), dev(pos))
    (yd * dev(r)).sum().backward()
    err = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    assert err(yd.detach(), y64) <= 3.0 * err(y32, y64) + 2e-6
    row_err = (xd.grad.double().cpu() - gx64).abs().amax(-1).reshape(-1) / float(gx64.abs().max())
    outliers = int((row_err > 1e-5).sum())
    assert outliers <= max(4, B * n // 500), (outliers, float(row_err.max()))
    errs = {k: err(q.grad, gw64[k]) for k, q in mine.named_parameters()}
    assert max(errs.values()) < 5e-3, errs
    assert sorted(errs.values())[len(errs) // 2] < 3e-5, errs
    with torch.no_grad():   # the inference variants (no activation stores, no bounds) give the same forward
        assert torch.equal(mine(dev(x), dev(g), dev(pos)), yd.detach())


# ------------------------------------------------------------------------------------ A8,A9 BSGMP
@pytest.mark.parametrize("tag,D,p", [("line11", 32, 3), ("del300", 32, 2), ("del64_d128", 128, 2)])
def test_bsgmp_golden(eng, graphs, tag, D, p):
    z = load_golden(f"bsgmp_{tag}")
    L = int(z.np("depth"))
    es, ids = graphs.levels(str(z.np("graph")))
    net = load_sd(eng.BSGMP(L, D, 3, p), z.state_dict())
    assert set(net.state_dict()) == set(z.state_dict())                # checkpoint layout identical
    h = dev(z.t("h")).requires_grad_(True)
    y = net(h, [dev(i) for i in ids[:L]], [dev(e) for e in es[: L + 1]], dev(z.t("pos")))
    (y * dev(z.t("cot"))).sum().backward()
    assert rel_err(y.cpu(), z.t("y")) < FWD_TOL
    assert rel_err(h.grad.cpu(), z.t("dh")) < BWD_TOL
    for k, prm in net.named_parameters():
        assert rel_err(prm.grad.cpu(), z.t("g/" + k)) < BWD_TOL, k


# ------------------------------------------------------------------------------------ A10-A12, A16
def test_bsgmp_single_call_equals_module_tree(eng, graphs):
    """bsms_bsgmp_fwd/_bwd (one library call for the U-Net) sequences exactly the kernels the per-module Python tree
    launches: outputs, input gradient and every parameter gradient must be bit-identical."""
    from bsms_gnn_amd import ops
    z = load_golden("bsgmp_del300")
    L = int(z.np("depth"))
    es, ids = graphs.levels(str(z.np("graph")))
    net = load_sd(eng.BSGMP(L, 32, 3, 2), z.state_dict())
    res = {}
    for mode in (False, True):
        net.per_block = mode
        try:
            net.zero_grad()
            h = dev(z.t("h")).requires_grad_(True)
            y = net(h, [dev(i) for i in ids[:L]], [dev(e) for e in es[: L + 1]], dev(z.t("pos")))
            (y * dev(z.t("cot"))).sum().backward()
            with torch.no_grad():
                yi = net(h.detach(), [dev(i) for i in ids[:L]], [dev(e) for e in es[: L + 1]], dev(z.t("pos")))
            res[mode] = (y.detach().clone(), h.grad.clone(), [q.grad.clone() for q in net.parameters()], yi.clone())
        finally:
            net.per_block = False
    a, b = res[False], res[True]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    assert all(torch.equal(u, v) for u, v in zip(a[2], b[2]))
    assert rel_err(a[0].cpu(), z.t("y")) < FWD_TOL


def test_bsgmp_single_call_layouts(eng):
    """The one-call U-Net against the module tree on the layouts the reference accepts: depth 0 (bottom block only),
    2-D h with 2-D pos, 3-D h with a 2-D pos shared by the batch (ops/basic.py:87-88) -- bit-identical each time."""
    from bsms_gnn_amd import ops
    N, D = 120, 32
    src = np.arange(N); dst = (src + 1) % N
    flat = np.stack([np.concatenate([src, dst]), np.concatenate([dst, src])])
    pts = np.stack([np.cos(2 * np.pi * src / N), np.sin(2 * np.pi * src / N)], 1)
    _, m_es, m_ids = eng.BistrideMultiLayerGraph(flat, 2, N, pts).get_multi_layer_graphs()
    m_gs = [dev(torch.tensor(e, dtype=torch.int64)) for e in m_es]
    ids = [dev(torch.tensor(i, dtype=torch.int64)) for i in m_ids]
    pos2 = dev(torch.tensor(pts, dtype=torch.float32))
    torch.manual_seed(0)
    cases = [(eng.BSGMP(0, D, 2, 2), torch.randn(2, N, D), [], m_gs[:1], dev(torch.rand(2, N, 2))),
             (eng.BSGMP(2, D, 2, 2), torch.randn(N, D), ids, m_gs, pos2),
             (eng.BSGMP(2, D, 2, 2), torch.randn(3, N, D), ids, m_gs, pos2)]
    for net, h0, i_, g_, pos in cases:
        net = net.cuda()
        res = {}
        for mode in (False, True):
            net.per_block = mode
            try:
                net.zero_grad()
                h = dev(h0).requires_grad_(True)
                y = net(h, i_, g_, pos)
                y.square().sum().backward()
                res[mode] = (y.detach().clone(), h.grad.clone(), [q.grad.clone() for q in net.parameters()])
            finally:
                net.per_block = False
        a, b = res[False], res[True]
        assert a[0].shape == h0.shape and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert all(torch.equal(u, v) for u, v in zip(a[2], b[2]))


def test_simulator_step_and_rollout_golden(eng, graphs):
    z = load_golden("sim")
    es, ids = graphs.levels("del300")
    B = z.np("node_in").shape[0]
    sim = eng.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2))
    sd = z.state_dict()
    assert set(sd) == set(sim.state_dict())
    m_gs = [dev(e.unsqueeze(0).repeat(B, 1, 1)) for e in es]
    m_ids = [dev(i.unsqueeze(0).repeat(B, 1)) for i in ids]
    fresh = eng.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2)).cuda()
    for k in range(3):                                                # warm-up accumulates fp64 statistics
        out = fresh((dev(z.t(f"warm_in{k}")), dev(z.t(f"warm_tar{k}")), None, m_gs, m_ids), True, True)
        assert float(out.abs().sum()) == 0.0
    for k, v in fresh.state_dict().items():
        if "Normalizer" in k:
            assert v.dtype == torch.float64
            torch.testing.assert_close(v.cpu(), sd[k], rtol=1e-6, atol=0)   # fp32 batch means, different reduction order
    sim.load_state_dict(sd)
    sim = sim.cuda()
    node_in, tar, mask = dev(z.t("node_in")), dev(z.t("tar")), dev(z.t("mask"))
    pred = sim((node_in, tar, mask, m_gs, m_ids), True, False)
    loss = eng.masked_rmse(pred, tar, mask)
    loss.backward()
    assert rel_err(pred.cpu(), z.t("pred")) < FWD_TOL
    assert abs(float(loss) - float(z.t("loss"))) < 1e-5 * abs(float(z.t("loss")))
    for k, prm in sim.named_parameters():
        if prm.requires_grad:
            assert rel_err(prm.grad.cpu(), z.t("g/" + k)) < BWD_TOL, k
    dead = z.t("mask")[..., 0] == 0
    assert torch.equal(pred.detach().cpu()[dead], z.t("node_in")[..., :2][dead])   # masked nodes untouched (quirk 8)
    # 5-step autoregressive rollout (utils/rollout_utils.py:14-64)
    sim.zero_grad()
    with torch.no_grad():
        ic, rmask = dev(z.t("rollout_ic")), dev(z.t("rollout_mask"))
        g1, i1 = [dev(e.unsqueeze(0)) for e in es], [dev(i.unsqueeze(0)) for i in ids]
        cur, tail, frames = ic.clone(), ic[..., 2:].clone(), []
        for _ in range(5):
            pr = sim((cur, torch.zeros_like(cur), rmask, g1, i1), True, False)
            frames.append(pr[0])
            cur = torch.where(rmask == 0, ic, torch.cat([pr, tail], -1))
    assert rel_err(torch.stack(frames).cpu(), z.t("rollout")) < 5e-5
    # the product rollout engine (inference mode, HIP-graph replay) reproduces the same frames
    for use_graph in (False, True):
        res = torch.zeros(5, 300, 2, device="cuda")
        eng.rollout_one_traj(sim, ic, res, rmask, g1, i1, use_graph=use_graph)
        assert rel_err(res.cpu(), z.t("rollout")) < 5e-5, use_graph
        assert rel_err(res, torch.stack(frames)) < 2e-6, use_graph


def test_fused_glue_kernels_equal_the_torch_glue(eng, graphs):
    """bsms_sim_prologue / bsms_sim_epilogue (forward-only path of BSMS_Simulator) against the element-wise torch ops of
    `_forward` (the op-for-op mirror of models/model.py:127-164 and utils/normalizer.py): the fp64 normaliser
    arithmetic is reproduced exactly, so the predictions must be BIT-identical; then rollout_batch == B x rollout_one_traj."""
    z = load_golden("sim")
    es, ids = graphs.levels("del300")
    sim = eng.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2))
    sim.load_state_dict(z.state_dict())
    sim = sim.cuda()
    B = z.np("node_in").shape[0]
    m_gs = [dev(e.unsqueeze(0).repeat(B, 1, 1)) for e in es]
    m_ids = [dev(i.unsqueeze(0).repeat(B, 1)) for i in ids]
    node_in, mask = dev(z.t("node_in")), dev(z.t("mask"))
    with torch.no_grad():
        fast = sim((node_in, None, mask, m_gs, m_ids), True, False)                  # -> _infer
        slow = sim._forward([i[0] for i in m_ids], [g[0] for g in m_gs], node_in, mask)  # torch glue
    assert torch.equal(fast, slow)
    # degenerate statistics: zero variance -> std_eps, NaN variance -> nan_to_num
    sim._inputNormalizer._E_data_squared.data = sim._inputNormalizer._E_data.data ** 2
    sim._targetNormalizer._E_data_squared.data[0] = -1.0
    with torch.no_grad():
        fast = sim((node_in, None, mask, m_gs, m_ids), True, False)
        slow = sim._forward([i[0] for i in m_ids], [g[0] for g in m_gs], node_in, mask)
    assert torch.equal(fast, slow)
    sim.load_state_dict(z.state_dict())
    # batched rollout: three copies of one trajectory (one with a different IC) advance together
    ic, rmask = dev(z.t("rollout_ic")), dev(z.t("rollout_mask"))
    g1, i1 = [dev(e.unsqueeze(0)) for e in es], [dev(i.unsqueeze(0)) for i in ids]
    one = eng.rollout_one_traj(sim, ic, torch.zeros(4, 300, 2, device="cuda"), rmask, g1, i1)
    ic3 = torch.cat([ic, ic * 0.5, ic], 0)
    res = eng.rollout_batch(sim, ic3, torch.zeros(4, 3, 300, 2, device="cuda"), rmask.repeat(3, 1, 1),
                            [g.repeat(3, 1, 1) for g in g1], [i.repeat(3, 1) for i in i1])
    assert torch.equal(res[:, 0], one) and torch.equal(res[:, 2], one) and not torch.equal(res[:, 1], one)
    dead = rmask[0, :, 0] == 0
    assert torch.equal(res[-1, 1][dead], (ic * 0.5)[0, :, :2][dead])               # Dirichlet nodes keep their IC


def test_inference_mode_matches_training_forward(eng, graphs):
    """`saved` = NULL path (no activation stores) == the autograd forward, bit for bit."""
    es, _ = graphs.levels("del300")
    g = dev(es[0])
    torch.manual_seed(4)
    gmp = eng.GMP(64, 3, 2).cuda()
    x, pos = torch.randn(2, 300, 64, device="cuda"), torch.rand(2, 300, 2, device="cuda")
    y_train = gmp(x.clone().requires_grad_(True), g, pos)
    with torch.no_grad():
        y_inf = gmp(x, g, pos)
    assert torch.equal(y_train.detach(), y_inf)
    enc = eng.MLP(3, 64, 64, 3, True).cuda()
    xin = torch.randn(500, 3, device="cuda")
    with torch.no_grad():
        y0 = enc(xin)
    assert torch.equal(enc(xin).detach(), y0)


# ------------------------------------------------------------------------------------ A15
def test_block_diagonal_batch(eng, graphs):
    z = load_golden("blockdiag")
    net = load_sd(eng.BSGMP(2, 32, 3, 2), z.state_dict())
    m_gs = [dev(z.t(f"cat/e{l}")) for l in range(3)]
    m_ids = [dev(z.t(f"cat/ids{l}")) for l in range(2)]
    h = torch.cat([z.t("del64/h"), z.t("del300/h")])[None]
    pos = torch.cat([torch.tensor(graphs.np(f"{nm}/pos")[:, :2], dtype=torch.float32) for nm in ("del64", "del300")])[None]
    with torch.no_grad():
        y = net(dev(h), m_ids, m_gs, dev(pos)).cpu()
    assert rel_err(y, z.t("y_cat")) < FWD_TOL
    assert rel_err(y[0, :64], z.t("del64/y")) < FWD_TOL and rel_err(y[0, 64:], z.t("del300/y")) < FWD_TOL


def test_variable_mesh_branch(eng, graphs):
    """consistent_mesh=False: two DIFFERENT meshes collated block-diagonally (datasets/base.py:325-349 + PyG Batch)
    through BSMS_Simulator == each mesh alone through the CPU oracle."""
    cfg = ro.make_cfg(2, 32, 3, 2, 2)
    torch.manual_seed(5)
    ref = ro.BSMS_Simulator(cfg)
    samples, per_sample = [], []
    for nm in ("del64", "del300"):
        es, ids = graphs.levels(nm)
        n = graphs.np(f"{nm}/pos").shape[0]
        pos = torch.tensor(graphs.np(f"{nm}/pos")[:, :2], dtype=torch.float32)
        state, ntype = torch.randn(n, 2), (torch.rand(n, 1) < 0.1).float()
        x, y, mask = torch.cat([state, pos, ntype], -1), state + 0.1 * torch.randn(n, 2), (ntype == 0).float()
        sizes = [n] + [i.numel() for i in ids[:2]]
        samples.append([eng.LevelData(es[l], sizes[l], face=ids[l] if l < 2 else None, x=x if l == 0 else None,
                                      y=y if l == 0 else None, mask=mask if l == 0 else None) for l in range(3)])
        per_sample.append((x, y, mask, es[:3], ids[:2]))
    for x, y, mask, es, ids in per_sample:                                  # statistics from both meshes
        ref((x[None], y[None], mask[None], [e[None] for e in es], [i[None] for i in ids]), True, True)
    mine = eng.BSMS_Simulator(cfg)
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda()
    batch = [d.to("cuda") for d in eng.collate_variable_meshes(samples)]
    assert batch[0].num_nodes == 364 and batch[0].x.shape == (364, 5)
    with torch.no_grad():
        pred = mine(batch, False, False).cpu()
        off = 0
        for x, y, mask, es, ids in per_sample:
            want = ref((x[None], y[None], mask[None], [e[None] for e in es], [i[None] for i in ids]), True, False)
            assert rel_err(pred[:, off:off + x.shape[0]], want) < FWD_TOL
            off += x.shape[0]


# ------------------------------------------------------------------------------------ full size
def test_full_size_properties(eng):
    """BASELINE.json config sizes (airfoil-like, B=8, D=128): size-independent properties."""
    from bench import build_workload
    wl = build_workload("airfoil", batch=8, device="cuda")
    g0, ids0, n0 = wl["m_gs"][0][0], wl["m_ids"][0][0], wl["levels"][0][0]
    plan = eng.plan_for(g0, n0, ids0)
    from bsms_gnn_amd.ops import _edge_conv
    torch.manual_seed(0)
    a, b = torch.randn(8, g0.shape[1], 128, device="cuda"), torch.randn(8, g0.shape[1], 128, device="cuda")
    s = lambda t: eng.scatter_sum(t, g0[1], dim=-2, dim_size=n0)
    assert rel_err(s(a + b), s(a) + s(b)) < 1e-6                       # linearity
    assert abs(float(s(a).double().sum()) - float(a.double().sum())) < 1e-6 * float(a.abs().double().sum())  # mass
    ew, _ = eng.WeightedEdgeConv().cal_ew(torch.ones(n0, 1, device="cuda"), g0)
    ones = eng.scatter_sum(ew, g0[1], dim=-1, dim_size=n0)
    assert float((ones - 1).abs().max()) < 1e-5                         # weights into a node sum to 1
    h, c = torch.randn(8, n0, 128, device="cuda"), torch.randn(8, ids0.numel(), 128, device="cuda")
    lhs = float((_edge_conv(h, ew, plan, True, True).double() * c.double()).sum())
    rhs = float((h.double() * _edge_conv(c, ew, plan, False, True).double()).sum())
    assert abs(lhs - rhs) < 1e-6 * abs(lhs)                             # restrict / prolong adjoint
    # batch independence: dense batch == per-sample results
    net = eng.BSGMP(2, 128, 3, 2).cuda()
    with torch.no_grad():
        x = torch.randn(2, n0, 128, device="cuda")
        pos = wl["node_in"][:2, :, 3:5].contiguous()
        both = net(x, [i[0] for i in wl["m_ids"][:2]], [g[0] for g in wl["m_gs"][:3]], pos)
        one = net(x[1:], [i[0] for i in wl["m_ids"][:2]], [g[0] for g in wl["m_gs"][:3]], pos[1:])
    assert torch.equal(both[1:], one)


def test_cylinder_blockdiag_equals_dense_at_full_size(eng):
    """BASELINE configs[1] size (cylinder-like, 1885 nodes, 4 levels, D=128), B=3: the same mesh three times as a
    block-diagonal batch (variable-mesh path) == the dense consistent-mesh batch, forward and loss."""
    from bench import WORKLOADS, build_mesh, make_cfg
    w = WORKLOADS["cylinder"]
    pts, m_es, m_ids = build_mesh("cylinder")
    n, c, B = w["nodes"], w["out_dim"], 3
    torch.manual_seed(1)
    sim = eng.BSMS_Simulator(make_cfg(w)).cuda()
    state, target = torch.randn(B, n, c), torch.randn(B, n, c)
    pos = torch.tensor(pts, dtype=torch.float32)
    node_in = torch.cat([state, pos.expand(B, n, 2), torch.zeros(B, n, 1)], -1)
    mask = torch.ones(B, n, 1)
    dense = (node_in.cuda(), target.cuda(), mask.cuda(), [torch.tensor(e).unsqueeze(0).repeat(B, 1, 1).cuda() for e in m_es],
             [torch.tensor(i).unsqueeze(0).repeat(B, 1).cuda() for i in m_ids])
    sizes = [n] + [len(i) for i in m_ids]
    samples = [[eng.LevelData(torch.tensor(m_es[l]), sizes[l], face=torch.tensor(m_ids[l]) if l < w["levels"] else None,
                              x=node_in[b] if l == 0 else None, y=target[b] if l == 0 else None,
                              mask=mask[b] if l == 0 else None) for l in range(w["levels"] + 1)] for b in range(B)]
    blk = [d.to("cuda") for d in eng.collate_variable_meshes(samples)]
    sim(dense, True, True)
    with torch.no_grad():
        p_dense = sim(dense, True, False)
        p_blk = sim(blk, False, False)
    assert p_blk.shape == (1, B * n, c)
    assert rel_err(p_blk.view(B, n, c), p_dense) < 2e-6
    l_dense = eng.masked_rmse(p_dense, dense[1], dense[2])
    l_blk = eng.masked_rmse(p_blk, blk[0].y.unsqueeze(0), blk[0].mask.unsqueeze(0))
    assert abs(float(l_dense) - float(l_blk)) < 1e-6 * float(l_dense)
