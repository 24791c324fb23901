"""CPU: pin the oracle (oracle/*.py) against golden vectors generated from the reference itself.

Integers are compared exactly; fp32 with rtol 1e-5 / atol 1e-6 -- the oracle
follows the reference's op order, so most of these are in fact bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import bistride_oracle as bo
from oracle import bsms_oracle as ro

RTOL, ATOL = 1e-5, 1e-6
torch.set_num_threads(1)


def close(a, b, rtol=RTOL, atol=ATOL):
    torch.testing.assert_close(torch.as_tensor(a), torch.as_tensor(b), rtol=rtol, atol=atol)


# ------------------------------------------------------------------ integer side (A13, A14)
def test_line11_known_answer(graphs):
    """The reference's own example: m_ids=[[1,3,5,7,9],[1,3]] (bsms_graph_wrapper.py:157-175)."""
    es, ids = graphs.levels("line11")
    m_es, m_ids = bo.build_hierarchy(es[0].numpy(), 2, 11, graphs.np("line11/pos"))
    assert [i.tolist() for i in m_ids] == [[1, 3, 5, 7, 9], [1, 3]]
    assert m_es[1].tolist() == [[0, 1, 1, 2, 2, 3, 3, 4], [1, 2, 0, 3, 1, 4, 2, 3]]
    assert m_es[2].tolist() == [[0, 1], [1, 0]]
    for mine, ref in zip(m_ids, ids):
        assert np.array_equal(mine, ref.numpy())


def test_cyc6_graph_conversions(graphs):
    """graph_wrapper.py:216-241: clusters [[0,1,2],[3,4,5]], adjacency round trips."""
    e0 = graphs.np("cyc6/e0")
    adj = bo.adjacency_lists(e0, 6)
    assert np.array_equal(np.array(adj), graphs.np("cyc6/adj_list"))
    assert bo.find_clusters(adj) == graphs.np("cyc6/clusters").tolist() == [[0, 1, 2], [3, 4, 5]]
    d = bo.bfs_depth(adj, 0).astype(float)
    d[d < 0] = 1 + 1e10
    assert np.array_equal(d, graphs.np("cyc6/bfs0"))
    assert np.array_equal(graphs.np("cyc6/from_adj_list"), e0)
    assert np.array_equal(graphs.np("cyc6/from_adj_mat"), e0)


@pytest.mark.parametrize("name,depth", [("cyc6bi", 1), ("del64", 3), ("del300", 3), ("surf200", 3)])
def test_hierarchy_matches_reference(graphs, name, depth):
    es, ids = graphs.levels(name)
    n = graphs.np(f"{name}/pos").shape[0]
    m_es, m_ids = bo.build_hierarchy(es[0].numpy(), depth, n, graphs.np(f"{name}/pos"))
    for mine, ref in zip(m_ids, ids):                       # pool masks: bit-exact
        assert mine.dtype == np.int64 and np.array_equal(mine, ref.numpy())
    for mine, ref in zip(m_es, es):                         # coarse edges: equal as sets
        assert np.array_equal(bo.canonical_edges(mine), bo.canonical_edges(ref.numpy()))
        assert len({(a, b) for a, b in mine.T.tolist()}) == mine.shape[1]  # no duplicates
        assert not np.any(mine[0] == mine[1])                              # no self loops


@pytest.mark.parametrize("name,kind", [("del64", "tri"), ("del300", "tri"), ("surf200", "tri"),
                                       ("quad", "quad"), ("tetra", "tetra"), ("line", "line")])
def test_to_flat_edge_order_exact(graphs, name, kind):
    assert np.array_equal(bo.to_flat_edge(graphs.np(f"{name}/cells"), kind), graphs.np(f"{name}/e0"))


def test_to_flat_edge_errors():
    with pytest.raises(ValueError):
        bo.to_flat_edge(np.zeros((1, 3), dtype=np.int64), "hexa")


# ------------------------------------------------------------------ tensor prims (A1,A2,A5-A8)
@pytest.mark.parametrize("name", ["del64", "del300"])
def test_prims(graphs, name):
    z = load_golden("prims")
    es, ids = graphs.levels(name)
    n0 = graphs.np(f"{name}/pos").shape[0]
    g0 = es[0]
    assert torch.equal(ro.scatter_sum(z.t(f"{name}/scatter_src"), g0[1], -2, n0), z.t(f"{name}/scatter_out"))
    assert torch.equal(ro.degree(g0[0]), z.t(f"{name}/degree"))
    w = torch.ones(n0, 1)
    for l in range(len(ids)):
        ew, aw = ro.cal_ew(w, es[l])
        assert torch.equal(ew, z.t(f"{name}/ew{l}")) and torch.equal(aw, z.t(f"{name}/aggr_w{l}"))
        w = aw[ids[l]]
    ew = z.t(f"{name}/ew0")
    x3, x2 = z.t(f"{name}/x3"), z.t(f"{name}/x2")
    assert torch.equal(ro.edge_conv(x3, g0, ew), z.t(f"{name}/conv_down3"))
    assert torch.equal(ro.edge_conv(x3, g0, ew, False), z.t(f"{name}/conv_up3"))
    assert torch.equal(ro.edge_conv(x2, g0, ew), z.t(f"{name}/conv_down2"))
    assert torch.equal(ro.edge_conv(x2, g0, ew, False), z.t(f"{name}/conv_up2"))
    un = ro.unpool(z.t(f"{name}/coarse3"), n0, ids[0])
    assert torch.equal(un, z.t(f"{name}/unpool3"))
    assert torch.equal(ro.edge_conv(un, g0, ew, False), z.t(f"{name}/prolong3"))
    a, b = z.np(f"{name}/adjoint")
    assert abs(a - b) <= 1e-4 * abs(a)                     # <R h, c> == <h, P c>
    s = ro.scatter_sum(ew, g0[1], -1, n0)                   # edge weights sum to ~1 per target
    assert float((s - 1).abs().max()) < 1e-5


def test_degree_docstring_example():
    assert ro.degree(torch.tensor([0, 1, 0, 2, 0]), dtype=torch.long).tolist() == [3, 1, 1]


def test_rank_errors():
    g = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(NotImplementedError):
        ro.GMP(8, 1, 2)(torch.zeros(1, 1, 2, 8), g, torch.zeros(2, 2))
    with pytest.raises(NotImplementedError):
        ro.edge_conv(torch.zeros(1, 1, 2, 8), g, torch.ones(2))


# ------------------------------------------------------------------ GMP (A3, A4)
@pytest.mark.parametrize("tag,D,p", [("d32p2", 32, 2), ("d128p2", 128, 2), ("d32p3", 32, 3)])
def test_gmp(graphs, tag, D, p):
    z = load_golden(f"gmp_{tag}")
    es, _ = graphs.levels(str(z.np("graph")))
    g = es[0]
    gmp = ro.GMP(D, 3, p)
    gmp.load_state_dict(z.state_dict())
    for xk, pk, ck, yk, dk, gk in (("x3", "pos3", "cot3", "y33", "dx33", "g33/"),
                                   ("x3", "pos2", "cot3", "y32", "dx32", None),
                                   ("x2", "pos2", None, "y22", "dx22", "g22/")):
        gmp.zero_grad()
        x = z.t(xk).clone().requires_grad_(True)
        cot = z.t("cot3") if ck else z.t("cot3")[0]
        y = gmp(x, g, z.t(pk))
        (y * cot).sum().backward()
        close(y, z.t(yk))
        close(x.grad, z.t(dk), rtol=1e-4, atol=1e-5)
        if gk:
            for k, prm in gmp.named_parameters():
                assert rel_err(prm.grad, z.t(gk + k)) < 1e-5, k


# ------------------------------------------------------------------ BSGMP (A8, A9)
@pytest.mark.parametrize("tag,D,p", [("line11", 32, 3), ("del300", 32, 2), ("del64_d128", 128, 2)])
def test_bsgmp(graphs, tag, D, p):
    z = load_golden(f"bsgmp_{tag}")
    L = int(z.np("depth"))
    es, ids = graphs.levels(str(z.np("graph")))
    net = ro.BSGMP(L, D, 3, p)
    missing = net.load_state_dict(z.state_dict(), strict=True)
    h = z.t("h").clone().requires_grad_(True)
    y = net(h, ids[:L], es[: L + 1], z.t("pos"))
    (y * z.t("cot")).sum().backward()
    close(y, z.t("y"))
    assert rel_err(h.grad, z.t("dh")) < 1e-5
    for k, prm in net.named_parameters():
        assert rel_err(prm.grad, z.t("g/" + k)) < 1e-5, k


def test_seeded_init_parity(graphs):
    """Same construction order as the reference => same default init under the same seed."""
    z = load_golden("bsgmp_line11")
    torch.manual_seed(17)
    net = ro.BSGMP(2, 32, 3, 3)
    for k, v in net.state_dict().items():
        assert torch.equal(v, z.t("sd/" + k)), k


# ------------------------------------------------------------------ Simulator step + rollout (A10-A12, A16)
def test_simulator_step_and_rollout(graphs):
    z = load_golden("sim")
    es, ids = graphs.levels("del300")
    B = z.np("node_in").shape[0]
    sim = ro.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2))
    sd = z.state_dict()
    assert set(sd) == set(sim.state_dict())                 # identical checkpoint layout
    # replay warm-up from scratch and compare the fp64 statistics
    fresh = ro.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2))
    m_gs = [e.unsqueeze(0).repeat(B, 1, 1) for e in es]
    m_ids = [i.unsqueeze(0).repeat(B, 1) for i in ids]
    for k in range(3):
        out = fresh((z.t(f"warm_in{k}"), z.t(f"warm_tar{k}"), None, m_gs, m_ids), True, True)
        assert float(out.abs().sum()) == 0.0
    for k, v in fresh.state_dict().items():
        if "Normalizer" in k:
            assert v.dtype == torch.float64
            torch.testing.assert_close(v, sd[k], rtol=1e-12, atol=0)
    sim.load_state_dict(sd)
    pred = sim((z.t("node_in"), z.t("tar"), z.t("mask"), m_gs, m_ids), True, False)
    loss = ro.masked_rmse(pred, z.t("tar"), z.t("mask"))
    loss.backward()
    close(pred, z.t("pred"))
    close(loss, z.t("loss"))
    for k, prm in sim.named_parameters():
        if prm.requires_grad:
            assert rel_err(prm.grad, z.t("g/" + k)) < 1e-5, k
    # masked-out nodes return exactly the input state (model.py:162-163)
    dead = z.t("mask")[..., 0] == 0
    assert torch.equal(pred[dead], z.t("node_in")[..., :2][dead])
    sim.zero_grad()
    roll = ro.rollout(sim, z.t("rollout_ic"), z.t("rollout_mask"),
                      [e.unsqueeze(0) for e in es], [i.unsqueeze(0) for i in ids], 5)
    close(roll, z.t("rollout"), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ block-diagonal batch (A15)
def test_block_diagonal_collate(graphs):
    z = load_golden("blockdiag")
    net = ro.BSGMP(2, 32, 3, 2)
    net.load_state_dict(z.state_dict())
    samples = []
    for nm in ("del64", "del300"):
        es, ids = graphs.levels(nm)
        pos = torch.tensor(graphs.np(f"{nm}/pos")[:, :2], dtype=torch.float32)
        samples.append(dict(x=torch.cat([z.t(f"{nm}/h"), pos], 1), m_gs=es[:3], m_ids=ids[:2]))
    x, m_gs, m_ids = ro.collate_block_diagonal(samples)
    for l in range(3):
        assert torch.equal(m_gs[l], z.t(f"cat/e{l}"))
    for l in range(2):
        assert torch.equal(m_ids[l], z.t(f"cat/ids{l}"))
    with torch.no_grad():
        y = net(x[None, :, :32], m_ids, m_gs, x[None, :, 32:])
    close(y, z.t("y_cat"))
    n64 = 64
    close(y[0, :n64], z.t("del64/y"))                       # batched == per-graph
    close(y[0, n64:], z.t("del300/y"))
