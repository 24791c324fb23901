#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by IMPORTING THE REFERENCE (read-only mount at
/root/reference) in the build container.  The reference never travels to the GPU box; only the
small .npz files produced here do.  Re-run:  python tests/golden/make_golden.py

Two stub modules are injected because the image lacks them:
  * wandb            (imported by utils/basic.py:14, unused on this path)
  * sparse_dot_mkl   (bsms_graph_wrapper.py:2,100) -> SciPy SpGEMM; same sparsity pattern.
Everything else is the reference's own code, unmodified.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch

REF = os.environ.get("BSMS_REFERENCE", "/root/reference")
sys.modules["wandb"] = types.ModuleType("wandb")
_mkl = types.ModuleType("sparse_dot_mkl")
_mkl.dot_product_mkl = lambda a, b: (a @ b).tocsr()
sys.modules["sparse_dot_mkl"] = _mkl
sys.path.insert(0, os.path.join(REF, "src"))

from graph_wrappers import BistrideMultiLayerGraph, Graph  # noqa: E402
from graph_wrappers.graph_wrapper import GraphType  # noqa: E402
from models import BSMS_Simulator  # noqa: E402
from ops import BSGMP, GMP, MLP, Unpool, WeightedEdgeConv  # noqa: E402
from utils import Normalizer, degree, scatter_sum, to_flat_edge  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)  # deterministic reduction order


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrays)} arrays)")


def t2n(t):
    return t.detach().cpu().numpy()


def delaunay(n, seed, dim3=False):
    from scipy.spatial import Delaunay

    pts = np.random.default_rng(seed).random((n, 2))
    cells = Delaunay(pts).simplices.astype(np.int64)
    if dim3:
        z = np.sin(3 * pts[:, 0]) * np.cos(3 * pts[:, 1])
        pts = np.concatenate([pts, z[:, None]], 1)
    return pts, cells


def hierarchy(flat_edge, depth, n, pos):
    m = BistrideMultiLayerGraph(flat_edge, depth, n, pos)
    _, m_flat_es, m_ids = m.get_multi_layer_graphs()
    return [np.asarray(e, dtype=np.int64) for e in m_flat_es], [np.asarray(i, dtype=np.int64) for i in m_ids]


def pack_levels(prefix, m_es, m_ids):
    d = {f"{prefix}/n_levels": np.int64(len(m_es))}
    for l, e in enumerate(m_es):
        d[f"{prefix}/e{l}"] = e
    for l, i in enumerate(m_ids):
        d[f"{prefix}/ids{l}"] = i
    return d


class KinkMargin:
    """Smallest |pre-activation| seen by any ReLU of `model` during the forward passes run inside the
    `with` block.  The gradient of a ReLU network is discontinuous where a pre-activation crosses 0, and two
    correct fp32 implementations (different summation order) can land on different sides when |z| is within
    round-off (~1e-6 here).  Fixtures are therefore drawn (by advancing the seed) until every pre-activation
    is at least MARGIN away from 0, so gradient parity is well defined."""

    def __init__(self, model):
        self.model, self.min, self._h = model, float("inf"), []

    def __enter__(self):
        def hook(_m, inp, _out):
            self.min = min(self.min, float(inp[0].detach().abs().min()))
        self._h = [m.register_forward_hook(hook) for m in self.model.modules() if isinstance(m, torch.nn.ReLU)]
        return self

    def __exit__(self, *a):
        for h in self._h:
            h.remove()


MARGIN = 1.5e-6


def sd_arrays(prefix, module):
    return {f"{prefix}/{k}": t2n(v) for k, v in module.state_dict().items()}


def grad_arrays(prefix, module):
    return {f"{prefix}/{k}": t2n(p.grad) for k, p in module.named_parameters() if p.grad is not None}


# ------------------------------------------------------------------------------------------------
def make_graphs():
    out = {}
    # G1: the reference's own 11-node line example (bsms_graph_wrapper.py:157-175, BSMS.py:109-119)
    fe = np.array([[0, 1, 2, 3, 4, 5, 6, 7, 8, 9], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]])
    fe = np.concatenate((fe, fe[::-1]), axis=1)
    pos = np.stack([np.arange(11.0), np.zeros(11), np.zeros(11)], 1)
    es, ids = hierarchy(fe, 2, 11, pos)
    out.update(pack_levels("line11", es, ids))
    out["line11/pos"] = pos

    # G2: 6-node two 3-cycles digraph (graph_wrapper.py:216-241)
    fe2 = np.array([[0, 1, 2, 3, 4, 5], [1, 2, 0, 4, 5, 3]])
    g = Graph(fe2, GraphType.FLAT_EDGE, 6)
    out["cyc6/e0"] = fe2
    out["cyc6/clusters"] = np.array(g.clusters, dtype=np.int64)
    out["cyc6/adj_list"] = np.array(g.get_adj_list(), dtype=np.int64)
    out["cyc6/from_adj_list"] = Graph(g.get_adj_list(), GraphType.ADJ_LIST, 6).get_flat_edge()
    out["cyc6/from_adj_mat"] = Graph(g.get_sparse_adj_mat(), GraphType.ADJ_MAT, 6).get_flat_edge()
    out["cyc6/bfs0"] = g.bfs_dist(0)
    pos6 = np.random.default_rng(6).random((6, 2))
    # bidirectional version so that degree() is defined on every node
    fe2b = np.concatenate((fe2, fe2[::-1]), axis=1)
    es, ids = hierarchy(fe2b, 1, 6, pos6)
    out.update(pack_levels("cyc6bi", es, ids))
    out["cyc6bi/pos"] = pos6

    # G3: seeded Delaunay meshes, 2-D
    for n, depth in ((64, 3), (300, 3)):
        pts, cells = delaunay(n, n)
        fe = to_flat_edge(cells, "tri")
        es, ids = hierarchy(fe, depth, n, pts)
        out.update(pack_levels(f"del{n}", es, ids))
        out[f"del{n}/pos"] = pts
        out[f"del{n}/cells"] = cells
    # G3b: 3-D surface
    pts, cells = delaunay(200, 7, dim3=True)
    fe = to_flat_edge(cells, "tri")
    es, ids = hierarchy(fe, 3, 200, pts)
    out.update(pack_levels("surf200", es, ids))
    out["surf200/pos"] = pts
    out["surf200/cells"] = cells
    # quad / tetra / line conversions
    quads = np.array([[0, 1, 4, 3], [1, 2, 5, 4], [3, 4, 7, 6], [4, 5, 8, 7]], dtype=np.int64)
    out["quad/cells"] = quads
    out["quad/e0"] = to_flat_edge(quads, "quad")
    from scipy.spatial import Delaunay

    p3 = np.random.default_rng(3).random((20, 3))
    tets = Delaunay(p3).simplices.astype(np.int64)
    out["tetra/cells"] = tets
    out["tetra/e0"] = to_flat_edge(tets, "tetra")
    lines = np.array([[0, 1, 2, 3], [1, 2, 3, 4]], dtype=np.int64)
    out["line/cells"] = lines
    out["line/e0"] = to_flat_edge(torch.tensor(lines).numpy(), "line")
    save("graphs", **out)
    return out


def levels_from(graphs, name):
    n = int(graphs[f"{name}/n_levels"])
    es = [torch.tensor(graphs[f"{name}/e{l}"]) for l in range(n)]
    ids = [torch.tensor(graphs[f"{name}/ids{l}"]) for l in range(n - 1)]
    return es, ids


def make_prims(graphs):
    out = {}
    conv = WeightedEdgeConv()
    for name, B, D in (("del64", 2, 8), ("del300", 3, 32)):
        es, ids = levels_from(graphs, name)
        torch.manual_seed(11)
        n0 = graphs[f"{name}/pos"].shape[0]
        g0 = es[0]
        src = torch.randn(B, g0.shape[1], D)
        out[f"{name}/scatter_src"] = t2n(src)
        out[f"{name}/scatter_out"] = t2n(scatter_sum(src, g0[1], dim=-2, dim_size=n0))
        out[f"{name}/degree"] = t2n(degree(g0[0], dtype=torch.float))
        # cal_ew chain across all levels exactly as BSGMP does (BSMS.py:64-89)
        w = torch.ones(n0, 1)
        x3 = torch.randn(B, n0, D)
        x2 = torch.randn(n0, D)
        out[f"{name}/x3"] = t2n(x3)
        out[f"{name}/x2"] = t2n(x2)
        n_l = n0
        for l in range(len(ids)):
            ew, w_full = conv.cal_ew(w, es[l])
            out[f"{name}/ew{l}"] = t2n(ew)
            out[f"{name}/aggr_w{l}"] = t2n(w_full)
            if l == 0:
                down3 = conv(x3, es[0], ew)
                up3 = conv(x3, es[0], ew, aggragating=False)
                out[f"{name}/conv_down3"] = t2n(down3)
                out[f"{name}/conv_up3"] = t2n(up3)
                out[f"{name}/conv_down2"] = t2n(conv(x2, es[0], ew))
                out[f"{name}/conv_up2"] = t2n(conv(x2, es[0], ew, aggragating=False))
                coarse = torch.randn(B, ids[0].numel(), D)
                out[f"{name}/coarse3"] = t2n(coarse)
                un = Unpool()(coarse, n0, ids[0])
                out[f"{name}/unpool3"] = t2n(un)
                out[f"{name}/prolong3"] = t2n(conv(un, es[0], ew, aggragating=False))
                out[f"{name}/restrict3"] = t2n(down3[:, ids[0]])
                # adjointness  <R h, c> == <h, P c>
                out[f"{name}/adjoint"] = np.array(
                    [float((down3[:, ids[0]] * coarse).sum()), float((x3 * conv(un, es[0], ew, aggragating=False)).sum())]
                )
            w = w_full[ids[l]]
    save("prims", **out)


def pick_seed(build, first, tries=300):
    """First seed whose fixture has kink margin > MARGIN, else the best of `tries` (see KinkMargin)."""
    best_seed, best_margin = None, -1.0
    for seed in range(first, first + tries):
        out, margin = build(seed, False)
        if margin > MARGIN:
            break
        if margin > best_margin:
            best_seed, best_margin = seed, margin
    else:
        seed = best_seed
    out, margin = build(seed, True)
    out["seed"] = np.int64(seed)
    out["kink_margin"] = np.float64(margin)
    return out, seed, margin


def make_gmp(graphs):
    for tag, name, D, p, B in (("d32p2", "del64", 32, 2, 2), ("d128p2", "del300", 128, 2, 1), ("d32p3", "surf200", 32, 3, 2)):
        es, ids = levels_from(graphs, name)
        g = es[0]
        n = graphs[f"{name}/pos"].shape[0]

        def build(seed, full):
            out = {}
            torch.manual_seed(seed)
            gmp = GMP(D, 3, p)
            pos2 = torch.tensor(graphs[f"{name}/pos"][:, :p], dtype=torch.float32)
            x3 = torch.randn(B, n, D, requires_grad=True)
            pos3 = pos2.unsqueeze(0).repeat(B, 1, 1) + 0.01 * torch.randn(B, n, p)
            cot = torch.randn(B, n, D)
            x3b = x3.detach().clone().requires_grad_(True)
            x2 = torch.randn(n, D, requires_grad=True)
            with KinkMargin(gmp) as km:
                y33 = gmp(x3, g, pos3)   # 3-D x, 3-D pos
                y32 = gmp(x3b, g, pos2)  # 3-D x, 2-D pos (the `repeat` branch, ops/basic.py:87-88)
                y22 = gmp(x2, g, pos2)   # 2-D x, 2-D pos
            if not full:
                return out, km.min
            out.update(sd_arrays("sd", gmp))
            (y33 * cot).sum().backward()
            out.update(x3=t2n(x3), pos3=t2n(pos3), cot3=t2n(cot), y33=t2n(y33), dx33=t2n(x3.grad))
            out.update(grad_arrays("g33", gmp))
            gmp.zero_grad()
            (y32 * cot).sum().backward()
            out.update(pos2=t2n(pos2), y32=t2n(y32), dx32=t2n(x3b.grad))
            gmp.zero_grad()
            (y22 * cot[0]).sum().backward()
            out.update(x2=t2n(x2), y22=t2n(y22), dx22=t2n(x2.grad))
            out.update(grad_arrays("g22", gmp))
            out["graph"] = np.array(name)
            return out, km.min

        out, seed, margin = pick_seed(build, 5)
        print(f"  gmp_{tag}: seed {seed}, kink margin {margin:.2e}")
        save(f"gmp_{tag}", **out)


def make_bsgmp(graphs):
    for tag, name, L, D, p, B in (("line11", "line11", 2, 32, 3, 0), ("del300", "del300", 3, 32, 2, 1), ("del64_d128", "del64", 2, 128, 2, 2)):
        es, ids = levels_from(graphs, name)
        n = graphs[f"{name}/pos"].shape[0]

        def build(seed, full):
            out = {}
            torch.manual_seed(seed)
            net = BSGMP(L, D, 3, p)
            pos = torch.tensor(graphs[f"{name}/pos"][:, :p], dtype=torch.float32)
            if B == 0:
                h = torch.randn(n, D, requires_grad=True)
                pos_in = pos
            else:
                h = torch.randn(B, n, D, requires_grad=True)
                pos_in = pos.unsqueeze(0).repeat(B, 1, 1)
            cot = torch.randn_like(h)
            with KinkMargin(net) as km:
                y = net(h, ids[:L], es[: L + 1], pos_in)
            if not full:
                return out, km.min
            out.update(sd_arrays("sd", net))
            (y * cot).sum().backward()
            out.update(h=t2n(h), pos=t2n(pos_in), cot=t2n(cot), y=t2n(y), dh=t2n(h.grad))
            out.update(grad_arrays("g", net))
            out["graph"] = np.array(name)
            out["depth"] = np.int64(L)
            return out, km.min

        out, seed, margin = pick_seed(build, 17)
        print(f"  bsgmp_{tag}: seed {seed}, kink margin {margin:.2e}")
        save(f"bsgmp_{tag}", **out)


def make_sim(graphs):
    import contextlib
    import io
    from types import SimpleNamespace

    from utils.rollout_utils import rollout_one_traj

    name, L, D, C, p, B = "del300", 3, 32, 2, 2, 2
    es, ids = levels_from(graphs, name)
    n = graphs[f"{name}/pos"].shape[0]
    cfg = SimpleNamespace(out_dim=C, latent_dim=D, hidden_layer=3, unet_depth=L, pos_dim=p)
    pos = torch.tensor(graphs[f"{name}/pos"], dtype=torch.float32)
    m_gs = [e.unsqueeze(0).repeat(B, 1, 1) for e in es]
    m_ids = [i.unsqueeze(0).repeat(B, 1) for i in ids]

    def build(seed, full):
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            sim = BSMS_Simulator(cfg).to("cpu")
        out, warm = {}, []
        for k in range(3):  # three warm-up accumulations (model.py:108-125)
            state = torch.randn(B, n, C) * (1.0 + k)
            ntype = (torch.rand(B, n, 1) < 0.1).float()
            node_in = torch.cat([state, pos.unsqueeze(0).repeat(B, 1, 1), ntype], -1)
            tar = state + 0.1 * torch.randn(B, n, C)
            mask = (ntype == 0).float()
            z = sim((node_in, tar, mask, m_gs, m_ids), True, True)
            assert float(z.abs().sum()) == 0.0
            warm.append((node_in, tar))
            out[f"warm_in{k}"] = t2n(node_in)
            out[f"warm_tar{k}"] = t2n(tar)
        out.update(sd_arrays("sd", sim))  # includes fp64 normaliser stats after warm-up
        node_in, tar = warm[-1]
        mask = (node_in[..., -1:] == 0).float()
        with KinkMargin(sim) as km:
            pred = sim((node_in, tar, mask, m_gs, m_ids), True, False)
        if not full:
            return {}, km.min
        se = (pred - tar) ** 2
        loss = torch.sqrt((se * mask).sum() / mask.sum() / se.shape[-1])  # trainer/trainer.py:96-97
        loss.backward()
        out.update(node_in=t2n(node_in), tar=t2n(tar), mask=t2n(mask), pred=t2n(pred), loss=t2n(loss))
        out.update(grad_arrays("g", sim))
        # A16: 5-step rollout (utils/rollout_utils.py:14-64), B = 1
        ic = node_in[:1].clone()
        results = torch.zeros(5, n, C)
        rollout_one_traj(SimpleNamespace(model=sim), ic, results, mask[:1], [e.unsqueeze(0) for e in es],
                         [i.unsqueeze(0) for i in ids], cfg)
        out.update(rollout_ic=t2n(ic), rollout_mask=t2n(mask[:1]), rollout=t2n(results))
        out["graph"] = np.array(name)
        return out, km.min

    out, seed, margin = pick_seed(build, 23)
    print(f"  sim: seed {seed}, kink margin {margin:.2e}")
    save("sim", **out)


def make_blockdiag(graphs):
    """G8: two different meshes offset-concatenated by hand the way PyG `Batch` would."""
    L, D, p = 2, 32, 2
    names = ("del64", "del300")
    torch.manual_seed(31)
    net = BSGMP(L, D, 3, p)
    out = sd_arrays("sd", net)
    per = []
    for nm in names:
        es, ids = levels_from(graphs, nm)
        n = graphs[f"{nm}/pos"].shape[0]
        h = torch.randn(n, D)
        pos = torch.tensor(graphs[f"{nm}/pos"][:, :p], dtype=torch.float32)
        y = net(h, ids[:L], es[: L + 1], pos)
        per.append((h, pos, es, ids, y))
        out[f"{nm}/h"] = t2n(h)
        out[f"{nm}/y"] = t2n(y)
    # concatenate with cumulative per-level node offsets
    gs, idl = [], []
    for l in range(L + 1):
        off, parts, iparts = 0, [], []
        for (h, pos, es, ids, y) in per:
            n_l = h.shape[0] if l == 0 else ids[l - 1].numel()
            parts.append(es[l] + off)
            if l < L:
                iparts.append(ids[l] + off)
            off += n_l
        gs.append(torch.cat(parts, 1))
        if l < L:
            idl.append(torch.cat(iparts))
    hcat = torch.cat([q[0] for q in per]).unsqueeze(0)
    pcat = torch.cat([q[1] for q in per]).unsqueeze(0)
    ycat = net(hcat, idl, gs, pcat)
    out["y_cat"] = t2n(ycat)
    for l, g in enumerate(gs):
        out[f"cat/e{l}"] = t2n(g)
    for l, i in enumerate(idl):
        out[f"cat/ids{l}"] = t2n(i)
    save("blockdiag", **out)


if __name__ == "__main__":
    graphs = make_graphs()
    make_prims(graphs)
    make_gmp(graphs)
    make_bsgmp(graphs)
    make_sim(graphs)
    make_blockdiag(graphs)
    # the survey had to clean up bytecode it created on the (root-writable) reference mount
    import glob
    assert not glob.glob(os.path.join(REF, "src", "**", "__pycache__"), recursive=True)
