"""GPU-box copy of the integer side of the path: the native bi-stride hierarchy builder
(csrc/hierarchy.hip, host C++ inside libbsms_hip.so; reference graph_wrappers/bsms_graph_wrapper.py:8-154) and
`to_flat_edge` (utils/mesh_convertions.py:4-100) against the golden vectors generated from the reference, run where the
driver runs the `-m gpu` set -- so the builder that feeds every full-size test is itself pinned ON the GPU box, and its
output is carried through `bsms_plan_create` into HBM and read back (indices bit-exact).

The cases are the CPU tests of tests/test_abi_and_host.py (same functions, same fixtures), plus the plan round trip."""
import numpy as np
import pytest
import torch

import test_abi_and_host as host
from oracle import bistride_oracle as bo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import bsms_gnn_amd as eng
    return eng


@pytest.mark.parametrize("name,depth", [("line11", 2), ("cyc6bi", 1), ("del64", 3), ("del300", 3), ("surf200", 3)])
def test_hierarchy_builder_on_the_gpu_box(eng, graphs, name, depth):
    host.test_product_hierarchy_builder(eng, graphs, name, depth)


@pytest.mark.parametrize("name,kind", host.F32_CASES)
def test_hierarchy_from_float32_positions_on_the_gpu_box(eng, name, kind):
    host.test_hierarchy_from_float32_positions(eng, name, kind)


@pytest.mark.parametrize("name,kind", [("del64", "tri"), ("surf200", "tri"), ("quad", "quad"), ("tetra", "tetra"), ("line", "line")])
def test_to_flat_edge_on_the_gpu_box(eng, graphs, name, kind):
    host.test_product_to_flat_edge(eng, graphs, name, kind)


def test_directed_multi_component_on_the_gpu_box(eng):
    host.test_native_hierarchy_directed_and_large(eng)


def test_bench_hierarchy_reaches_hbm_bit_exact(eng):
    """The airfoil bench mesh: native builder == oracle builder (m_ids exact, coarse edges as sets), and every level's plan
    (dst-sorted CSR in HBM) read back from the device equals the oracle's stable counting sort of the same edges."""
    from bench import build_mesh, mesh_points, WORKLOADS
    pts, m_es, m_ids = build_mesh("airfoil")
    w = WORKLOADS["airfoil"]
    _, cells = mesh_points("airfoil")
    flat = eng.to_flat_edge(cells, "tri")
    o_es, o_ids = bo.build_hierarchy(flat, w["levels"], w["nodes"], pts)
    for mine, ref in zip(m_ids, o_ids):
        assert mine.dtype == np.int64 and np.array_equal(mine, ref)
    n = w["nodes"]
    for l, e in enumerate(m_es):
        e = np.asarray(e)
        assert np.array_equal(e, o_es[0] if l == 0 else bo.canonical_edges(o_es[l]))
        plan = eng.LevelPlan(torch.tensor(e), n, ids=torch.tensor(m_ids[l]) if l < len(m_ids) else None, device="cuda")
        rowptr, src_sorted, perm, t_rowptr = plan.export()
        order = np.argsort(e[1], kind="stable")                    # a sequential scatter_add_ visits a target's edges in this order
        assert np.array_equal(perm, order.astype(np.int32)) and np.array_equal(src_sorted, e[0][order].astype(np.int32))
        assert np.array_equal(rowptr, np.concatenate([[0], np.cumsum(np.bincount(e[1], minlength=n))]).astype(np.int32))
        assert np.array_equal(t_rowptr, np.concatenate([[0], np.cumsum(np.bincount(e[0], minlength=n))]).astype(np.int32))
        if l < len(m_ids):
            n = len(m_ids[l])


@pytest.mark.gpu
def test_streams_overlap_probe_and_side_lanes():
    """bsms_streams_overlap: a stream never overtakes itself; among the first streams of a fresh pool at least one overlaps the
    current stream (four hardware queues); the engine's two side lanes, created by the first backward, overlap each other -- the
    property the probe exists for (in a process that built variable-mesh plans first they shared a queue)."""
    import torch
    import bsms_gnn_amd as eng
    L = eng._abi.lib()
    cur = torch.cuda.current_stream()
    assert L.bsms_streams_overlap(cur.cuda_stream, cur.cuda_stream) == 0
    pool = [torch.cuda.Stream() for _ in range(6)]
    flags = [L.bsms_streams_overlap(cur.cuda_stream, s.cuda_stream) for s in pool]
    assert all(f in (0, 1) for f in flags) and any(f == 1 for f in flags), flags
    # two streams on which NOTHING distinguishes the queues must at least give a stable answer
    assert L.bsms_streams_overlap(cur.cuda_stream, pool[flags.index(1)].cuda_stream) == 1


# ------------------------------------------------------------------------------------ variable meshes collated on the device (round 6)
def _mesh_levels(graphs, name, depth=2):
    es, ids = graphs.levels(name)
    n = graphs.np(f"{name}/pos").shape[0]
    sizes = [n] + [int(i.numel()) for i in ids[:depth]]
    return es[:depth + 1], ids[:depth], sizes


@pytest.mark.gpu
def test_plan_concat_equals_plan_create_on_the_concatenated_mesh(eng, graphs):
    """bsms_plan_concat: the block-diagonal union of per-mesh plans built ON THE GPU equals, array for
    array (all eighteen: CSR, transpose, pool maps, compact transition lists, gathered edge weights), the plan bsms_plan_create +
    bsms_plan_set_pool + bsms_plan_bind_edge_weights build from the host collate of the same meshes (PyG Batch semantics,
    datasets/base.py:325-349); so do the edge list / kept ids it writes and the plan's scalar facts.  Pooled and un-pooled parts."""
    from bsms_gnn_amd import _abi
    from bsms_gnn_amd.ops import _stream
    from oracle import bsms_oracle as ro
    names = ("del64", "del300", "surf200", "del64")
    L = _abi.lib()
    for lvl, pooled in ((0, True), (1, True), (2, False)):
        parts, ews, coos, idss, n_off = [], [], [], [], 0
        for nm in names:
            es, ids, sizes = _mesh_levels(graphs, nm)
            plan = eng.LevelPlan(es[lvl].cuda(), sizes[lvl], ids=ids[lvl].cuda() if pooled else None)
            if pooled:
                w = torch.rand(es[lvl].shape[1], device="cuda") + 0.1
                _abi.check(L.bsms_plan_bind_edge_weights(plan.handle, w.data_ptr(), _stream()), "bind")
                ews.append(w)
                idss.append(ids[lvl] + n_off)
            parts.append(plan)
            coos.append(es[lvl] + n_off)
            n_off += sizes[lvl]
        ew_cat = torch.cat(ews) if pooled else None
        got, coo, kept = eng.concat_plans(parts, ew_cat)
        torch.cuda.synchronize()
        coo_h = torch.cat(coos, dim=-1)
        want = eng.LevelPlan(coo_h.cuda(), n_off, ids=torch.cat(idss).cuda() if pooled else None)
        if pooled:
            _abi.check(L.bsms_plan_bind_edge_weights(want.handle, ew_cat.data_ptr(), _stream()), "bind")
        torch.cuda.synchronize()
        assert torch.equal(coo.cpu(), coo_h) and (kept is None) == (not pooled)
        if pooled:
            assert torch.equal(kept.cpu(), torch.cat(idss))
        a, b = got.export_ex(), want.export_ex()
        for k in eng.LevelPlan.ARRAYS:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (lvl, k)
        assert (got.N, got.E, got.Nk, got.max_source, got.min_out_degree) == (want.N, want.E, want.Nk, want.max_source, want.min_out_degree)
        for f in ("bsms_plan_num_pooled",):
            assert getattr(L, f)(got.handle) == getattr(L, f)(want.handle)
        assert (L.bsms_plan_bound_edge_weights(got.handle) or 0) == (ew_cat.data_ptr() if pooled else 0)


@pytest.mark.gpu
def test_mesh_bank_batch_equals_host_collate(eng, graphs):
    """MeshBank.collate (per-mesh plans + edge weights resident in HBM, batch assembled by bsms_plan_concat) against the host
    collate `collate_variable_meshes` + upload: same tensors, and a whole fused training step -- loss, prediction, every gradient
    -- BIT FOR BIT, for three batches of different mesh combinations; no plan is built from an edge list after the meshes' first use."""
    from oracle import bsms_oracle as ro
    cfg = ro.make_cfg(2, 32, 3, 2, 2)
    torch.manual_seed(11)
    pool = {}
    for nm in ("del64", "del300", "surf200"):
        es, ids, sizes = _mesh_levels(graphs, nm)
        n = sizes[0]
        pos = torch.tensor(graphs.np(f"{nm}/pos")[:, :2], dtype=torch.float32)
        state, ntype = torch.randn(n, 2), (torch.rand(n, 1) < 0.1).float()
        x, y, mask = torch.cat([state, pos, ntype], -1), state + 0.1 * torch.randn(n, 2), (ntype == 0).float()
        pool[nm] = [eng.LevelData(es[l], sizes[l], face=ids[l] if l < 2 else None, x=x if l == 0 else None, y=y if l == 0 else None,
                                  mask=mask if l == 0 else None) for l in range(3)]
    sim = eng.BSMS_Simulator(cfg).cuda()
    first = [d.to("cuda") for d in eng.collate_variable_meshes([pool["del64"], pool["del300"]])]
    sim(first, False, True)                                             # normaliser statistics
    grads = eng.GradBuckets(list(sim.parameters()))
    step = eng.FusedStep(sim, grads)
    bank = eng.MeshBank(sim.process, "cuda")
    combos = (("del64", "del300"), ("surf200", "del64", "del64"), ("del300", "surf200"))
    for nm in pool:
        bank.entry(pool[nm])                                            # meshes become resident
    built = eng.LevelPlan.constructed
    for combo in combos:
        samples = [pool[nm] for nm in combo]
        host = [d.to("cuda") for d in eng.collate_variable_meshes(samples)]
        grads.flat.zero_()
        loss_h = step(host, False).clone()
        pred_h, flat_h = step.prediction().clone(), grads.flat.clone()
        built_host = eng.LevelPlan.constructed
        devb = bank.collate(samples)
        assert eng.LevelPlan.constructed == built_host + 3              # the three unions -- wrappers of bsms_plan_concat, no CSR build
        for a, b in zip(devb, host):
            assert a.num_nodes == b.num_nodes and torch.equal(a.edge_index, b.edge_index)
            assert (a.face is None) == (b.face is None) and (a.face is None or torch.equal(a.face, b.face))
        assert torch.equal(devb[0].x, host[0].x) and torch.equal(devb[0].y, host[0].y) and torch.equal(devb[0].mask, host[0].mask)
        grads.flat.zero_()
        loss_d = step(devb, False).clone()
        assert eng.LevelPlan.constructed == built_host + 3              # ... and the step found them: nothing else was built
        assert torch.equal(loss_d, loss_h) and torch.equal(step.prediction(), pred_h) and torch.equal(grads.flat, flat_h), combo
    assert built < eng.LevelPlan.constructed
