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
