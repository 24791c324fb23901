"""CPU: the C-ABI library loads and exports every symbol include/bsms_hip.h declares; host-side logic
(hierarchy builder, size queries, argument validation) works without a GPU.  No compute is launched."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import bistride_oracle as bo


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__
    __graft_entry__.build()          # hipcc cross-compiles gfx950 without a GPU
    import bsms_gnn_amd as eng
    return eng


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bsms_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bsms_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(eng):
    from bsms_gnn_amd import _abi
    names = declared_symbols()
    assert len(names) >= 25
    lib = C.CDLL(_abi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in bsms_hip.h but not exported"
    assert set(names) == set(_abi.SIGNATURES), set(names) ^ set(_abi.SIGNATURES)
    assert _abi.lib().bsms_abi_version() >= 1
    # ... and the converse: the production library exports no C-ABI entry the header does not declare (experiment-only
    # entries such as bsms_debug_* exist in -DBSMS_EXPERIMENTS builds only)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("bsms_") and ln.split()[-2] in ("T", "t", "W")}
    assert exported, "nm found no bsms_* entry points"
    assert exported <= set(names), f"exported but not declared in include/bsms_hip.h: {sorted(exported - set(names))}"


def test_size_queries_and_validation_without_gpu(eng):
    L = eng._abi.lib()
    B, N, E, D, H = 8, 5233, 31354, 128, 3
    saved, work = L.bsms_gmp_saved_bytes(B, N, E, D, H), L.bsms_gmp_work_bytes(B, N, E, D, H)
    assert saved > (H + 1) * B * E * D * 4 and work > (H + 1) * B * E * D * 4     # edge activations / gradients dominate
    assert L.bsms_gmp_saved_bytes(B, N, E, D, 0) == 0 and L.bsms_mlp_saved_bytes(100, 3, D, D, 99) == 0
    assert L.bsms_mlp_saved_bytes(1000, 4, 128, 128, 3) >= 4 * 1000 * 128 * 4
    # null / unsupported arguments come back as error codes with a message, never a crash
    assert L.bsms_gmp_fwd(None, None, None, 1, 128, 2, 0, 3, None, None, None, None, None) == -1
    assert b"plan is null" in L.bsms_last_error()
    assert L.bsms_mlp_fwd(None, 10, 5, 48, 48, 3, 1, None, None, None, None, None) == -3      # D = 48
    assert b"not supported" in L.bsms_last_error()
    assert L.bsms_plan_destroy(None) == 0 and L.bsms_plan_num_nodes(None) == -1


def test_build_lists_cover_the_source_tree():
    """Every .hip of csrc/ is compiled and every header is part of the build digest (a header missing from the digest means a
    stale library after an edit: chain_dev.h was added in round 5)."""
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bsms-gnn_amd")
    spec = importlib.util.spec_from_file_location("_bsms_build", os.path.join(here, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    on_disk = sorted(f for f in os.listdir(os.path.join(here, "csrc")) if f.endswith(".hip"))
    assert sorted(b.SOURCES) == on_disk
    headers = sorted(f for f in os.listdir(os.path.join(here, "csrc")) if f.endswith(".h"))
    assert set(headers) <= {os.path.basename(h) for h in b.HEADERS}


def test_precision_levels_match_the_header(eng):
    """`bsms_precision` (include/bsms_hip.h) and the Python names of the drop-in modules agree."""
    import os, re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "bsms_hip.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"(BSMS_F32|BSMS_BF16_NODES|BSMS_BF16) = (\d+)", hdr))
    from bsms_gnn_amd.ops import PRECISIONS
    assert PRECISIONS == {"f32": enum["BSMS_F32"], "bf16": enum["BSMS_BF16"], "bf16_nodes": enum["BSMS_BF16_NODES"]}


def test_no_cpu_fallback(eng):
    """The product path must fail loudly instead of degrading to PyTorch when used without a GPU."""
    g = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(eng._abi.BsmsError):
        eng.GMP(32, 1, 2)(torch.zeros(2, 32), g, torch.zeros(2, 2))
    with pytest.raises(eng._abi.BsmsError):
        eng.scatter_sum(torch.zeros(2, 4), torch.tensor([0, 0]), dim=-2, dim_size=1)
    with pytest.raises(eng._abi.BsmsError):
        eng.MLP(3, 32, 32, 2)(torch.zeros(5, 3))
    if not torch.cuda.is_available():
        with pytest.raises(eng._abi.BsmsError):
            eng.LevelPlan(g, 2, device="cpu")


def test_state_dict_layout_matches_reference(eng):
    """108 keys at depth 2, identical names/shapes/dtypes to the golden checkpoint."""
    from conftest import load_golden
    from oracle import bsms_oracle as ro
    z = load_golden("sim")
    sim = eng.BSMS_Simulator(ro.make_cfg(2, 32, 3, 3, 2))
    sd, ref = sim.state_dict(), z.state_dict()
    assert set(sd) == set(ref)
    for k in sd:
        assert sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype, k
    sim.load_state_dict(ref)
    assert len(eng.BSMS_Simulator(ro.make_cfg(2, 32, 3, 2, 2)).state_dict()) == 108
    n = sum(p.numel() for p in eng.BSMS_Simulator(ro.make_cfg(3, 128, 3, 5, 2)).parameters() if p.requires_grad)
    assert n == 1917827                                                     # airfoil model, SURVEY.md section 8b


@pytest.mark.parametrize("name,depth", [("line11", 2), ("cyc6bi", 1), ("del64", 3), ("del300", 3), ("surf200", 3)])
def test_product_hierarchy_builder(eng, graphs, name, depth):
    """bsms_gnn_amd.hierarchy (SciPy-accelerated host builder) == golden m_ids bit-exact, coarse edges as sets."""
    es, ids = graphs.levels(name)
    n = graphs.np(f"{name}/pos").shape[0]
    for backend in ("native", "scipy"):                      # C++ builder in the library and the SciPy one
        _, m_es, m_ids = eng.BistrideMultiLayerGraph(es[0].numpy(), depth, n, graphs.np(f"{name}/pos"),
                                                     backend=backend).get_multi_layer_graphs()
        for mine, ref in zip(m_ids, ids):
            assert mine.dtype == np.int64 and np.array_equal(mine, ref.numpy()), backend
        for l, (mine, ref) in enumerate(zip(m_es, es)):
            want = ref.numpy() if l == 0 else bo.canonical_edges(ref.numpy())      # level 0 keeps the caller's order
            assert np.array_equal(np.asarray(mine), want), (backend, l)


def test_native_hierarchy_directed_and_large(eng):
    """Directed graph with several components + the bench meshes: native == SciPy builder == oracle."""
    rng = np.random.default_rng(3)
    n = 400
    src, dst = rng.integers(0, n, 900), rng.integers(0, n, 900)
    keep = src != dst
    g = np.stack([np.concatenate([src[keep], dst[keep]]), np.concatenate([dst[keep], src[keep]])])
    g = np.unique(g, axis=1)
    pos = rng.random((n, 3))
    a = eng.BistrideMultiLayerGraph(g, 3, n, pos, backend="native")
    b = eng.BistrideMultiLayerGraph(g, 3, n, pos, backend="scipy")
    o_es, o_ids = bo.build_hierarchy(g, 3, n, pos)
    for x, y, z in zip(a.m_ids, b.m_ids, o_ids):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    for l in range(1, 4):
        assert np.array_equal(a.m_flat_es[l], b.m_flat_es[l]) and np.array_equal(a.m_flat_es[l], bo.canonical_edges(o_es[l]))


@pytest.mark.parametrize("name,kind", [("del64", "tri"), ("surf200", "tri"), ("quad", "quad"), ("tetra", "tetra"), ("line", "line")])
def test_product_to_flat_edge(eng, graphs, name, kind):
    assert np.array_equal(eng.to_flat_edge(graphs.np(f"{name}/cells"), kind), graphs.np(f"{name}/e0"))
    with pytest.raises(ValueError):
        eng.to_flat_edge(np.zeros((1, 3), dtype=np.int64), "hexa")


def test_bench_workload_sizes(eng):
    """The synthetic airfoil-like workload has exactly the level sizes quoted in SURVEY.md section 8."""
    from bench import build_mesh
    _, m_es, m_ids = build_mesh("airfoil")
    assert [e.shape[1] for e in m_es] == [31354, 25362, 20896, 16076, 10078, 3878]
    assert [len(i) for i in m_ids] == [2609, 1263, 591, 236, 70]
    _, m_es, m_ids = build_mesh("cylinder")
    assert [e.shape[1] for e in m_es] == [11264, 9120, 7574, 5628, 3280] and [len(i) for i in m_ids] == [937, 448, 206, 61]


def test_product_collate_equals_oracle_collate(eng, graphs):
    """A15: product block-diagonal collate == the oracle's restatement == the golden hand-built batch."""
    from conftest import load_golden
    from oracle import bsms_oracle as ro
    z = load_golden("blockdiag")
    samples, osamples = [], []
    for nm in ("del64", "del300"):
        es, ids = graphs.levels(nm)
        n = graphs.np(f"{nm}/pos").shape[0]
        sizes = [n] + [i.numel() for i in ids[:2]]
        x = torch.zeros(n, 3)
        samples.append([eng.LevelData(es[l], sizes[l], face=ids[l] if l < 2 else None, x=x if l == 0 else None) for l in range(3)])
        osamples.append(dict(x=x, m_gs=es[:3], m_ids=ids[:2]))
    batch = eng.collate_variable_meshes(samples)
    _, o_gs, o_ids = ro.collate_block_diagonal(osamples)
    for l in range(3):
        assert torch.equal(batch[l].edge_index, o_gs[l]) and torch.equal(batch[l].edge_index, z.t(f"cat/e{l}"))
    for l in range(2):
        assert torch.equal(batch[l].face, o_ids[l]) and torch.equal(batch[l].face, z.t(f"cat/ids{l}"))
    assert batch[2].face is None and batch[0].x.shape[0] == 364


F32_CASES = [("grid20x20", "quad"), ("grid31x17", "quad"), ("grid40x25", "quad"), ("grid16x8", "quad"), ("del300f", "tri"),
             ("del500f", "tri"), ("surf200f", "tri")]


@pytest.mark.parametrize("name,kind", F32_CASES)
def test_hierarchy_from_float32_positions(eng, name, kind):
    """datasets/base.py:47 hands the builder FLOAT32 mesh positions and the reference picks each cluster's seed with
    NumPy arithmetic in that dtype (bsms_graph_wrapper.py:118-124).  Structured grids are full of near-ties, where
    fp64 arithmetic picks a different seed -> different m_ids.  Golden: tests/golden/graphs_f32.npz (generated from the
    reference with float32 positions); native C++ builder, SciPy builder and the oracle must all be bit-exact."""
    from conftest import load_golden
    z = load_golden("graphs_f32")
    es, ids = z.levels(name)
    pos = z.np(f"{name}/pos")
    assert pos.dtype == np.float32
    fe = eng.to_flat_edge(z.np(f"{name}/cells"), kind)
    assert np.array_equal(fe, es[0].numpy())
    for backend in ("native", "scipy"):
        _, m_es, m_ids = eng.BistrideMultiLayerGraph(fe, 3, pos.shape[0], pos, backend=backend).get_multi_layer_graphs()
        for l, (mine, ref) in enumerate(zip(m_ids, ids)):
            assert np.array_equal(mine, ref.numpy()), (backend, l)
        for l in range(1, 4):
            assert np.array_equal(np.asarray(m_es[l]), bo.canonical_edges(es[l].numpy())), (backend, l)
    _, o_ids = bo.build_hierarchy(fe, 3, pos.shape[0], pos)
    for mine, ref in zip(o_ids, ids):
        assert np.array_equal(mine, ref.numpy())


def test_float32_and_float64_positions_really_differ(eng):
    """The dtype matters: on a structured grid the fp64 evaluation of the same float32 coordinates keeps different
    nodes (this is the bug the float32 entry fixes -- the fixture would not catch a builder that upcasts)."""
    from conftest import load_golden
    z = load_golden("graphs_f32")
    differ = 0
    for name, kind in F32_CASES[:4]:
        pos = z.np(f"{name}/pos")
        fe = eng.to_flat_edge(z.np(f"{name}/cells"), kind)
        a = eng.BistrideMultiLayerGraph(fe, 3, pos.shape[0], pos).m_ids
        b = eng.BistrideMultiLayerGraph(fe, 3, pos.shape[0], pos.astype(np.float64)).m_ids
        differ += any(x.shape != y.shape or not np.array_equal(x, y) for x, y in zip(a, b))
    assert differ >= 1


def test_intern_index_confirms_content_and_needs_no_xxhash(eng, monkeypatch):
    """graph.intern_index: equal content -> the same tensor object; a digest collision must not alias two meshes (the
    hit is confirmed against a host copy); works without the optional xxhash package (stdlib blake2b)."""
    from bsms_gnn_amd import graph
    graph._INTERNED.clear()
    a = torch.arange(40, dtype=torch.int64).reshape(2, 20)
    t1 = graph.intern_index(a.clone(), "cpu")
    assert graph.intern_index(a.clone(), "cpu") is t1
    monkeypatch.setattr(graph, "_xxhash", None)
    t2 = graph.intern_index(a.clone(), "cpu")            # other digest -> new entry, then stable
    assert graph.intern_index(a.clone(), "cpu") is t2 and torch.equal(t2, a)
    monkeypatch.setattr(graph, "_content_key", lambda t: ("collide",))
    graph._INTERNED.clear()
    x = graph.intern_index(a.clone(), "cpu")
    y = graph.intern_index((a + 1).clone(), "cpu")       # same key, different content
    assert torch.equal(x, a) and torch.equal(y, a + 1)
    # shared batch axis: one slice kept, stride-0 view handed out
    b = a.unsqueeze(0).repeat(4, 1, 1)
    v = graph.intern_index(b, "cpu", shared_batch_axis=True)
    assert v.shape == b.shape and v.stride(0) == 0 and torch.equal(v, b)
    monkeypatch.setenv("BSMS_PLAN_CACHE", "8")
    for k in range(20):
        graph.intern_index(a + 100 + k, "cpu")
    assert len(graph._INTERNED) <= 8
    graph._INTERNED.clear()


def test_retired_plans_are_destroyed_exactly_once_under_concurrency(eng, monkeypatch):
    """graph._retire / _reap run from LevelPlan.__del__ on any thread while bsms_plan_destroy (ctypes) releases the GIL:
    every retired handle must reach the library exactly once -- a double destroy would push one device block into the
    recycling pool twice and two later plans would share it.  Stand-in library: destroy = sleep + record."""
    import threading
    import time
    from bsms_gnn_amd import graph

    class Ev:                        # an event that completes after a few queries
        def __init__(self, n):
            self.n = n

        def query(self):
            self.n -= 1
            return self.n < 0

    seen, lock = [], threading.Lock()

    class FakeLib:
        def bsms_plan_destroy(self, h):
            time.sleep(0.0005)       # the real call releases the GIL too
            with lock:
                seen.append(h)
            return 0

    monkeypatch.setattr(graph._abi, "lib", lambda: FakeLib())
    graph._GRAVE.clear()
    for h in range(64):              # a backlog of unfinished plans that every reaper will walk over
        graph._GRAVE.append(([Ev(h % 5)], ("old", h)))

    def worker(t):
        for k in range(50):
            graph._retire(("new", t, k), None)       # device None: no event, finished at once
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for _ in range(10):
        graph._reap()
    assert not graph._GRAVE
    assert len(seen) == len(set(seen)) == 64 + 8 * 50
