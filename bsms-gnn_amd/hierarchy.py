"""Host-side input contract of the hot path: mesh cells -> bidirectional edge list, and the bi-stride
multi-level hierarchy (`m_ids`, per-level edge lists) the BSGMP forward consumes.

Mirrors the reference interface
    to_flat_edge(cells, mesh_type)                                   (src/utils/mesh_convertions.py:88-100)
    BistrideMultiLayerGraph(flat_edge, num_layers, num_nodes, pos_mesh).get_multi_layer_graphs()
                                                                     (src/graph_wrappers/bsms_graph_wrapper.py:8-56)
with the pure-Python BFS (`list.pop(0)`, graph_wrapper.py:67-105) and the MKL SpGEMM replaced by
compiled SciPy csgraph / sparse routines.  Off the timed path; runs once per mesh.

`m_ids` is bit-exact w.r.t. the reference; level-0 edges keep the reference's order; coarse-level
edge lists are emitted in canonical row-major sorted order (the reference's order there is whatever
its SpGEMM backend produced; the edge SET is identical and only the fp32 summation order of a few
messages depends on it)."""
import numpy as np
import scipy.sparse as sp
from scipy.sparse import csgraph

_SIDES = {"tri": ((0, 1), (1, 2), (2, 0)), "quad": ((0, 1), (1, 2), (2, 3), (3, 0)),
          "tetra": ((0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (1, 3))}


def to_flat_edge(mesh, mesh_type):
    mesh = np.asarray(mesh)
    if mesh_type == "flat":
        return mesh
    if mesh_type == "line":
        s, r = mesh[0], mesh[1]
    elif mesh_type in _SIDES:
        ends = np.concatenate([mesh[:, side] for side in _SIDES[mesh_type]], axis=0).astype(np.int64)
        uniq = np.unique(np.stack([ends.max(axis=1), ends.min(axis=1)], axis=1), axis=0)  # lexicographic, like torch.unique(dim=0)
        s, r = uniq[:, 0], uniq[:, 1]
    else:
        raise ValueError(f"Unsupported mesh type {mesh_type} in to_flat_edge.")
    return np.stack([np.concatenate([s, r]), np.concatenate([r, s])]).astype(np.int64)


def _adjacency(flat_edge, n):
    e = flat_edge.shape[1]
    return sp.csr_matrix((np.ones(e, dtype=np.int8), (flat_edge[0], flat_edge[1])), shape=(n, n))


def _hops(adj, seed):
    """Hop distance from `seed` following edge direction; -1 where unreachable."""
    d = csgraph.shortest_path(adj, method="D", directed=True, unweighted=True, indices=seed)
    out = np.full(adj.shape[0], -1, dtype=np.int64)
    ok = np.isfinite(d)
    out[ok] = d[ok].astype(np.int64)
    return out


def _clusters(adj):
    """Reference semantics (graph_wrapper.py:107-134): repeatedly take everything REACHABLE from the
    smallest unassigned node; a single leftover node is its own cluster."""
    n = adj.shape[0]
    left = np.ones(n, dtype=bool)
    out = []
    while left.any():
        rest = np.flatnonzero(left)
        if rest.size == 1:
            out.append(rest)
            break
        reach = _hops(adj, int(rest[0])) >= 0
        members = rest[reach[rest]]
        out.append(members)
        left[members] = False
    return out


def bstride_selection(flat_edge, pos_mesh, n):
    """One level (bsms_graph_wrapper.py:59-154) -> (kept ids ascending int64, coarse flat edges int64 [2,E'])."""
    adj = _adjacency(flat_edge, n)
    keep_mask = np.zeros(n, dtype=bool)
    for members in _clusters(adj):
        pts = pos_mesh[members]
        dist = np.linalg.norm(pts - np.mean(pts, axis=0)[None, :], 2, axis=-1)
        seed = int(members[np.argmin(dist)])                       # nearest to centroid, first on ties (:118-124)
        hops = _hops(adj, seed)
        even = (hops >= 0) & (hops % 2 == 0)
        odd = (hops >= 0) & (hops % 2 == 1)
        n_even, n_odd = int(even.sum()), int(odd.sum())
        keep_mask |= even if (n_even <= n_odd or n_odd == 0) else odd   # the SMALLER class (:90-93)
    keep = np.flatnonzero(keep_mask).astype(np.int64)
    # pattern of (A + I)^2 minus the diagonal, restricted to kept nodes (:74-75, :99-102, :129-154)
    a = (_adjacency(flat_edge, n) + sp.identity(n, dtype=np.int8, format="csr")).astype(bool).astype(np.int32)
    a2 = (a @ a).tocsr()
    sub = a2[keep][:, keep].tocoo()
    off = sub.row != sub.col
    rows, cols = sub.row[off].astype(np.int64), sub.col[off].astype(np.int64)
    order = np.lexsort((cols, rows))
    return keep, np.stack([rows[order], cols[order]])


def _native_hierarchy(flat_edge, num_layers, num_nodes, pos_mesh):
    """The C++ builder of libbsms_hip.so (csrc/hierarchy.hip; host code, works without a GPU)."""
    import ctypes as C
    from . import _abi
    L = _abi.lib()
    coo = np.ascontiguousarray(flat_edge, dtype=np.int64)
    # the seed arithmetic runs in the dtype of the positions, like the reference's NumPy expression
    # (bsms_graph_wrapper.py:118-124): float32 mesh_pos from the datapipe stays float32
    f32 = np.asarray(pos_mesh).dtype == np.float32
    pos = np.ascontiguousarray(pos_mesh, dtype=np.float32 if f32 else np.float64)
    h = C.c_void_p()
    create = L.bsms_hierarchy_create_f32 if f32 else L.bsms_hierarchy_create
    _abi.check(create(coo.ctypes.data, coo.shape[1], num_nodes, pos.ctypes.data, pos.shape[1], num_layers, C.byref(h)),
               "bsms_hierarchy_create")
    try:
        es, ids = [], []
        for l in range(num_layers + 1):
            e = np.empty((2, L.bsms_hierarchy_level_edges(h, l)), dtype=np.int64)
            _abi.check(L.bsms_hierarchy_copy_edges(h, l, e.ctypes.data), "bsms_hierarchy_copy_edges")
            es.append(e)
        for l in range(num_layers):
            k = np.empty(L.bsms_hierarchy_level_nodes(h, l + 1), dtype=np.int64)
            _abi.check(L.bsms_hierarchy_copy_ids(h, l, k.ctypes.data), "bsms_hierarchy_copy_ids")
            ids.append(k)
    finally:
        L.bsms_hierarchy_destroy(h)
    return es, ids


class BistrideMultiLayerGraph:
    def __init__(self, flat_edge, num_layers, num_nodes, pos_mesh, backend="native"):
        """backend="native": C++ builder in libbsms_hip.so (default); "scipy": the SciPy-csgraph builder below."""
        self.num_nodes, self.num_layers = num_nodes, num_layers
        self.pos_mesh = np.asarray(pos_mesh)
        if backend == "native":
            self.m_flat_es, self.m_ids = _native_hierarchy(np.asarray(flat_edge), num_layers, num_nodes, self.pos_mesh)
            self.m_flat_es[0] = np.asarray(flat_edge)
            return
        self.m_flat_es = [np.asarray(flat_edge)]
        self.m_ids = []
        g, pos, n = self.m_flat_es[0], self.pos_mesh, num_nodes
        for _ in range(num_layers):
            keep, g = bstride_selection(g, pos, n)
            pos, n = pos[keep], len(keep)
            self.m_flat_es.append(g)
            self.m_ids.append(keep)

    def get_multi_layer_graphs(self):
        """(m_gs, m_flat_es, m_ids) like the reference; `m_gs` here is the same list as `m_flat_es`
        (the reference returns its Graph wrappers there, which no caller on this path uses)."""
        return self.m_flat_es, self.m_flat_es, self.m_ids
