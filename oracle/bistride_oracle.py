"""CPU oracle for the integer side of the path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates (NumPy/SciPy) how the reference turns mesh cells into a bidirectional edge list and
how it builds the bi-stride multi-level hierarchy (`m_ids`, per-level edge lists).  Integer work:
parity with the reference is bit-exact for `m_ids` and for level-0 edge order; coarse-level edge
lists are equal as SETS (their order comes from a third-party SpGEMM: `sparse_dot_mkl` 0.9.4 /
MKL in the reference, SciPy here; values are discarded, only the sparsity pattern is used).

Parity status: PINNED by tests/golden/graphs_*.npz (generated from the reference, see
tests/golden/make_golden.py), including the reference's own known answer for the 11-node line graph
(graph_wrappers/bsms_graph_wrapper.py:157-175) and the 6-node two-cycle graph
(graph_wrappers/graph_wrapper.py:216-241).
"""
from __future__ import annotations

from collections import deque
from typing import List, Tuple

import numpy as np
import scipy.sparse as sp

_CELL_SIDES = {  # utils/mesh_convertions.py:4-75 : which vertex pairs of a cell are mesh edges
    "tri": [(0, 1), (1, 2), (2, 0)],
    "quad": [(0, 1), (1, 2), (2, 3), (3, 0)],
    "tetra": [(0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (1, 3)],
}


def to_flat_edge(cells: np.ndarray, mesh_type: str) -> np.ndarray:
    """utils/mesh_convertions.py:88-100.  Returns int64 [2, 2*E_u].

    Unique undirected edges as (max, min) pairs in lexicographic order; first half of the output
    runs max->min, second half min->max (mesh_convertions.py:15-21)."""
    cells = np.asarray(cells)
    if mesh_type == "flat":
        return cells
    if mesh_type == "line":                       # :78-85 : cells is [2, E] (senders, receivers)
        s, r = cells[0], cells[1]
        return np.stack([np.concatenate([s, r]), np.concatenate([r, s])]).astype(np.int64)
    if mesh_type not in _CELL_SIDES:
        raise ValueError(f"Unsupported mesh type {mesh_type} in to_flat_edge.")
    pairs = np.concatenate([cells[:, list(side)] for side in _CELL_SIDES[mesh_type]], 0)
    packed = np.stack([pairs.max(1), pairs.min(1)], 1).astype(np.int64)
    uniq = np.unique(packed, axis=0)
    hi, lo = uniq[:, 0], uniq[:, 1]
    return np.stack([np.concatenate([hi, lo]), np.concatenate([lo, hi])])


def adjacency_lists(flat_edge: np.ndarray, n: int) -> List[List[int]]:
    """graph_wrapper.py:152-166 : per-source neighbour lists in edge order."""
    adj: List[List[int]] = [[] for _ in range(n)]
    for s, r in zip(flat_edge[0].tolist(), flat_edge[1].tolist()):
        adj[s].append(r)
    return adj


def bfs_depth(adj: List[List[int]], seed: int) -> np.ndarray:
    """graph_wrapper.py:67-105 : hop distance from `seed`, -1 where unreachable (reference: 1+1e10)."""
    depth = np.full(len(adj), -1, dtype=np.int64)
    depth[seed] = 0
    queue = deque([seed])
    while queue:
        u = queue.popleft()
        for v in adj[u]:
            if depth[v] < 0:
                depth[v] = depth[u] + 1
                queue.append(v)
    return depth


def find_clusters(adj: List[List[int]]) -> List[List[int]]:
    """graph_wrapper.py:107-134 : components reachable from the smallest unassigned node, each
    listed in ascending node order; a single leftover node forms its own cluster."""
    remaining = list(range(len(adj)))
    clusters = []
    while remaining:
        if len(remaining) == 1:
            clusters.append([remaining[0]])
            break
        reach = bfs_depth(adj, remaining[0]) >= 0
        clusters.append([v for v in remaining if reach[v]])
        remaining = [v for v in remaining if not reach[v]]
    return clusters


def cluster_seeds(pos: np.ndarray, clusters: List[List[int]]) -> List[int]:
    """bsms_graph_wrapper.py:107-126 : node nearest the cluster centroid (first on ties)."""
    seeds = []
    for members in clusters:
        pts = pos[members]
        far = np.linalg.norm(pts - np.mean(pts, axis=0)[None, :], 2, axis=-1)
        seeds.append(members[int(np.argmin(far))])
    return seeds


def bistride_level(flat_edge: np.ndarray, pos: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
    """One pooling step (bsms_graph_wrapper.py:59-154).

    Returns (kept node ids ascending [relative to this level], coarse flat edges int64 [2,E'])."""
    adj = adjacency_lists(flat_edge, n)
    clusters = find_clusters(adj)
    kept = set()
    for seed in cluster_seeds(pos, clusters):
        depth = bfs_depth(adj, seed)
        even = np.flatnonzero((depth >= 0) & (depth % 2 == 0))
        odd = np.flatnonzero((depth >= 0) & (depth % 2 == 1))
        # keep the SMALLER parity class; ties and an empty odd class keep `even` (:90-93)
        kept.update((even if (len(even) <= len(odd) or len(odd) == 0) else odd).tolist())
    keep = np.array(sorted(kept), dtype=np.int64)

    # (A + I)^2 with the diagonal removed (:74-75, :99-101); only the pattern matters
    # Same SciPy call sequence as the (stubbed) reference so that even the within-row edge ORDER
    # of the coarse lists agrees with the golden vectors; MKL may order columns differently.
    a = sp.coo_matrix((np.ones(flat_edge.shape[1], dtype=np.int64), (flat_edge[0], flat_edge[1])), shape=(n, n))
    a.setdiag(1)
    a = a.tocsr().astype(float)
    a2 = (a @ a).tocsr()
    a2.setdiag(0)
    a2 = sp.coo_matrix(a2)
    live = a2.data.astype(bool)                       # graph_wrapper.py:203-208
    s, r = a2.row[live], a2.col[live]

    # keep edges whose two ends survive; renumber by rank in `keep` (:141-152)
    rank = -np.ones(n, dtype=np.int64)
    rank[keep] = np.arange(len(keep))
    ok = (rank[s] >= 0) & (rank[r] >= 0)
    return keep, np.stack([rank[s][ok], rank[r][ok]]).astype(np.int64)


def build_hierarchy(flat_edge: np.ndarray, num_layers: int, num_nodes: int, pos: np.ndarray):
    """bsms_graph_wrapper.py:30-44 : (`m_flat_es` [L+1 edge lists], `m_ids` [L id arrays])."""
    m_flat_es, m_ids = [np.asarray(flat_edge)], []
    g, p, n = np.asarray(flat_edge), np.asarray(pos), num_nodes
    for _ in range(num_layers):
        keep, g = bistride_level(g, p, n)
        p, n = p[keep], len(keep)
        m_flat_es.append(g)
        m_ids.append(keep)
    return m_flat_es, m_ids


def canonical_edges(flat_edge: np.ndarray) -> np.ndarray:
    """Row-major sorted copy, for set comparison of coarse edge lists."""
    order = np.lexsort((flat_edge[1], flat_edge[0]))
    return flat_edge[:, order]
