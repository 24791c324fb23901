// Native host-side bi-stride hierarchy builder (no device code): replaces the reference's pure-Python BFS
// (graph_wrappers/graph_wrapper.py:67-134, list.pop(0) frontier), its MKL SpGEMM dependency
// (sparse_dot_mkl.dot_product_mkl, bsms_graph_wrapper.py:99-100) and the NumPy renumbering (:129-154).
// Per level: components by reachability from the smallest unassigned node -> seed = member nearest the
// component centroid (first on ties; arithmetic in the DTYPE OF THE POSITIONS -- fp32 or fp64 -- and in NumPy's
// evaluation order, so the argmin is bit-equal to the reference's np.mean / np.linalg.norm / np.argmin) ->
// BFS hop parity -> keep the SMALLER parity class (ties / empty odd class keep even) -> coarse edges = pattern of
// (A+I)^2 minus the diagonal among kept nodes, renumbered by rank, emitted row-major with sorted columns.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

using namespace bsms;

struct bsms_hierarchy {
  std::vector<std::vector<int64_t>> src, dst;  // per level 0..L
  std::vector<std::vector<int64_t>> ids;       // per level 0..L-1 (relative to that level)
  std::vector<int64_t> nodes;                  // per level 0..L
};

namespace {

struct Csr {
  std::vector<int64_t> ptr, col;
};

Csr build_csr(const std::vector<int64_t>& s, const std::vector<int64_t>& d, int64_t n) {
  Csr c;
  c.ptr.assign(n + 1, 0);
  for (int64_t v : s) c.ptr[v + 1]++;
  for (int64_t i = 0; i < n; ++i) c.ptr[i + 1] += c.ptr[i];
  c.col.resize(s.size());
  std::vector<int64_t> cur(c.ptr.begin(), c.ptr.end() - 1);
  for (size_t e = 0; e < s.size(); ++e) c.col[cur[s[e]]++] = d[e];
  return c;
}

// hop distance from seed along edge direction, -1 = unreachable
void bfs(const Csr& g, int64_t seed, std::vector<int64_t>& depth, std::vector<int64_t>& queue) {
  std::fill(depth.begin(), depth.end(), -1);
  queue.clear();
  queue.push_back(seed);
  depth[seed] = 0;
  for (size_t head = 0; head < queue.size(); ++head) {
    const int64_t u = queue[head];
    for (int64_t q = g.ptr[u]; q < g.ptr[u + 1]; ++q) {
      const int64_t v = g.col[q];
      if (depth[v] < 0) {
        depth[v] = depth[u] + 1;
        queue.push_back(v);
      }
    }
  }
}

template <typename T>   // T = dtype of the mesh positions (the reference computes in whatever dtype pos_mesh has)
void one_level(const std::vector<int64_t>& s, const std::vector<int64_t>& d, int64_t n, const std::vector<T>& pos,
               int p, std::vector<int64_t>& keep, std::vector<int64_t>& cs, std::vector<int64_t>& cd) {
  const Csr g = build_csr(s, d, n);
  std::vector<int64_t> depth(n), queue;
  std::vector<char> assigned(n, 0), kept(n, 0);
  queue.reserve(n);
  int64_t remaining = n, first = 0;
  std::vector<int64_t> members;
  while (remaining > 0) {
    while (first < n && assigned[first]) ++first;
    members.clear();
    if (remaining == 1) {  // a single leftover node is its own cluster (graph_wrapper.py:131-133)
      members.push_back(first);
    } else {
      bfs(g, first, depth, queue);
      for (int64_t v = 0; v < n; ++v)
        if (!assigned[v] && depth[v] >= 0) members.push_back(v);  // ascending node order
    }
    for (int64_t v : members) assigned[v] = 1;
    remaining -= (int64_t)members.size();
    // seed: nearest to the centroid (bsms_graph_wrapper.py:118-124).  np.mean(axis=0) of an [M,p] array adds the rows
    // sequentially in T and divides by T(M); np.linalg.norm(., 2, axis=-1) = sqrt(sum x*x) in T, p < 8 terms added in
    // order; np.argmin takes the first minimum.  Every operation below is one rounding in T (no contraction:
    // host code, and the product is stored before the add).
    T c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t v : members)
      for (int k = 0; k < p; ++k) c[k] = T(c[k] + pos[v * p + k]);
    for (int k = 0; k < p; ++k) c[k] = T(c[k] / T(members.size()));
    int64_t seed = members[0];
    T best = INFINITY;
    for (int64_t v : members) {
      T s2 = 0;
      for (int k = 0; k < p; ++k) {
        const T t = T(pos[v * p + k] - c[k]);
        volatile T sq = T(t * t);   // rounded product, then the add: never an fma
        s2 = T(s2 + sq);
      }
      const T dist = std::sqrt(s2);
      if (dist < best) {
        best = dist;
        seed = v;
      }
    }
    bfs(g, seed, depth, queue);
    int64_t n_even = 0, n_odd = 0;
    for (int64_t v = 0; v < n; ++v)
      if (depth[v] >= 0) ((depth[v] & 1) ? n_odd : n_even)++;
    const int want = (n_even <= n_odd || n_odd == 0) ? 0 : 1;  // the smaller class (bsms_graph_wrapper.py:90-93)
    for (int64_t v = 0; v < n; ++v)
      if (depth[v] >= 0 && (depth[v] & 1) == want) kept[v] = 1;
  }
  keep.clear();
  std::vector<int64_t> rank(n, -1);
  for (int64_t v = 0; v < n; ++v)
    if (kept[v]) {
      rank[v] = (int64_t)keep.size();
      keep.push_back(v);
    }
  // pattern of (A+I)^2 without the diagonal, kept rows and columns only
  cs.clear();
  cd.clear();
  std::vector<int64_t> mark(n, -1), row;
  for (int64_t i : keep) {
    row.clear();
    auto visit = [&](int64_t j) {
      if (mark[j] != i) {
        mark[j] = i;
        if (j != i && rank[j] >= 0) row.push_back(rank[j]);
      }
    };
    for (int64_t q = g.ptr[i]; q < g.ptr[i + 1]; ++q) {
      const int64_t k = g.col[q];
      visit(k);
      for (int64_t r = g.ptr[k]; r < g.ptr[k + 1]; ++r) visit(g.col[r]);
    }
    std::sort(row.begin(), row.end());
    for (int64_t j : row) {
      cs.push_back(rank[i]);
      cd.push_back(j);
    }
  }
}

}  // namespace

namespace {
template <typename T>
int hierarchy_create_t(const int64_t* coo, int64_t E, int64_t N, const T* pos, int64_t pos_dim, int num_layers,
                       bsms_hierarchy_t** out) {
  BSMS_REQUIRE(out != nullptr, BSMS_E_INVALID_ARG, "hierarchy_create: out is null");
  *out = nullptr;
  BSMS_REQUIRE(E >= 0 && N >= 0 && num_layers >= 0, BSMS_E_SHAPE, "hierarchy_create: negative size");
  BSMS_REQUIRE(pos_dim >= 1 && pos_dim <= 8, BSMS_E_UNSUPPORTED, "hierarchy_create: pos_dim=%lld (1..8)", (long long)pos_dim);
  BSMS_REQUIRE((coo || E == 0) && (pos || N == 0), BSMS_E_INVALID_ARG, "hierarchy_create: null argument");
  for (int64_t e = 0; e < E; ++e)
    BSMS_REQUIRE(coo[e] >= 0 && coo[e] < N && coo[E + e] >= 0 && coo[E + e] < N, BSMS_E_INVALID_ARG,
                 "hierarchy_create: edge %lld out of range", (long long)e);
  bsms_hierarchy* h = new bsms_hierarchy();
  h->src.emplace_back(coo, coo + E);
  h->dst.emplace_back(coo + E, coo + 2 * E);
  h->nodes.push_back(N);
  std::vector<T> p(pos, pos + N * pos_dim);
  for (int l = 0; l < num_layers; ++l) {
    std::vector<int64_t> keep, cs, cd;
    one_level<T>(h->src[l], h->dst[l], h->nodes[l], p, (int)pos_dim, keep, cs, cd);
    std::vector<T> np(keep.size() * pos_dim);
    for (size_t k = 0; k < keep.size(); ++k)
      for (int c = 0; c < pos_dim; ++c) np[k * pos_dim + c] = p[keep[k] * pos_dim + c];
    p.swap(np);
    h->nodes.push_back((int64_t)keep.size());
    h->ids.push_back(std::move(keep));
    h->src.push_back(std::move(cs));
    h->dst.push_back(std::move(cd));
  }
  *out = h;
  return BSMS_OK;
}
}  // namespace

extern "C" int bsms_hierarchy_create(const int64_t* coo, int64_t E, int64_t N, const double* pos, int64_t pos_dim,
                                     int num_layers, bsms_hierarchy_t** out) {
  return hierarchy_create_t<double>(coo, E, N, pos, pos_dim, num_layers, out);
}
extern "C" int bsms_hierarchy_create_f32(const int64_t* coo, int64_t E, int64_t N, const float* pos, int64_t pos_dim,
                                         int num_layers, bsms_hierarchy_t** out) {
  return hierarchy_create_t<float>(coo, E, N, pos, pos_dim, num_layers, out);
}

extern "C" int bsms_hierarchy_destroy(bsms_hierarchy_t* h) {
  delete h;
  return BSMS_OK;
}
extern "C" int64_t bsms_hierarchy_level_nodes(const bsms_hierarchy_t* h, int level) {
  return (h && level >= 0 && level < (int)h->nodes.size()) ? h->nodes[level] : -1;
}
extern "C" int64_t bsms_hierarchy_level_edges(const bsms_hierarchy_t* h, int level) {
  return (h && level >= 0 && level < (int)h->src.size()) ? (int64_t)h->src[level].size() : -1;
}
extern "C" int bsms_hierarchy_copy_edges(const bsms_hierarchy_t* h, int level, int64_t* out) {
  BSMS_REQUIRE(h && level >= 0 && level < (int)h->src.size() && (out || h->src[level].empty()), BSMS_E_INVALID_ARG,
               "hierarchy_copy_edges: bad argument");
  const size_t e = h->src[level].size();
  std::copy(h->src[level].begin(), h->src[level].end(), out);
  std::copy(h->dst[level].begin(), h->dst[level].end(), out + e);
  return BSMS_OK;
}
extern "C" int bsms_hierarchy_copy_ids(const bsms_hierarchy_t* h, int level, int64_t* out) {
  BSMS_REQUIRE(h && level >= 0 && level < (int)h->ids.size() && (out || h->ids[level].empty()), BSMS_E_INVALID_ARG,
               "hierarchy_copy_ids: bad argument");
  std::copy(h->ids[level].begin(), h->ids[level].end(), out);
  return BSMS_OK;
}
