// BSGMP: the whole bi-stride U-Net of GMP blocks in ONE library call (ops/BSMS.py:39-104).
//
//   down  i = 0..L-1 :  s_i = GMP_down[i](h_i, pos_i) ;  h_{i+1} = restrict_i(s_i) ;  pos_{i+1} = restrict_i(pos_i)
//   bottom           :  h   = GMP_bottom(h_L, pos_L)
//   up    i = 0..L-1 :  d = L-1-i ;  u_d = prolong_d(h) ;  h = GMP_up[i](u_d, pos_d) + s_d
//
// restrict = WeightedEdgeConv + index by m_ids fused, prolong = Unpool + WeightedEdgeConv(aggragating=False) fused
// (bsms_edge_conv with `pooled`).  The skip additions are fused: forward into the up block's node-chain epilogue,
// backward into the adjoint of the restriction (same additions in the same order as separate elementwise passes).
// The entry only sequences the block / transition kernels of this library on the
// caller's stream -- no new arithmetic -- so that a training step costs the host two calls instead of ~60 autograd
// nodes (the Python mirror of the reference's module tree needed 5.4 ms per step to enqueue what the GPU runs in
// 7.6 ms at airfoil size and was the limit outright at cylinder size).
#include "common.h"

using namespace bsms;

namespace {

constexpr int kMaxLevels = 16;

struct Shape {
  int L;
  int64_t B, D, p, N[kMaxLevels + 1], E[kMaxLevels + 1];
  int H;
  int prec = BSMS_F32;
};

struct Carve {
  char* base;
  size_t off = 0;
  explicit Carve(void* b) : base(reinterpret_cast<char*>(b)) {}
  char* bytes(size_t n) {
    char* r = base ? base + off : nullptr;
    off += align_up(n);
    return r;
  }
  float* floats(size_t n) { return reinterpret_cast<float*>(bytes(n * sizeof(float))); }
};

// what the backward needs: every block's own saved blob, the inputs of the blocks (x of a GMP is needed by its
// backward) and the coarse positions
struct Saved {
  void* gmp[2 * kMaxLevels + 1];   // down 0..L-1, bottom, up 0..L-1
  float* hin[kMaxLevels + 1];      // input of GMP_down[i] for i >= 1 and of the bottom block (hin[0] = caller's h)
  float* pos[kMaxLevels + 1];      // positions of level i >= 1 (pos[0] = caller's)
  float* upin[kMaxLevels];         // input of the up block acting on level d
  size_t bytes;
};
Saved carve_saved(void* base, const Shape& s, bool training) {
  Carve c(base);
  Saved v{};
  for (int i = 0; i < s.L; ++i) v.gmp[i] = training ? c.bytes(gmp_saved_bytes_p(s.B, s.N[i], s.E[i], s.D, s.H, s.prec)) : nullptr;
  v.gmp[s.L] = training ? c.bytes(gmp_saved_bytes_p(s.B, s.N[s.L], s.E[s.L], s.D, s.H, s.prec)) : nullptr;
  for (int i = 0; i < s.L; ++i) {
    const int d = s.L - 1 - i;
    v.gmp[s.L + 1 + i] = training ? c.bytes(gmp_saved_bytes_p(s.B, s.N[d], s.E[d], s.D, s.H, s.prec)) : nullptr;
  }
  for (int i = 1; i <= s.L; ++i) {
    v.hin[i] = c.floats(size_t(s.B) * s.N[i] * s.D);
    v.pos[i] = c.floats(size_t(s.B) * s.N[i] * s.p);
  }
  for (int d = 0; d < s.L; ++d) v.upin[d] = c.floats(size_t(s.B) * s.N[d] * s.D);
  v.bytes = c.off;
  return v;
}

constexpr size_t kPerBlockScratchCap = size_t(8) << 30;   // the size query serves training and inference alike: bounds what a forward-only caller carries unused
// level of the block that runs k-th in the BACKWARD: up blocks on levels 0 .. L-1, the bottom block, down blocks on levels L-1 .. 0
inline int level_of_block_bwd(int k, int L) { return k < L ? k : (k == L ? L : 2 * L - k); }

struct Work {
  void* gmp;                      // scratch of one block call (largest level)
  void* gmp_b;                    // second scratch set: the backward alternates so that block k's weight gradients (side
                                  // lanes) can still be reading set k & 1 while block k+1 runs in the other
  void* gmp_blk[2 * kMaxLevels + 1];  // OR one scratch set per block of the backward (in execution order; null when the sets together
                                  // would exceed kPerBlockScratchCap): no block then waits for the lanes of the block before last
  float* skip[kMaxLevels];        // fwd: outputs of the down blocks; bwd: gradient arriving at the skip connections
  float* a[2];                    // two ping-pong level-0 sized buffers
  void* packs[2 * kMaxLevels + 1];  // inference: weight packs of every block (training keeps them in the saved blobs)
  size_t fwd_bytes;               // the part a forward call touches (an inference call keeps its level inputs right behind it)
  size_t bytes;
};
Work carve_work(void* base, const Shape& s) {
  Carve c(base);
  Work w{};
  size_t g = 0;
  for (int i = 0; i <= s.L; ++i) g = std::max(g, bsms_gmp_work_bytes(s.B, s.N[i], s.E[i], s.D, s.H));
  // ---- what a FORWARD touches (training or inference) comes first: bsms_bsgmp_infer_work_bytes stops behind it, so a forward-only
  // caller does not carry the backward's scratch sets (1.4 GB had become 4.7 GB at the airfoil batch-8 shape)
  w.gmp = c.bytes(g);
  for (int i = 0; i < s.L; ++i) w.skip[i] = c.floats(size_t(s.B) * s.N[i] * s.D);
  for (int k = 0; k < 2; ++k) w.a[k] = c.floats(size_t(s.B) * s.N[0] * s.D);
  for (int k = 0; k <= 2 * s.L; ++k) w.packs[k] = c.bytes(gmp_pack_bytes(s.D, s.H));
  w.fwd_bytes = c.off;
  // ---- backward only
  w.gmp_b = c.bytes(s.L > 0 ? g : 0);
  // Per-block sets for the backward (round 5): 2L + 1 scratch sets sized for their own level -- 4.7 GB instead of 1.4 GB at the
  // airfoil batch-8 shape, out of 288 GB; larger shapes than the cap keep the two alternating sets -- remove the 2L - 1 barrier packets with which block k waited for the side lanes of block
  // k - 2 before reusing its set (each ~5 us of idle caller's stream; profiles/r05_perblock_scratch_ab.txt).  The first two blocks
  // use gmp / gmp_b.
  size_t extra = 0;
  for (int k = 2; k <= 2 * s.L; ++k) {
    const int lv = level_of_block_bwd(k, s.L);
    extra += align_up(bsms_gmp_work_bytes(s.B, s.N[lv], s.E[lv], s.D, s.H));
  }
  const bool per_block = s.L > 0 && extra <= kPerBlockScratchCap;
  for (int k = 0; k <= 2 * s.L; ++k) {
    const int lv = level_of_block_bwd(k, s.L);
    w.gmp_blk[k] = !per_block ? nullptr : (k == 0 ? w.gmp : (k == 1 ? w.gmp_b : c.bytes(bsms_gmp_work_bytes(s.B, s.N[lv], s.E[lv], s.D, s.H))));
  }
  w.bytes = c.off;
  return w;
}
inline int level_of_block(int k, int L) { return k <= L ? k : 2 * L - k; }   // down 0..L-1, bottom L, up i acts on L-1-i

int make_shape(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int H, Shape* s, const char* who) {
  BSMS_REQUIRE(plans != nullptr && L >= 0 && L <= kMaxLevels, BSMS_E_INVALID_ARG, "%s: unet_depth %d (0..%d)", who, L, kMaxLevels);
  s->L = L; s->B = B; s->D = D; s->p = p; s->H = H;
  for (int i = 0; i <= L; ++i) {
    BSMS_REQUIRE(plans[i] != nullptr, BSMS_E_INVALID_ARG, "%s: plan of level %d is null", who, i);
    s->N[i] = plans[i]->N;
    s->E[i] = plans[i]->E;
    if (i > 0)
      BSMS_REQUIRE(plans[i - 1]->Nk == s->N[i], BSMS_E_SHAPE, "%s: level %d keeps %lld nodes but level %d has %lld", who, i - 1,
                   (long long)plans[i - 1]->Nk, i, (long long)s->N[i]);
  }
  return BSMS_OK;
}

// parameters of block k: 4 (H + 1) pointers (mlp_node then mlp_edge), blocks ordered down 0..L-1, bottom, up 0..L-1
inline const float* const* block(const float* const* params, int k, int H) { return params + size_t(k) * 4 * (H + 1); }
inline float* const* block(float* const* grads, int k, int H) { return grads + size_t(k) * 4 * (H + 1); }

}  // namespace

extern "C" size_t bsms_bsgmp_saved_bytes(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int hidden) {
  return bsms_bsgmp_saved_bytes_p(plans, L, B, D, p, hidden, BSMS_F32);
}
extern "C" size_t bsms_bsgmp_saved_bytes_p(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int hidden,
                                           int precision) {
  Shape s;
  if (make_shape(plans, L, B, D, p, hidden, &s, "bsgmp_saved_bytes")) return 0;
  s.prec = precision;
  return carve_saved(nullptr, s, true).bytes;
}
extern "C" size_t bsms_bsgmp_work_bytes(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int hidden) {
  Shape s;
  if (make_shape(plans, L, B, D, p, hidden, &s, "bsgmp_work_bytes")) return 0;
  // (an inference call keeps its level inputs behind the forward part; covered, as the backward's sets are larger)
  const Work w = carve_work(nullptr, s);
  return std::max(w.bytes, w.fwd_bytes + carve_saved(nullptr, s, false).bytes);
}
extern "C" size_t bsms_bsgmp_infer_work_bytes(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int hidden) {
  Shape s;
  if (make_shape(plans, L, B, D, p, hidden, &s, "bsgmp_infer_work_bytes")) return 0;
  return carve_work(nullptr, s).fwd_bytes + carve_saved(nullptr, s, false).bytes;
}

extern "C" int bsms_bsgmp_fwd(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                              int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                              const float* const* params, float* out, void* saved, void* work, bsms_stream_t stream) {
  return bsms_bsgmp_fwd_p(plans, ew, L, h, pos, B, D, p, pos_batch_stride, hidden, params, out, saved, work, 0, BSMS_F32, stream);
}

extern "C" int bsms_bsgmp_fwd_ex(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                                 int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                                 const float* const* params, float* out, void* saved, void* work, int reuse, bsms_stream_t stream) {
  return bsms_bsgmp_fwd_p(plans, ew, L, h, pos, B, D, p, pos_batch_stride, hidden, params, out, saved, work, reuse, BSMS_F32, stream);
}

extern "C" int bsms_bsgmp_fwd_p(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                                int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                                const float* const* params, float* out, void* saved, void* work, int reuse, int precision,
                                bsms_stream_t stream) {
  Shape s;
  int rc = make_shape(plans, L, B, D, p, hidden, &s, "bsgmp_fwd");
  if (rc) return rc;
  BSMS_REQUIRE(precision == BSMS_F32 || precision == BSMS_BF16 || precision == BSMS_BF16_NODES, BSMS_E_UNSUPPORTED, "bsgmp_fwd: precision %d", precision);
  s.prec = precision;
  BSMS_REQUIRE(h && pos && params && out && work && (ew || L == 0), BSMS_E_INVALID_ARG, "bsgmp_fwd: null argument");
  hipStream_t st = as_stream(stream);
  const bool training = saved != nullptr;
  Work w = carve_work(work, s);
  Saved v = training ? carve_saved(saved, s, true) : carve_saved(reinterpret_cast<char*>(work) + w.fwd_bytes, s, false);
  const int64_t posB = pos_batch_stride ? B : 1;   // a 2-D pos is shared by the batch (ops/basic.py:87-88)

  // Two side lanes run ahead of the blocks; both are joined before level 1 starts:
  //  lane 1: positions of all coarse levels -- they depend on pos and ew only (ops/BSMS.py:75,85-88);
  //  lane 0: the weight prepacks of blocks 1..2L (block 0's stays in front of block 0 on the caller's stream): the
  //          weights are fixed for the whole call, so none of these small launches has to sit between two blocks.
  const float* pos_l[kMaxLevels + 1];
  int64_t pstride_l[kMaxLevels + 1];
  pos_l[0] = pos; pstride_l[0] = pos_batch_stride;
  // `reuse` (inference only; the caller passes the SAME work buffer as in its previous call and nothing else wrote to
  // it): bit 0 = the weights have not changed, the packs in `work` are still valid; bit 1 = pos and the mesh have not
  // changed, the coarse positions in `work` are still valid.  An autoregressive rollout sets both after its first step.
  const bool packs_ok = !training && (reuse & 1), pos_ok = !training && (reuse & 2);
  SideLane *lane = nullptr, *lane0 = nullptr;
  auto packs_of = [&](int k) { return training ? nullptr : w.packs[k]; };
  for (int i = 0; i < L; ++i) {
    pos_l[i + 1] = v.pos[i + 1];
    pstride_l[i + 1] = pos_batch_stride ? s.N[i + 1] * p : 0;
  }
  // The lanes FORK here (in front of block 0) but their launches are enqueued BEHIND block 0's: the 15 small lane kernels
  // come first in time on the GPU either way, and enqueued first they kept the caller's stream waiting for the host --
  // block 0 started 80-210 us late in every step of the round-6 traces (profiles/r06_schedule_probes.txt).
  if (L > 0) {
    if (!pos_ok && ((rc = side_lane(&lane, 1, st)) || (rc = side_fork(lane, st)))) return rc;
    if (!packs_ok && ((rc = side_lane(&lane0, 0, st)) || (rc = side_fork(lane0, st)))) return rc;
  }
  auto enqueue_lanes = [&]() -> int {
    int r;
    for (int i = 0; i < L && !pos_ok; ++i)
      if ((r = bsms_edge_conv(plans[i], pos_l[i], posB, p, ew[i], 1, 1, v.pos[i + 1], lane->stream))) return r;
    if (!packs_ok) {
      // two marks on the lane: the packs of the down path + bottom block (slot 0: waited for behind block 0) and those of the up path
      // (slot 1: waited for in front of the first up block).  One join behind block 0 made the caller's stream wait for all 2L
      // prepacks -- 130-160 us per step at D = 256, where a block's prepack takes 55 us (profiles/r05_surface_join_ab.txt)
      for (int k = 1; k <= 2 * L; ++k) {
        const int lv = level_of_block(k, L);
        if ((r = gmp_prepack(B, s.N[lv], s.E[lv], D, p, hidden, block(params, k, hidden), v.gmp[k], w.gmp, packs_of(k), lane0->stream, precision))) return r;
        if ((k == L || k == 2 * L) && (r = side_mark(lane0, k == L ? 0 : 1))) return r;
      }
    }
    return 0;
  };
  const float* hi = h;
  for (int i = 0; i < L; ++i) {
    if ((rc = gmp_fwd_core(plans[i], hi, pos_l[i], B, D, p, pstride_l[i], hidden, block(params, i, hidden), w.skip[i], v.gmp[i], w.gmp,
                           packs_of(i), i == 0 && !packs_ok, nullptr, st, precision))) return rc;
    if (i == 0 && ((rc = enqueue_lanes()) || (lane && (rc = side_join(lane, st))) || (lane0 && (rc = side_wait_mark(lane0, 0, st))))) return rc;
    // restrict the features to the kept nodes (ops/BSMS.py:74, 79-83)
    if ((rc = bsms_edge_conv(plans[i], w.skip[i], B, D, ew[i], 1, 1, v.hin[i + 1], stream))) return rc;
    hi = v.hin[i + 1];
  }
  const float* pi = pos_l[L];
  const int64_t pstride = pstride_l[L];
  float* cur = (L == 0) ? out : w.a[0];
  if ((rc = gmp_fwd_core(plans[L], hi, pi, B, D, p, pstride, hidden, block(params, L, hidden), cur, v.gmp[L], w.gmp, packs_of(L),
                         L == 0 && !packs_ok, nullptr, st, precision))) return rc;
  if (lane0 && (rc = side_wait_mark(lane0, 1, st))) return rc;   // the up blocks' packs
  for (int i = 0; i < L; ++i) {
    const int d = L - 1 - i;
    if ((rc = bsms_edge_conv(plans[d], cur, B, D, ew[d], 0, 1, v.upin[d], stream))) return rc;   // prolong (BSMS.py:98-100)
    // up block + skip connection (BSMS.py:101-102): the node chain's epilogue adds s_d after its own residual
    float* nxt = (d == 0) ? out : w.a[(i + 1) & 1];
    if ((rc = gmp_fwd_core(plans[d], v.upin[d], pos_l[d], B, D, p, pstride_l[d], hidden, block(params, L + 1 + i, hidden), nxt,
                           v.gmp[L + 1 + i], w.gmp, packs_of(L + 1 + i), false, w.skip[d], st, precision))) return rc;
    cur = nxt;
  }
  return BSMS_OK;
}

extern "C" int bsms_bsgmp_bwd(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                              const float* grad_out, int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                              const float* const* params, const void* saved, void* work, float* grad_h, float* const* grads,
                              bsms_stream_t stream) {
  return bsms_bsgmp_bwd_p(plans, ew, L, h, pos, grad_out, B, D, p, pos_batch_stride, hidden, params, saved, work, grad_h, grads,
                          BSMS_F32, stream);
}

extern "C" int bsms_bsgmp_bwd_p(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                                const float* grad_out, int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                                const float* const* params, const void* saved, void* work, float* grad_h, float* const* grads,
                                int precision, bsms_stream_t stream) {
  return bsms_bsgmp_bwd_ex(plans, ew, L, h, pos, grad_out, B, D, p, pos_batch_stride, hidden, params, saved, work, grad_h, grads,
                           precision, 0, stream);
}

extern "C" int bsms_side_lanes_join(bsms_stream_t stream) {
  SideLane *lane0 = nullptr, *lane1 = nullptr;
  int rc;
  hipStream_t st = as_stream(stream);
  if ((rc = side_lane(&lane0, 0, st)) || (rc = side_lane(&lane1, 1, st))) return rc;
  for (int slot = 0; slot < 2; ++slot)   // a slot nobody marked is an event that was never recorded: the wait is a no-op
    if ((!gmp_marks_chained() && (rc = side_wait_mark(lane0, slot, st))) || (rc = side_wait_mark(lane1, slot, st))) return rc;
  return BSMS_OK;
}

extern "C" int bsms_bsgmp_bwd_ex(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                                 const float* grad_out, int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                                 const float* const* params, const void* saved, void* work, float* grad_h, float* const* grads,
                                 int precision, int flags, bsms_stream_t stream) {
  return bsms_bsgmp_bwd_ev(plans, ew, L, h, pos, grad_out, B, D, p, pos_batch_stride, hidden, params, saved, work, grad_h, grads,
                           precision, flags, nullptr, stream);
}

extern "C" int bsms_bsgmp_bwd_ev(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                                 const float* grad_out, int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                                 const float* const* params, const void* saved, void* work, float* grad_h, float* const* grads,
                                 int precision, int flags, void* const* block_done_events, bsms_stream_t stream) {
  Shape s;
  int rc = make_shape(plans, L, B, D, p, hidden, &s, "bsgmp_bwd");
  if (rc) return rc;
  BSMS_REQUIRE(precision == BSMS_F32 || precision == BSMS_BF16 || precision == BSMS_BF16_NODES, BSMS_E_UNSUPPORTED, "bsgmp_bwd: precision %d", precision);
  s.prec = precision;
  BSMS_REQUIRE(h && pos && grad_out && params && saved && work && grad_h && grads && (ew || L == 0), BSMS_E_INVALID_ARG,
               "bsgmp_bwd: null argument");
  hipStream_t st = as_stream(stream);
  Work w = carve_work(work, s);
  Saved v = carve_saved(const_cast<void*>(saved), s, true);
  const float* pos_l[kMaxLevels + 1];
  int64_t pstride_l[kMaxLevels + 1];
  const float* hin_l[kMaxLevels + 1];
  for (int i = 0; i <= L; ++i) {
    pos_l[i] = i ? v.pos[i] : pos;
    pstride_l[i] = i ? (pos_batch_stride ? s.N[i] * p : 0) : pos_batch_stride;
    hin_l[i] = i ? v.hin[i] : h;
  }
  // Blocks run in reverse order.  The side lanes of a block (its weight gradients, csrc/gmp.hip) are NOT joined at the
  // end of the block: block k only marks them (slot k & 1) and the caller's stream goes straight on to block k+1, which
  // works in the other scratch set; block k+2 waits for slot k & 1 before it overwrites that set.  Measured before
  // this change the caller's stream idled 30-160 us per block at the join (the split-K weight gradients of a block
  // take longer than its gradient strand), ~1 ms of a 7.6 ms step.
  SideLane *lane0 = nullptr, *lane1 = nullptr;
  if ((rc = side_lane(&lane0, 0, st)) || (rc = side_lane(&lane1, 1, st))) return rc;
  int nblk = 0;
  bool marked[2] = {false, false};   // slots this call has marked (gmp_bwd_core marks both lanes of its slot)
  auto run_block = [&](int level, const float* x, const float* g_in, int k, float* gx) -> int {
    const int slot = nblk & 1;
    int r;
    void* const own = w.gmp_blk[nblk];   // this block's own scratch set, or null: alternate between the two shared ones
    if (!own && marked[slot] && ((!gmp_marks_chained() && (r = side_wait_mark(lane0, slot, st))) || (r = side_wait_mark(lane1, slot, st)))) return r;
    const int order = nblk;   // position of this block in the backward's execution order
    ++nblk;
    marked[slot] = true;
    if ((r = gmp_bwd_core(plans[level], x, pos_l[level], g_in, B, D, p, pstride_l[level], hidden, block(params, k, hidden), v.gmp[k],
                          own ? own : (slot ? w.gmp_b : w.gmp), gx, block(grads, k, hidden), slot, st, precision))) return r;
    // Gradient-bucket hand-off (bsms_bsgmp_bwd_ev): lane 1's mark of this block covers lane 0's (gmp_marks_chained) and both
    // lanes are in-order streams, so an event recorded on lane 1 HERE completes when every weight gradient of this block,
    // of the blocks before it and of anything queued on the lanes earlier (a deferred bsms_mlp_bwd_ex) has been written.
    if (block_done_events && block_done_events[order]) {
      if (!gmp_marks_chained() && (r = side_wait_mark(lane0, slot, lane1->stream))) return r;
      BSMS_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(block_done_events[order]), lane1->stream));
    }
    return BSMS_OK;
  };
  // up path, last block first.  The gradient reaching level d is both the up block's grad_out and the gradient of
  // the skip connection s_d: it stays in w.skip[d] until the down path picks it up.
  const float* g = grad_out;
  const float* skip_grad[kMaxLevels];
  for (int i = L - 1; i >= 0; --i) {
    const int d = L - 1 - i;
    skip_grad[d] = g;      // level 0: the caller's grad_out; deeper: w.skip[d], written by the level above
    float* gu = w.a[0];    // gradient w.r.t. the up block's input u_d
    if ((rc = run_block(d, v.upin[d], g, L + 1 + i, gu))) return rc;
    // adjoint of prolong_d: a restrict-shaped gather onto level d + 1, kept for that level's skip connection
    float* gnext = (d + 1 < L) ? w.skip[d + 1] : w.a[1];
    if ((rc = bsms_edge_conv(plans[d], gu, B, D, ew[d], 1, 1, gnext, stream))) return rc;
    g = gnext;
  }
  // bottom block
  float* gb = w.a[0];
  if ((rc = run_block(L, hin_l[L], g, L, (L == 0) ? grad_h : gb))) return rc;
  const float* gl = gb;
  for (int i = L - 1; i >= 0; --i) {
    // adjoint of restrict_i, plus the gradient that arrived at the skip connection of level i (fused add)
    float* gs = w.a[1];
    if ((rc = edge_conv_add(plans[i], gl, B, D, ew[i], 0, 1, gs, skip_grad[i], st))) return rc;
    float* gx = (i == 0) ? grad_h : w.a[0];
    if ((rc = run_block(i, hin_l[i], gs, i, gx))) return rc;
    gl = gx;
  }
  // every weight gradient has to be complete when the call returns -- unless the caller has more work for this stream
  // that touches neither `work` nor `grads` and joins the lanes itself afterwards (BSMS_BWD_DEFER_JOIN)
  if (flags & BSMS_BWD_DEFER_JOIN) return BSMS_OK;
  for (int slot = 0; slot < 2; ++slot)
    if (marked[slot] && ((!gmp_marks_chained() && (rc = side_wait_mark(lane0, slot, st))) || (rc = side_wait_mark(lane1, slot, st)))) return rc;
  return BSMS_OK;
}
