// Host orchestration of the two coarse C-ABI entries: a whole MLP (encoder / decoder) and a whole GMP
// block (ops/basic.py:26-98), forward and backward, as sequences of the chain / rowsum / wgrad kernels.
//
// GMP forward (H = hidden layers):
//   prepack            all weight matrices of the block -> MFMA fragment order (one launch; kept for bwd)
//   proj  x2           Ps = x Wi^T + b0,  Pd = x Wj^T         (the first edge Linear split by linearity:
//                      W0 [fiber | x_i | x_j] = Wf fiber + Wi x_i + Wj x_j, so 2 D^2 MACs per NODE
//                      instead of per EDGE; ops/basic.py:90 concatenation order [fiber, x_i, x_j])
//   edge chain         relu(Ps[i] + Pd[j] + Wf fiber) -> H-1 x (Linear, ReLU) -> Linear -> LayerNorm
//   segment sum        aggr[b,n] = sum of messages into n (dst-sorted CSR, no atomics)
//   node chain         [x, aggr] -> MLP -> LayerNorm -> + x
// GMP backward mirrors it: node chain bwd -> edge chain bwd -> segment sums of the edge gradient by
// source and by target -> all weight gradients in one batched split-K launch -> input gradient.
#include <cstdlib>

#include "chain.h"

using namespace bsms;

namespace {

struct Carver {  // identical walk for size queries (base == nullptr) and real buffers
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(reinterpret_cast<char*>(b)) {}
  float* take(size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align_up(nfloat * sizeof(float));
    return p;
  }
  char* take_bytes(size_t n) {
    char* p = base ? base + off : nullptr;
    off += align_up(n);
    return p;
  }
};

bool supported_D(int64_t D) { return D == 32 || D == 64 || D == 128 || D == 256; }

// Experiment switches exist only in a library built with -DBSMS_EXPERIMENTS (BSMS_EXPERIMENTS=1 python
// bsms-gnn_amd/build.py; used by profiles/experiments.py, tile_timeline.py, ab.sh): the production build has no
// environment variable or exported setter that could change kernel behaviour (bit 0 skips the activation stores the
// backward needs).
#ifdef BSMS_EXPERIMENTS
static int env_flags() { const char* e = getenv("BSMS_DEBUG_FLAGS"); return e ? atoi(e) : 0; }
unsigned long long* g_timing = nullptr;
int g_debug_flags = env_flags();  // bit0 skip fwd activation stores, bit1 plain (not nt) stores, bit2 nt final y, bit3 no side lanes, ...
#else
constexpr unsigned long long* g_timing = nullptr;
constexpr int g_debug_flags = 0;
#endif

void add_pack(PackTable& t, const float* W, int ld, int row0, int col0, int N, int K, int kind, float* dst,
              const float* bias = nullptr) {
  PackDesc& d = t.d[t.n++];
  d.W = W; d.bias = bias; d.dst = dst; d.ld = ld; d.row0 = row0; d.col0 = col0; d.N = N; d.K = K; d.kind = kind;
}

// ================================================================================== GMP layout
struct GmpSaved {
  float *e_act[kMaxStages], *e_y, *e_rstd, *aggr, *e_fiber;
  float *e_Ps, *e_Pd;   // fused edge backward of the bf16 precisions (efuse.hip): the two node projections are kept for the backward's recompute
  int* e_aexp[kMaxStages];   // fused fp32 edge backward (efuse32.hip): scale exponents of the rows of e_act[l], which then holds fp16 x 2 pieces
  float *n_act[kMaxStages], *n_yln, *n_rstd;
  // packs (fragment order)
  float *e_wi, *e_wj, *e_wft, *e_w[kMaxStages], *e_wt[kMaxStages], *e_wit, *e_wjt;
  float* e_wr[kMaxStages];   // fused edge backward: PACK_ROWS_BF16 images of the D x D edge Linears (efuse.hip)
  float *n_w0x, *n_w0a, *n_w[kMaxStages], *n_wt[kMaxStages], *n_w0xt, *n_w0at;
  // magnitude bounds (training; chain.h: kBoundSlots floats, zeroed by the block's prepack): [0..7] tensor entering edge
  // forward stage l (e_act[l]); [8..15] node forward stage l ([8]: x and aggr jointly, [8 + l]: n_act[l - 1]);
  // [16..23] gradient entering edge backward stage k (gE[H - k]; [16 + H]: gE[0]); [24..31] node backward (gN[H - k])
  float* bound;
  size_t bytes;
};
// training = true: everything the backward needs.  training = false (inference, `saved` == NULL at the ABI):
// only the messages, the aggregate and the forward packs, carved from the scratch buffer instead.
// `packs_base` (inference only, nullable): carve the weight packs from there instead of behind the messages, so that
// they can be filled ahead of the call and survive the scratch reuse of other blocks.
// `bf`: bf16 precision -- the edge activations and the messages are bf16 (half the floats; the sign bits keep their size);
// `bfn` (BSMS_BF16_NODES): the node MLP's saved activations likewise
// `fused` (use_edge_fused): bf16 precisions -- the edge backward recomputes its activations (efuse.hip): no e_act tensors, the projections
// instead; fp32 -- e_act[l] hold the fp16 x 2 pieces of the rows (chain.hip: k_edge_fwd SAVE == 2) and e_aexp[l] their scale exponents (efuse32.hip)
GmpSaved carve_gmp_saved(void* base, int64_t B, int64_t N, int64_t E, int64_t D, int H, bool training = true,
                         void* packs_base = nullptr, bool bf = false, bool bfn = false, bool fused = false) {
  Carver c(base);
  GmpSaved s{};
  const size_t re = size_t(B) * E, rn = size_t(B) * N, dd = pack_floats(D);
  const size_t edge_act = bf ? pad_rows(re) * (size_t(D) / 2 + mask_words_per_row(D)) : act_floats(re, D);
  if (training && (!fused || !bf))
    for (int l = 0; l < H; ++l) s.e_act[l] = c.take(edge_act);
  if (training && fused && !bf)   // fp32 fused backward: the activations are kept as the pieces of their own stage + one exponent per row
    for (int l = 0; l < H; ++l) s.e_aexp[l] = reinterpret_cast<int*>(c.take(pad_rows(re)));
  if (training && fused && bf) { s.e_Ps = c.take(rn * D); s.e_Pd = c.take(rn * D); }
  s.e_y = c.take(bf ? re * D / 2 : re * D);
  if (training) s.e_rstd = c.take(re);
  if (training) s.e_fiber = c.take(re * 8);   // [B*E, fiber_ld(p)]: sized for the widest pitch (p is not part of the size query)
  s.aggr = c.take(rn * D);
  if (training) {
    for (int l = 0; l < H; ++l) s.n_act[l] = c.take(bfn ? pad_rows(rn) * (size_t(D) / 2 + mask_words_per_row(D)) : act_floats(rn, D));
    s.n_yln = c.take(rn * D);
    s.n_rstd = c.take(rn);
    s.bound = c.take(size_t(kBoundSlots) * kBoundWidth);
  }
  Carver cp(packs_base);
  Carver& k = packs_base ? cp : c;
  s.e_wi = k.take(dd); s.e_wj = k.take(dd); s.e_wft = k.take(size_t(8) * D);
  if (training) { s.e_wit = k.take(dd); s.e_wjt = k.take(dd); }
  // transposed packs: the unfused backward and the fp32 fused one (efuse32.hip streams them); row images: the bf16 fused one (efuse.hip)
  for (int l = 1; l <= H; ++l) { s.e_w[l] = k.take(dd); if (training && (!fused || !bf)) s.e_wt[l] = k.take(dd); if (training && fused && bf) s.e_wr[l] = k.take(size_t(D) * 72); }
  s.n_w0x = k.take(dd); s.n_w0a = k.take(dd);
  if (training) { s.n_w0xt = k.take(dd); s.n_w0at = k.take(dd); }
  for (int l = 1; l <= H; ++l) { s.n_w[l] = k.take(dd); if (training) s.n_wt[l] = k.take(dd); }
  s.bytes = c.off;
  return s;
}

struct GmpWork {
  float *Ps, *Pd;                    // fwd
  float *gN[kMaxStages + 1], *daggr; // bwd
  float *gE[kMaxStages + 1], *dPs, *dPd;
  char *wg, *wg2, *sw;
  size_t sw_bytes;
  float* ef_part;   // fused edge backward: per-workgroup partial weight gradients (efuse.hip)
  size_t bytes;
};
bool edge_fused_possible(int64_t D, int H);   // below: use_edge_fused for any precision
GmpWork carve_gmp_work(void* base, int64_t B, int64_t N, int64_t E, int64_t D, int H) {
  Carver c(base);
  GmpWork w{};
  const size_t re = size_t(B) * E, rn = size_t(B) * N;
  w.Ps = c.take(rn * D); w.Pd = c.take(rn * D);
  w.dPs = w.Ps; w.dPd = w.Pd;  // the backward reuses the two projection buffers for their gradients
  for (int l = 0; l <= H; ++l) w.gN[l] = c.take(pad_rows(rn) * D);   // written tile-wise by the backward chains
  w.daggr = c.take(rn * D);
  for (int l = 0; l <= H; ++l) w.gE[l] = c.take(pad_rows(re) * D);
  w.wg = c.take_bytes(wgrad_work_bytes((int)D, 0));
  w.wg2 = c.take_bytes(wgrad_work_bytes((int)D, 0));   // second split-K area: two wgrad launches run concurrently
  w.sw_bytes = small_wgrad_work_bytes_rows((int)D, B * N);   // one partial block per workgroup of the fused scatter kernel
  w.sw = c.take_bytes(w.sw_bytes);
  // one partial per workgroup of the fused edge backward, where a precision of this build can take it (the size query has no
  // precision argument): min(tiles, CUs) x 198 KB instead of 52 MB for every set
  w.ef_part = edge_fused_possible(D, H) ? c.take(edge_fused_part_floats(int64_t(re))) : nullptr;
  w.bytes = c.off;
  return w;
}
// The fused edge backward (efuse.hip) is THE path of the bf16 precisions at D = 128, hidden = 3; experiment builds can switch
// it off for A/B runs (BSMS_EDGE_FUSED=0).  A pure function of the shape: size queries and calls agree.
#ifdef BSMS_EXPERIMENTS
static int env_fused() { const char* e = getenv("BSMS_EDGE_FUSED"); return e ? atoi(e) : 1; }
static const int g_edge_fused = env_fused();
#else
constexpr int g_edge_fused = 1;
#endif
// Weight-gradient jobs whose `A` operand is a CALLER's tensor -- every job of bsms_mlp_bwd, and in a GMP block the x half of the
// first node Linear and the two projections (A = x) -- run the range-free three-way bf16 split (wgrad.hip: BF3, six products)
// instead of fp16 x 2 pieces with one scale per tensor: a caller's columns may differ by any factor (VERDICT
// round 4 item 2).  The other node-level jobs multiply the block's own post-ReLU activations / LayerNorm-bounded sums, like
// the edge-level ones, and keep the three-product arithmetic: all node-level jobs range-free measured 0 on the airfoil step
// but -2.6 % / -7.5 % on the cylinder steps (dense / block-diagonal), where the side lane is co-critical
// (profiles/r05_node_bf3_ab.txt).  Experiment builds: BSMS_NODE_BF3 = 0 none / 1 this / 2 all node-level jobs.
#ifdef BSMS_EXPERIMENTS
static int env_node_bf3() { const char* e = getenv("BSMS_NODE_BF3"); return e ? atoi(e) : 1; }
static const int g_node_bf3 = env_node_bf3();
#else
constexpr int g_node_bf3 = 1;
#endif
#ifdef BSMS_EXPERIMENTS
static int env_fwd_res() { const char* e = getenv("BSMS_EDGE_FWD_RES"); return e ? atoi(e) : 1; }
static const int g_edge_fwd_res = env_fwd_res();   // A/B against the generic ring kernel (profiles/r05_efwd_ab.sh)
#else
constexpr int g_edge_fwd_res = 1;
#endif
// The fp32 fused edge backward (csrc/experiments/efuse32.hip, round 6: dgrad + dW on chip without forward recompute) is NOT in the
// product library: same-box 176-180 against 186.5 steps/s for the unfused dataflow (profiles/r06_efuse32_notes.txt).  Experiment
// builds (profiles/build_efv.sh) link it and switch it on with BSMS_EDGE_FUSED_F32=1.
#ifdef BSMS_EXPERIMENTS
static int env_fused32() { const char* e = getenv("BSMS_EDGE_FUSED_F32"); return e ? atoi(e) : 0; }
static const int g_edge_fused32 = env_fused32();
static bool fused32_on(int64_t D, int H, int precision) { return g_edge_fused32 && edge_fused32_supported(D, H, 1, precision); }
#else
static bool fused32_on(int64_t, int, int) { return false; }
#endif
bool use_edge_fused(int64_t D, int H, int precision) {
  return precision == BSMS_F32 ? fused32_on(D, H, precision) : (g_edge_fused && edge_fused_supported(D, H, 1, precision));
}

bool edge_fused_possible(int64_t D, int H) { return use_edge_fused(D, H, BSMS_F32) || use_edge_fused(D, H, BSMS_BF16) || use_edge_fused(D, H, BSMS_BF16_NODES); }

int check_gmp(const bsms_plan_t* plan, int64_t B, int64_t D, int64_t p, int H, const char* who) {
  BSMS_REQUIRE(plan != nullptr, BSMS_E_INVALID_ARG, "%s: plan is null", who);
  BSMS_REQUIRE(supported_D(D), BSMS_E_UNSUPPORTED, "%s: latent width D=%lld not supported (32, 64, 128, 256)", who, (long long)D);
  BSMS_REQUIRE(p >= 1 && p <= 7, BSMS_E_UNSUPPORTED, "%s: pos_dim=%lld (1..7)", who, (long long)p);
  BSMS_REQUIRE(H >= 1 && H < kMaxStages, BSMS_E_UNSUPPORTED, "%s: hidden=%d (1..%d)", who, H, kMaxStages - 1);
  BSMS_REQUIRE(B >= 0, BSMS_E_SHAPE, "%s: B=%lld", who, (long long)B);
  BSMS_REQUIRE(B * std::max(plan->E, plan->N) * D < (int64_t(1) << 40), BSMS_E_SHAPE, "%s: tensor too large", who);
  return BSMS_OK;
}

}  // namespace

// =================================================================================== GMP entries
#ifdef BSMS_EXPERIMENTS
extern "C" void bsms_debug_set_flags(int flags) { g_debug_flags = flags; }  // not in bsms_hip.h: experiments only
extern "C" void bsms_debug_set_timing(unsigned long long* dev_buf) { g_timing = dev_buf; }
#endif

extern "C" size_t bsms_gmp_saved_bytes(int64_t B, int64_t N, int64_t E, int64_t D, int hidden) {
  if (hidden < 1 || hidden >= kMaxStages) return 0;
  return carve_gmp_saved(nullptr, B, N, E, D, hidden).bytes;
}
extern "C" size_t bsms_gmp_work_bytes(int64_t B, int64_t N, int64_t E, int64_t D, int hidden) {
  if (hidden < 1 || hidden >= kMaxStages) return 0;
  return carve_gmp_work(nullptr, B, N, E, D, hidden).bytes;
}

namespace {
// where the saved tensors / packs of a forward call live (see carve_gmp_saved)
GmpSaved locate_saved(void* saved, const GmpWork& wk, int64_t B, int64_t N, int64_t E, int64_t D, int H, void* packs_base,
                      bool bf = false, bool bfn = false) {
  return saved ? carve_gmp_saved(saved, B, N, E, D, H, true, nullptr, bf, bfn, use_edge_fused(D, H, bf ? (bfn ? BSMS_BF16_NODES : BSMS_BF16) : BSMS_F32))
               : carve_gmp_saved(wk.gN[0], B, N, E, D, H, false, packs_base, bf, bfn);  // inference: lives in the gradient scratch
}

int prepack_block(const GmpSaved& sv, int64_t D, int64_t p, int H, bool training, const float* const* params, hipStream_t s,
                  bool bf = false, bool bfn = false) {
  const int nl = H + 1;
  const float* const* pn = params;            // mlp_node: W_l = pn[2l], b_l = pn[2l+1]
  const float* const* pe = params + 2 * nl;   // mlp_edge
  const int ldE0 = int(2 * D + p + 1);
  PackTable t{};
  add_pack(t, pe[0], ldE0, 0, int(p + 1), (int)D, (int)D, PACK_FRAG, sv.e_wi, pe[1]);
  add_pack(t, pe[0], ldE0, 0, int(p + 1 + D), (int)D, (int)D, PACK_FRAG, sv.e_wj);
  add_pack(t, pe[0], ldE0, 0, 0, (int)D, int(p + 1), PACK_TRANSPOSE, sv.e_wft);
  if (training) {   // grad_x += dPs Wi + dPd Wj runs as ONE Linear over [dPs, dPd]: the two packs share their scale (chain.h: mate)
    add_pack(t, pe[0], ldE0, 0, int(p + 1), (int)D, (int)D, PACK_FRAG_T, sv.e_wit);
    add_pack(t, pe[0], ldE0, 0, int(p + 1 + D), (int)D, (int)D, PACK_FRAG_T, sv.e_wjt);
    t.d[t.n - 2].mate = t.n;        // 1 + index of the other
    t.d[t.n - 1].mate = t.n - 1;
  }
  for (int l = 1; l <= H; ++l) {   // the D x D Linears of the edge MLP: bf16 operands in the bf16 precision
    add_pack(t, pe[2 * l], (int)D, 0, 0, (int)D, (int)D, PACK_FRAG, sv.e_w[l], pe[2 * l + 1]);
    t.d[t.n - 1].bf16 = bf;
    if (training && sv.e_wt[l]) { add_pack(t, pe[2 * l], (int)D, 0, 0, (int)D, (int)D, PACK_FRAG_T, sv.e_wt[l]); t.d[t.n - 1].bf16 = bf; }
    if (training && sv.e_wr[l]) add_pack(t, pe[2 * l], (int)D, 0, 0, (int)D, (int)D, PACK_ROWS_BF16, sv.e_wr[l]);   // fused edge backward (efuse.hip)
  }
  // first node Linear over [x, aggr]: two packs, one scale; the bias rides in the second one (where the stage finishes).
  // BSMS_BF16_NODES: one-plane packs of the rounded weights; a bf16 stage STARTS from its pack's bias (chunk 0 header), so
  // the bias rides in the first pack there
  add_pack(t, pn[0], int(2 * D), 0, 0, (int)D, (int)D, PACK_FRAG, sv.n_w0x, bfn ? pn[1] : nullptr);
  add_pack(t, pn[0], int(2 * D), 0, (int)D, (int)D, (int)D, PACK_FRAG, sv.n_w0a, bfn ? nullptr : pn[1]);
  t.d[t.n - 2].mate = t.n;
  t.d[t.n - 1].mate = t.n - 1;
  t.d[t.n - 2].bf16 = t.d[t.n - 1].bf16 = bfn;
  if (training) {
    add_pack(t, pn[0], int(2 * D), 0, 0, (int)D, (int)D, PACK_FRAG_T, sv.n_w0xt);
    add_pack(t, pn[0], int(2 * D), 0, (int)D, (int)D, (int)D, PACK_FRAG_T, sv.n_w0at);
    t.d[t.n - 2].bf16 = t.d[t.n - 1].bf16 = bfn;
  }
  for (int l = 1; l <= H; ++l) {
    add_pack(t, pn[2 * l], (int)D, 0, 0, (int)D, (int)D, PACK_FRAG, sv.n_w[l], pn[2 * l + 1]);
    t.d[t.n - 1].bf16 = bfn;
    if (training) { add_pack(t, pn[2 * l], (int)D, 0, 0, (int)D, (int)D, PACK_FRAG_T, sv.n_wt[l]); t.d[t.n - 1].bf16 = bfn; }
  }
  t.zero = training ? sv.bound : nullptr;
  return launch_prepack(t, s);
}
}  // namespace

size_t bsms::gmp_pack_bytes(int64_t D, int hidden) {   // inference packs of one block (upper bound: every take() is 256-byte aligned)
  return (size_t(2 * hidden + 4) * (pack_floats(D) * sizeof(float) + 256)) + size_t(8) * D * sizeof(float) + 256;
}

int bsms::gmp_prepack(int64_t B, int64_t N, int64_t E, int64_t D, int64_t p, int H, const float* const* params, void* saved,
                      void* work, void* packs_base, hipStream_t s, int precision) {
  GmpWork wk = carve_gmp_work(work, B, N, E, D, H);
  GmpSaved sv = locate_saved(saved, wk, B, N, E, D, H, packs_base, precision != BSMS_F32, precision == BSMS_BF16_NODES);
  return prepack_block(sv, D, p, H, saved != nullptr, params, s, precision != BSMS_F32, precision == BSMS_BF16_NODES);
}
size_t bsms::gmp_saved_bytes_p(int64_t B, int64_t N, int64_t E, int64_t D, int hidden, int precision) {
  if (hidden < 1 || hidden >= kMaxStages) return 0;
  return carve_gmp_saved(nullptr, B, N, E, D, hidden, true, nullptr, precision != BSMS_F32, precision == BSMS_BF16_NODES,
                         use_edge_fused(D, hidden, precision)).bytes;
}

extern "C" int bsms_gmp_fwd(const bsms_plan_t* plan, const float* x, const float* pos, int64_t B, int64_t D, int64_t p,
                            int64_t pos_bstride, int H, const float* const* params, float* out, void* saved,
                            void* work, bsms_stream_t stream) {
  return bsms::gmp_fwd_core(plan, x, pos, B, D, p, pos_bstride, H, params, out, saved, work, nullptr, true, nullptr,
                            as_stream(stream), BSMS_F32);
}

int bsms::gmp_fwd_core(const bsms_plan* plan, const float* x, const float* pos, int64_t B, int64_t D, int64_t p,
                       int64_t pos_bstride, int H, const float* const* params, float* out, void* saved, void* work,
                       void* packs_base, bool do_prepack, const float* resid2, hipStream_t s, int precision) {
  int rc = check_gmp(plan, B, D, p, H, "gmp_fwd");
  if (rc) return rc;
  BSMS_REQUIRE(x && pos && params && out && work, BSMS_E_INVALID_ARG, "gmp_fwd: null argument");
  const bool bf = precision != BSMS_F32, bfn = precision == BSMS_BF16_NODES;
  BSMS_REQUIRE(!bf || ((D == 128 || D == 256) && p <= 3), BSMS_E_UNSUPPORTED, "gmp_fwd: bf16 precision needs D = 128 / 256 and pos_dim <= 3");
  const int64_t N = plan->N, E = plan->E;
  const bool training = saved != nullptr;     // saved == NULL: inference, nothing is kept for a backward
  GmpWork wk = carve_gmp_work(work, B, N, E, D, H);
  GmpSaved sv = locate_saved(saved, wk, B, N, E, D, H, packs_base, bf, bfn);
  if (do_prepack && (rc = prepack_block(sv, D, p, H, training, params, s, bf, bfn))) return rc;
  const bool fused_any = training && use_edge_fused(D, H, precision);
  const bool fused = fused_any && bf;       // the backward recomputes the edge activations (efuse.hip): the projections are kept instead
  const bool fused32 = fused_any && !bf;    // the activations are saved as fp16 x 2 pieces for efuse32.hip
  float* const Ps = fused ? sv.e_Ps : wk.Ps;
  float* const Pd = fused ? sv.e_Pd : wk.Pd;

  // node pre-projections
  {
    ChainFwdArgs a{};
    a.R = B * N; a.x = x; a.nstage = 2;   // both projections in one launch (OUT_PLAIN2); bias b0 rides in the first pack
    a.wp[0] = reinterpret_cast<const float4*>(sv.e_wi); a.y = Ps;
    a.wp[1] = reinterpret_cast<const float4*>(sv.e_wj); a.y2 = Pd;
    if ((rc = launch_chain_fwd((int)D, IN_ROWS, OUT_PLAIN2, a, s))) return rc;
  }
  // edge MLP + LayerNorm
  // bf16 precisions at D = 128, hidden = 3: the kernel with LDS-resident weights (efwd.hip) -- whenever no edge activation has to
  // be saved, i.e. inference and the training forward of the fused backward (efuse.hip recomputes them)
  if (g_edge_fwd_res && edge_fwd_res_supported(D, H, p, precision) && (!training || fused)) {
    EdgeFwdResArgs a{};
    a.R = B * E; a.E = (int32_t)E; a.N = (int32_t)N; a.src = plan->src; a.dst = plan->dst;
    a.Ps = Ps; a.Pd = Pd; a.pos = pos; a.pos_bstride = pos_bstride; a.p = (int)p; a.wft = sv.e_wft;
    for (int l = 1; l <= 3; ++l) a.wp[l - 1] = reinterpret_cast<const float4*>(sv.e_w[l]);
    a.y = sv.e_y; a.rstd = training ? sv.e_rstd : nullptr; a.fiber_out = training ? sv.e_fiber : nullptr;
    if ((rc = launch_edge_fwd_res(a, s))) return rc;
  } else {
    ChainFwdArgs a{};
    a.R = B * E; a.K0 = int(p + 1); a.w0t = sv.e_wft; a.store_in = fused ? nullptr : sv.e_act[0];
    a.src = plan->src; a.dst = plan->dst; a.E = (int32_t)E; a.N = (int32_t)N;
    a.Ps = Ps; a.Pd = Pd; a.pos = pos; a.pos_bstride = pos_bstride; a.p = (int)p;
    a.fiber_out = training ? sv.e_fiber : nullptr;
    a.bf16 = bf;
    a.nstage = H;
    for (int st = 0; st < H; ++st) {
      a.wp[st] = reinterpret_cast<const float4*>(sv.e_w[st + 1]);
      a.store[st] = (training && !fused && st < H - 1) ? sv.e_act[st + 1] : nullptr;
    }
    if (fused32) {
      a.pieces = 1;
      for (int st = 0; st < H; ++st) a.store_exp[st] = sv.e_aexp[st];
    }
    a.y = sv.e_y; a.rstd = sv.e_rstd;
    if (training && !bf) for (int st = 0; st < H; ++st) a.amax[st] = sv.bound + size_t(st) * kBoundWidth;
    a.timing = (g_debug_flags & 512) ? nullptr : g_timing;   // experiments: bit 9 hands the stamp buffer to the node chain instead
    a.store_mode = ((g_debug_flags & 2) ? 0 : 1) | ((g_debug_flags & 64) ? 4 : 0) | ((g_debug_flags & 128) ? 8 : 0);   // non-temporal stores (experiments: +4 no sign bits, +8 unpaired 64-byte pieces)
    a.out_mode = (g_debug_flags & 4) ? 1 : 0;
    if (g_debug_flags & 1) {
      a.store_in = nullptr;
      for (int st = 0; st < H; ++st) a.store[st] = nullptr;
    }
    if ((rc = launch_chain_fwd((int)D, IN_EDGE, OUT_LN, a, s))) return rc;
  }
  // aggregation (scatter_sum over targets, ops/basic.py:94)
  if ((rc = bf ? rowsum_plan_order_bf16(plan, sv.e_y, B, D, sv.aggr, s) : rowsum_plan_order(plan, sv.e_y, B, D, sv.aggr, s))) return rc;
  // node MLP + LayerNorm + residual
  {
    ChainFwdArgs a{};
    a.R = B * N; a.x = x; a.x2 = sv.aggr; a.nstage = H + 1;
    a.wp[0] = reinterpret_cast<const float4*>(sv.n_w0x);
    a.wp0b = reinterpret_cast<const float4*>(sv.n_w0a);
    a.store[0] = training ? sv.n_act[0] : nullptr;
    for (int st = 1; st <= H; ++st) {
      a.wp[st] = reinterpret_cast<const float4*>(sv.n_w[st]);
      a.store[st] = (training && st < H) ? sv.n_act[st] : nullptr;
    }
    a.y = out; a.yln = sv.n_yln; a.rstd = sv.n_rstd; a.resid = x; a.resid2 = resid2;
    a.bf16 = bfn;
    if (g_node_bf3 < 2) {   // bounds for the fp16 x 2 weight-gradient jobs of the node Linears
      if (training && !bfn) for (int st = 0; st <= H; ++st) a.amax[st] = sv.bound + size_t(8 + st) * kBoundWidth;
      else if (training) a.amax[0] = sv.bound + size_t(8) * kBoundWidth;   // BSMS_BF16_NODES: only [x, aggr] feed an fp32 weight-gradient job
    }
    a.store_mode = 1;
    a.timing = (g_debug_flags & 512) ? g_timing : nullptr;
    if ((rc = launch_chain_fwd((int)D, IN_ROWS2, OUT_LN, a, s))) return rc;
  }
  return BSMS_OK;
}

extern "C" int bsms_gmp_bwd(const bsms_plan_t* plan, const float* x, const float* pos, const float* grad_out, int64_t B,
                            int64_t D, int64_t p, int64_t pos_bstride, int H, const float* const* params,
                            const void* saved, void* work, float* grad_x, float* const* grads, bsms_stream_t stream) {
  return bsms::gmp_bwd_core(plan, x, pos, grad_out, B, D, p, pos_bstride, H, params, saved, work, grad_x, grads, -1,
                            as_stream(stream), BSMS_F32);
}

namespace {
// joins (or marks) the side lanes on EVERY exit path of gmp_bwd_core: an error return between fork and join would
// otherwise leave the lane dangling, which also breaks an in-progress HIP-graph capture
struct LaneScope {
  SideLane* lane = nullptr;
  hipStream_t main;
  int slot;
  explicit LaneScope(hipStream_t m, int defer_slot) : main(m), slot(defer_slot) {}
  int finish() {
    SideLane* l = lane;
    lane = nullptr;
    if (!l) return BSMS_OK;
    return slot < 0 ? side_join(l, main) : side_mark(l, slot);
  }
  ~LaneScope() { if (lane) side_join(lane, main); }
};
}  // namespace

// deferred-join mode: does lane 1's mark cover lane 0's (side_mark_chain)?  True unless an experiment switch disables a lane.
bool bsms::gmp_marks_chained() { return !(g_debug_flags & 8) && !(g_debug_flags & 32); }

int bsms::gmp_bwd_core(const bsms_plan* plan, const float* x, const float* pos, const float* grad_out, int64_t B, int64_t D,
                       int64_t p, int64_t pos_bstride, int H, const float* const* params, const void* saved, void* work,
                       float* grad_x, float* const* grads, int defer_slot, hipStream_t s, int precision) {
  int rc = check_gmp(plan, B, D, p, H, "gmp_bwd");
  if (rc) return rc;
  const bool bf = precision != BSMS_F32, bfn = precision == BSMS_BF16_NODES;
  BSMS_REQUIRE(!bf || ((D == 128 || D == 256) && p <= 3), BSMS_E_UNSUPPORTED, "gmp_bwd: bf16 precision needs D = 128 / 256 and pos_dim <= 3");
  BSMS_REQUIRE(x && pos && grad_out && params && saved && work && grad_x && grads, BSMS_E_INVALID_ARG,
               "gmp_bwd: null argument");
  const int64_t N = plan->N, E = plan->E;
  const int nl = H + 1;
  float* const* gn = grads;
  float* const* ge = grads + 2 * nl;
  const int ldE0 = int(2 * D + p + 1);
  const bool fused = use_edge_fused(D, H, precision);
  GmpSaved sv = carve_gmp_saved(const_cast<void*>(saved), B, N, E, D, H, true, nullptr, bf, bfn, fused);
  GmpWork wk = carve_gmp_work(work, B, N, E, D, H);
  int ef_nwg = 0;

  // node MLP backward: grad_x = grad_out (residual) + g0 W0x ; daggr = g0 W0a
  {
    ChainBwdArgs a{};
    a.R = B * N; a.dy = grad_out; a.yln = sv.n_yln; a.rstd = sv.n_rstd; a.store_mode = 1;
    a.nstage = H;
    a.gstore[0] = wk.gN[H];
    for (int k = 0; k < H; ++k) {
      a.wpt[k] = reinterpret_cast<const float4*>(sv.n_wt[H - k]);
      a.mask[k] = sv.n_act[H - k - 1];
      a.gstore[k + 1] = wk.gN[H - k - 1];
    }
    a.wh0 = reinterpret_cast<const float4*>(sv.n_w0xt);
    a.wh1 = reinterpret_cast<const float4*>(sv.n_w0at);
    a.dx = grad_x; a.dx2 = wk.daggr; a.dres = grad_out;
    a.bf16 = bfn;
    if (g_node_bf3 < 2) {
      if (!bfn) for (int k = 0; k <= H; ++k) a.gmax[k] = sv.bound + size_t(24 + k) * kBoundWidth;
      else a.gmax[H] = sv.bound + size_t(24 + H) * kBoundWidth;   // BSMS_BF16_NODES: only gN[0] (fp32) feeds an fp32 weight-gradient job
    }
    if ((rc = launch_chain_bwd((int)D, G_ROWS_LN, F_HEADS2, a, s))) return rc;
  }
  // edge MLP backward (gradient of the aggregation = gather by target)
#ifdef BSMS_EXPERIMENTS
  if (fused && !bf) {   // fp32 (experiments/efuse32.hip): dgrad chain + dW on chip, A tiles from the saved pieces by LDS-DMA, running block exponents
    EdgeFused32Args a{};
    a.R = B * E; a.E = (int32_t)E; a.N = (int32_t)N; a.dst = plan->dst;
    for (int k = 0; k < 3; ++k) {   // execution order: Linear 3, 2, 1
      a.wseq[k] = reinterpret_cast<const float4*>(sv.e_wt[3 - k]);
      a.act[k] = sv.e_act[2 - k];
      a.aexp[k] = sv.e_aexp[2 - k];
    }
    a.dy = wk.daggr; a.y = sv.e_y; a.rstd = sv.e_rstd; a.g0 = wk.gE[0];
    a.part = wk.ef_part;
    a.timing = g_timing;
    if ((rc = launch_edge_fused32_bwd(a, &ef_nwg, s))) return rc;
  } else
#endif
  if (fused) {   // recompute + LayerNorm backward + dgrad + the weight gradients of Linears 1..H on chip (efuse.hip)
    EdgeFusedBwdArgs a{};
    a.R = B * E; a.E = (int32_t)E; a.N = (int32_t)N; a.src = plan->src; a.dst = plan->dst;
    a.Ps = sv.e_Ps; a.Pd = sv.e_Pd; a.fiber = sv.e_fiber; a.wft = sv.e_wft; a.p = (int)p;
    const float* const* pe = params + 2 * nl;
    for (int l = 1; l <= 3; ++l) { a.wr[l - 1] = reinterpret_cast<const float4*>(sv.e_wr[l]); a.b[l - 1] = pe[2 * l + 1]; }
    a.dy = wk.daggr; a.y = sv.e_y; a.rstd = sv.e_rstd; a.g0 = wk.gE[0];
    a.gmax = g_node_bf3 ? nullptr : sv.bound + size_t(16 + H) * kBoundWidth;   // only the projections' weight gradients used it
    a.part = wk.ef_part;
    a.timing = g_timing;
    if ((rc = launch_edge_fused_bwd(a, &ef_nwg, s))) return rc;
  } else {
    ChainBwdArgs a{};
    a.R = B * E; a.dy = wk.daggr; a.yln = sv.e_y; a.rstd = sv.e_rstd; a.store_mode = ((g_debug_flags & 2) ? 0 : (g_debug_flags & 256) ? 2 : 1) | ((g_debug_flags & 128) ? 8 : 0);
    a.dst = plan->dst; a.E = (int32_t)E; a.N = (int32_t)N;
    a.bf16 = bf;
    a.nstage = H;
    a.gstore[0] = wk.gE[H];
    for (int k = 0; k < H; ++k) {
      a.wpt[k] = reinterpret_cast<const float4*>(sv.e_wt[H - k]);
      a.mask[k] = sv.e_act[H - k - 1];
      a.gstore[k + 1] = wk.gE[H - k - 1];
    }
    if (!bf) for (int k = 0; k <= (g_node_bf3 ? H - 1 : H); ++k) a.gmax[k] = sv.bound + size_t(16 + k) * kBoundWidth;   // gE[0] feeds only the projections' jobs
    else if (!g_node_bf3) a.gmax[H] = sv.bound + size_t(16 + H) * kBoundWidth;   // bf16 precision: only gE[0] (fp32 scatter sums dPs / dPd feed fp32 weight-gradient jobs)
    a.ablate = (g_debug_flags & 2048) ? 1 : 0;   // experiments: bit 11 = no stream of gE[1..H] (with bits 0 and 10: the traffic a fused backward would not have)
    if ((rc = launch_chain_bwd((int)D, G_EDGE_LN, F_NONE, a, s))) return rc;
  }
  // From here two independent strands run CONCURRENTLY (fork/join on an internal side stream):
  //   side: the MFMA-bound batched weight gradients of all D x D layers (reads gE[1..H], gN[*], saved activations)
  //   main: the HBM-bound strand -- scatter of gE[0] to the two projections, fiber/bias gradients of the first edge
  //         Linear, the projection weight gradients and finally the input gradient.
  // They touch disjoint outputs; main joins the side stream before returning, so callers see ordinary semantics.
  // (Forking the node-layer gradients earlier, against the edge backward chain, measured SLOWER: two MFMA-bound
  // kernels contend; an MFMA-bound kernel against HBM-bound ones is the pairing that pays: +5.6 % steps/s.)
  SideLane* lane = nullptr;
  const bool overlap = !(g_debug_flags & 8);
  hipStream_t ws = s;
  LaneScope scope1(s, defer_slot), scope2(s, defer_slot);
  if (overlap) {
    if ((rc = side_lane(&lane, 0, s)) || (rc = side_fork(lane, s))) return rc;
    ws = lane->stream;
    scope1.lane = lane;
  }
  {
    WgradJob jobs[kMaxWgradJobs] = {};
    int nj = 0;
    // every operand comes with a magnitude bound from the chain kernel that produced or consumed it (GmpSaved::bound)
    auto add_job = [&](const float* G, const float* A, float* dW, float* db, int64_t R, int ldw, int col0, const float* gb, const float* ab) {
      WgradJob& j = jobs[nj++];
      j.G = G; j.A = A; j.dW = dW; j.db = db; j.R = R; j.ldg = (int)D; j.lda = (int)D; j.ldw = ldw; j.col0 = col0; j.bf16 = 0;
      j.g_bound = gb; j.a_bound = ab; j.g_mul = j.a_mul = 1.f;
    };
    auto bd = [&](int slot) { return sv.bound + size_t(slot) * kBoundWidth; };
    if (fused) {   // the edge Linears' gradients were accumulated by the fused kernel: only the fixed-order sum of its partials is left
      float* const dWs[3] = {ge[2], ge[4], ge[6]};
      float* const dbs[3] = {ge[3], ge[5], ge[7]};
      if ((rc = launch_edge_fused_reduce(wk.ef_part, ef_nwg, dWs, dbs, ws))) return rc;
    }
    for (int l = 1; l <= H && !fused && !(g_debug_flags & 1024); ++l) {   // edge Linears: bf16 gradient and activation tensors in the bf16 precision (experiments: bit 10 drops these jobs)
      add_job(wk.gE[l], sv.e_act[l - 1], ge[2 * l], ge[2 * l + 1], B * E, (int)D, 0, bd(16 + (H - l)), bd(l - 1));
      jobs[nj - 1].bf16 = bf;
    }
    // no bounds = the range-free arithmetic: the job over the caller's x (g_node_bf3 >= 1), every node-level job (2)
    auto nbd = [&](int slot, int level) -> const float* { return g_node_bf3 >= level ? nullptr : bd(slot); };
    for (int l = 1; l <= H; ++l) {   // node Linears 1..H: bf16 tensors in BSMS_BF16_NODES
      add_job(wk.gN[l], sv.n_act[l - 1], gn[2 * l], gn[2 * l + 1], B * N, (int)D, 0, nbd(24 + (H - l), 2), nbd(8 + l, 2));
      jobs[nj - 1].bf16 = bfn;
    }
    // the job over the caller's x (with the first node Linear's bias gradient): range-free arithmetic = another launch, so
    // at level 1 it rides in lane 2's launch with the two projections (all three range-free) and this lane stays ONE launch of
    // three-product jobs -- two launches here cost the cylinder steps 1.6 % / 6 % (side lane co-critical, r05_node_bf3_ab.txt)
    if (g_node_bf3 != 1) add_job(wk.gN[0], x, gn[0], gn[1], B * N, int(2 * D), 0, nbd(24 + H, 1), nbd(8, 1));
    add_job(wk.gN[0], sv.aggr, gn[0], nullptr, B * N, int(2 * D), (int)D, nbd(24 + H, 2), nbd(8, 2));
    if ((rc = launch_wgrad((int)D, jobs, nj, wk.wg, ws))) return rc;
  }
  // a second side stream takes the remaining weight gradients of the first edge Linear (fiber columns + bias now,
  // the x-columns once dPs/dPd exist), so that the caller's stream only carries what grad_x depends on:
  // gE[0] -> dPs, dPd -> grad_x
  SideLane* lane2 = nullptr;
  hipStream_t s2 = s;
  const bool overlap2 = overlap && !(g_debug_flags & 32);
  bool lane2_forked = false;   // its first fork is only needed by the unfused small-wgrad path; the fused path forks once, below
  if (overlap2) {
    if ((rc = side_lane(&lane2, 1, s))) return rc;
    s2 = lane2->stream;
  }
  // gradient of the first edge Linear w.r.t. the two per-node projections -- and, in the same pass over gE[0], the
  // partial sums of its fiber columns and bias (rowsum.hip: k_rowsum_pair_fiber); only their small reduction is left
  SmallWgradArgs sw{};
  sw.G = wk.gE[0]; sw.S = sv.e_fiber; sw.S_cols = int(p + 1); sw.S_ld = fiber_ld(p);   // the fiber rows the forward kept
  sw.p = (int)p;
  sw.out = ge[0]; sw.os = 1; sw.of = ldE0; sw.colsum = ge[1];
  sw.R = B * E; sw.D = (int)D;
  int nwg = 0;
  if ((rc = rowsum_source_target_fiber(plan, wk.gE[0], B, D, wk.dPs, wk.dPd, sv.e_fiber, fiber_ld(p), int(p + 1),
                                       reinterpret_cast<float*>(wk.sw), small_wgrad_part_blocks(wk.sw_bytes, (int)D), &nwg, s, bf))) return rc;
  BSMS_REQUIRE(nwg > 0 || !bf, BSMS_E_UNSUPPORTED, "gmp_bwd: bf16 precision needs the fused scatter kernel (D = 128 / 256, pos_dim <= 3)");
  if (nwg == 0) {   // shape not built into the fused kernel (D < 128, pos_dim > 3): separate passes
    if (overlap2) {
      if ((rc = side_fork(lane2, s))) return rc;
      scope2.lane = lane2;
      lane2_forked = true;
    }
    if ((rc = rowsum_source_and_target(plan, wk.gE[0], B, D, wk.dPs, wk.dPd, s))) return rc;
    if ((rc = launch_small_wgrad(sw, wk.sw, s2))) return rc;
  }
  (void)lane2_forked;
  // x-columns of W0_edge (the two projections)
  if (overlap2) {   // lane 2 waits for dPs / dPd (and the partials)
    if ((rc = side_fork(lane2, s))) return rc;
    scope2.lane = lane2;
  }
  if (nwg > 0 && (rc = launch_small_reduce(sw, wk.sw, nwg, s2))) return rc;
  {
    WgradJob jobs[3] = {};
    auto set = [&](WgradJob& j, const float* G, int col0) {
      j.G = G; j.A = x; j.dW = ge[0]; j.db = nullptr; j.R = B * N; j.ldg = (int)D; j.lda = (int)D; j.ldw = ldE0; j.col0 = col0; j.bf16 = 0;
      // dPs / dPd are sums of at most max-degree rows of gE[0]; x is covered by the joint bound of the node chain's input
      j.g_bound = g_node_bf3 ? nullptr : sv.bound + size_t(16 + H) * kBoundWidth;
      j.a_bound = g_node_bf3 ? nullptr : sv.bound + size_t(8) * kBoundWidth; j.a_mul = 1.f;
    };
    set(jobs[0], wk.dPs, int(p + 1));
    set(jobs[1], wk.dPd, int(p + 1 + D));
    jobs[0].g_mul = float(std::max<int64_t>(plan->max_out_degree, 1));   // scatter by source
    jobs[1].g_mul = float(std::max<int64_t>(plan->max_in_degree, 1));    // scatter by target
    int nj2 = 2;
    if (g_node_bf3 == 1) {   // + the x half of the first node Linear (see above)
      WgradJob& j = jobs[nj2++];
      j.G = wk.gN[0]; j.A = x; j.dW = gn[0]; j.db = gn[1]; j.R = B * N; j.ldg = (int)D; j.lda = (int)D; j.ldw = int(2 * D); j.col0 = 0;
      j.bf16 = 0; j.g_bound = j.a_bound = nullptr; j.g_mul = j.a_mul = 1.f;
    }
    if ((rc = launch_wgrad((int)D, jobs, nj2, wk.wg2, s2))) return rc;
  }
  // grad_x += dPs Wi + dPd Wj
  {
    ChainFwdArgs a{};
    a.R = B * N; a.x = wk.dPs; a.x2 = wk.dPd; a.nstage = 1;
    a.wp[0] = reinterpret_cast<const float4*>(sv.e_wit);
    a.wp0b = reinterpret_cast<const float4*>(sv.e_wjt);
    a.y = grad_x; a.accumulate = 1;
    if ((rc = launch_chain_fwd((int)D, IN_ROWS2, OUT_PLAIN, a, s))) return rc;
  }
  if (defer_slot >= 0 && scope1.lane && scope2.lane) {   // deferred join: one event covers both lanes (lane 2's mark waits for lane 1's)
    SideLane *a = scope1.lane, *b = scope2.lane;
    scope1.lane = scope2.lane = nullptr;
    return side_mark_chain(a, b, defer_slot);
  }
  if ((rc = scope1.finish()) || (rc = scope2.finish())) return rc;
  return BSMS_OK;
}

// =================================================================================== MLP entries
namespace {

enum MlpKind { MLP_SMALL_LN, MLP_ROWS_LN, MLP_ROWS_SMALL, MLP_BAD };
MlpKind mlp_kind(int64_t in_dim, int64_t D, int64_t out_dim, int layer_norm) {
  const bool in_small = in_dim >= 1 && in_dim <= 8 && in_dim != D, in_rows = in_dim == D;
  if (layer_norm && out_dim == D) return in_small ? MLP_SMALL_LN : (in_rows ? MLP_ROWS_LN : MLP_BAD);
  if (!layer_norm && out_dim >= 1 && out_dim <= 8 && in_rows) return MLP_ROWS_SMALL;
  return MLP_BAD;
}

struct MlpSaved {
  float *act[kMaxStages], *yln, *rstd;
  float *w[kMaxStages + 1], *wt[kMaxStages + 1], *w0t;
  float* bound;   // training: magnitude bounds, [st] tensor entering forward MFMA stage st, [16 + k] gradient entering backward stage k
  size_t bytes;
};
MlpSaved carve_mlp_saved(void* base, int64_t R, int64_t D, int H, bool training = true) {
  Carver c(base);
  MlpSaved s{};
  if (training) {
    for (int l = 0; l < H; ++l) s.act[l] = c.take(act_floats(size_t(R), D));
    s.yln = c.take(size_t(R) * D);
    s.rstd = c.take(size_t(R));
    s.bound = c.take(size_t(kBoundSlots) * kBoundWidth);
  }
  for (int l = 0; l <= H; ++l) { s.w[l] = c.take(pack_floats(D)); if (training) s.wt[l] = c.take(pack_floats(D)); }
  s.w0t = c.take(size_t(16) * D);
  s.bytes = c.off;
  return s;
}
struct MlpWork {
  float* g[kMaxStages + 1];
  char *wg, *sw;
  size_t bytes;
};
MlpWork carve_mlp_work(void* base, int64_t R, int64_t D, int H) {
  Carver c(base);
  MlpWork w{};
  for (int l = 0; l <= H; ++l) w.g[l] = c.take(pad_rows(size_t(R)) * D);
  w.wg = c.take_bytes(wgrad_work_bytes((int)D, 0));
  w.sw = c.take_bytes(small_wgrad_work_bytes((int)D));
  w.bytes = c.off;
  return w;
}

int check_mlp(int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int H, int layer_norm, const char* who) {
  BSMS_REQUIRE(supported_D(D), BSMS_E_UNSUPPORTED, "%s: latent width D=%lld not supported (32, 64, 128, 256)", who, (long long)D);
  BSMS_REQUIRE(H >= 1 && H < kMaxStages, BSMS_E_UNSUPPORTED, "%s: hidden=%d (1..%d)", who, H, kMaxStages - 1);
  BSMS_REQUIRE(R >= 0, BSMS_E_SHAPE, "%s: R=%lld", who, (long long)R);
  BSMS_REQUIRE(mlp_kind(in_dim, D, out_dim, layer_norm) != MLP_BAD, BSMS_E_UNSUPPORTED,
               "%s: MLP shape in=%lld D=%lld out=%lld ln=%d not supported", who, (long long)in_dim, (long long)D,
               (long long)out_dim, layer_norm);
  return BSMS_OK;
}

}  // namespace

extern "C" size_t bsms_mlp_saved_bytes(int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int hidden) {
  (void)in_dim; (void)out_dim;
  if (hidden < 1 || hidden >= kMaxStages) return 0;
  return carve_mlp_saved(nullptr, R, D, hidden).bytes;
}
extern "C" size_t bsms_mlp_work_bytes(int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int hidden) {
  (void)in_dim; (void)out_dim;
  if (hidden < 1 || hidden >= kMaxStages) return 0;
  return carve_mlp_work(nullptr, R, D, hidden).bytes;
}

extern "C" int bsms_mlp_fwd(const float* x, int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int H, int layer_norm,
                            const float* const* params, float* y, void* saved, void* work, bsms_stream_t stream) {
  return bsms_mlp_fwd_ex(x, R, in_dim, D, out_dim, H, layer_norm, params, y, saved, work, 0, stream);
}

extern "C" int bsms_mlp_fwd_ex(const float* x, int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int H, int layer_norm,
                               const float* const* params, float* y, void* saved, void* work, int flags, bsms_stream_t stream) {
  int rc = check_mlp(R, in_dim, D, out_dim, H, layer_norm, "mlp_fwd");
  if (rc) return rc;
  BSMS_REQUIRE((x && y) || R == 0, BSMS_E_INVALID_ARG, "mlp_fwd: null argument");
  BSMS_REQUIRE(params != nullptr && (saved != nullptr || work != nullptr), BSMS_E_INVALID_ARG, "mlp_fwd: null argument");
  hipStream_t s = as_stream(stream);
  const MlpKind kind = mlp_kind(in_dim, D, out_dim, layer_norm);
  const bool training = saved != nullptr;     // saved == NULL: inference (packs live in `work`)
  MlpSaved sv = training ? carve_mlp_saved(saved, R, D, H, true) : carve_mlp_saved(work, R, D, H, false);

  PackTable t{};
  const int lfirst = (kind == MLP_SMALL_LN) ? 1 : 0;          // first Linear that runs on the MFMA
  const int llast = (kind == MLP_ROWS_SMALL) ? H - 1 : H;     // last one
  if (kind == MLP_SMALL_LN) add_pack(t, params[0], (int)in_dim, 0, 0, (int)D, (int)in_dim, PACK_TRANSPOSE, sv.w0t);
  for (int l = lfirst; l <= llast; ++l) {
    add_pack(t, params[2 * l], (int)D, 0, 0, (int)D, (int)D, PACK_FRAG, sv.w[l], params[2 * l + 1]);
    if (training) add_pack(t, params[2 * l], (int)D, 0, 0, (int)D, (int)D, PACK_FRAG_T, sv.wt[l]);
  }
  t.zero = training ? sv.bound : nullptr;
  BSMS_REQUIRE(!(flags & BSMS_MLP_REUSE_PACKS) || !training, BSMS_E_INVALID_ARG, "mlp_fwd: BSMS_MLP_REUSE_PACKS is an inference flag (saved must be NULL)");
  if (!(flags & BSMS_MLP_REUSE_PACKS) && (rc = launch_prepack(t, s))) return rc;

  ChainFwdArgs a{};
  a.R = R; a.x = x;
  int st = 0;
  if (kind == MLP_SMALL_LN) {
    a.K0 = (int)in_dim; a.w0t = sv.w0t; a.bias_in = params[1]; a.store_in = training ? sv.act[0] : nullptr;
  }
  for (int l = lfirst; l <= llast; ++l, ++st) {
    a.wp[st] = reinterpret_cast<const float4*>(sv.w[l]);
    a.store[st] = (training && l < H) ? sv.act[l] : nullptr;
  }
  a.nstage = st;
  a.y = y;
  if (training && !g_node_bf3) for (int k = 0; k < st; ++k) a.amax[k] = sv.bound + size_t(k) * kBoundWidth;
  if (kind == MLP_ROWS_SMALL) {
    a.wout = params[2 * H]; a.bout = params[2 * H + 1]; a.C = (int)out_dim;
    return launch_chain_fwd((int)D, IN_ROWS, OUT_SMALL, a, s);
  }
  a.yln = training ? sv.yln : nullptr; a.rstd = training ? sv.rstd : nullptr;
  return launch_chain_fwd((int)D, kind == MLP_SMALL_LN ? IN_SMALL : IN_ROWS, OUT_LN, a, s);
}

extern "C" int bsms_mlp_bwd(const float* x, const float* grad_y, int64_t R, int64_t in_dim, int64_t D, int64_t out_dim,
                            int H, int layer_norm, const float* const* params, const void* saved, void* work,
                            float* grad_x, float* const* grads, bsms_stream_t stream) {
  return bsms_mlp_bwd_ex(x, grad_y, R, in_dim, D, out_dim, H, layer_norm, params, saved, work, grad_x, grads, 0, stream);
}

extern "C" int bsms_mlp_bwd_ex(const float* x, const float* grad_y, int64_t R, int64_t in_dim, int64_t D, int64_t out_dim,
                               int H, int layer_norm, const float* const* params, const void* saved, void* work,
                               float* grad_x, float* const* grads, int flags, bsms_stream_t stream) {
  int rc = check_mlp(R, in_dim, D, out_dim, H, layer_norm, "mlp_bwd");
  if (rc) return rc;
  BSMS_REQUIRE(params && grads && saved && work, BSMS_E_INVALID_ARG, "mlp_bwd: null argument");
  BSMS_REQUIRE((x && grad_y) || R == 0, BSMS_E_INVALID_ARG, "mlp_bwd: null tensor");
  hipStream_t s = as_stream(stream);
  const MlpKind kind = mlp_kind(in_dim, D, out_dim, layer_norm);
  BSMS_REQUIRE(!(kind == MLP_SMALL_LN && grad_x), BSMS_E_UNSUPPORTED,
               "mlp_bwd: input gradient of a narrow-input MLP is not implemented (pass grad_x = NULL)");
  BSMS_REQUIRE(kind == MLP_SMALL_LN || grad_x, BSMS_E_INVALID_ARG, "mlp_bwd: grad_x is null");
  MlpSaved sv = carve_mlp_saved(const_cast<void*>(saved), R, D, H);
  MlpWork wk = carve_mlp_work(work, R, D, H);

  // g[l] = gradient w.r.t. the output of Linear_l (pre-activation)
  ChainBwdArgs a{};
  a.R = R; a.dy = grad_y;
  int top;  // Linear index whose output gradient enters the chain
  if (kind == MLP_ROWS_SMALL) {
    a.wout = params[2 * H]; a.C = (int)out_dim; a.mask_in = sv.act[H - 1];
    top = H - 1;
  } else {
    a.yln = sv.yln; a.rstd = sv.rstd;
    top = H;
  }
  const int bottom = (kind == MLP_SMALL_LN) ? 1 : 1;  // dgrad stages run Linear_top .. Linear_1
  a.gstore[0] = wk.g[top];
  int k = 0;
  for (int l = top; l >= bottom; --l, ++k) {
    a.wpt[k] = reinterpret_cast<const float4*>(sv.wt[l]);
    a.mask[k] = sv.act[l - 1];
    a.gstore[k + 1] = wk.g[l - 1];
  }
  a.nstage = k;
  if (!g_node_bf3) for (int q = 0; q <= k; ++q) a.gmax[q] = sv.bound + size_t(16 + q) * kBoundWidth;
  if (kind == MLP_SMALL_LN) {
    rc = launch_chain_bwd((int)D, G_ROWS_LN, F_NONE, a, s);
  } else {
    a.wh0 = reinterpret_cast<const float4*>(sv.wt[0]);
    a.dx = grad_x;
    rc = launch_chain_bwd((int)D, kind == MLP_ROWS_SMALL ? G_SMALL : G_ROWS_LN, F_HEADS1, a, s);
  }
  if (rc) return rc;

  // BSMS_BWD_DEFER_JOIN: grad_x is complete in stream order here; the weight gradients go to side lane 0 and the caller
  // joins later (bsms_side_lanes_join) -- the fused step runs the decoder's weight gradients under the U-Net's first block
  if (flags & BSMS_BWD_DEFER_JOIN) {
    SideLane* lane = nullptr;
    if ((rc = side_lane(&lane, 0, s)) || (rc = side_fork(lane, s))) return rc;
    s = lane->stream;
  }
  WgradJob jobs[kMaxWgradJobs] = {};
  int nj = 0;
  for (int l = (kind == MLP_SMALL_LN ? 1 : 0); l <= top; ++l) {
    WgradJob& j = jobs[nj++];
    j.G = wk.g[l]; j.A = (l == 0) ? x : sv.act[l - 1];
    j.dW = grads[2 * l]; j.db = grads[2 * l + 1];
    j.R = R; j.ldg = j.lda = j.ldw = (int)D; j.col0 = 0;
    j.g_bound = g_node_bf3 ? nullptr : sv.bound + size_t(16 + (top - l)) * kBoundWidth;                          // g[l] = gstore[top - l]
    j.a_bound = g_node_bf3 ? nullptr : sv.bound + size_t(l - (kind == MLP_SMALL_LN ? 1 : 0)) * kBoundWidth;    // the tensor entering the forward stage of Linear l
    j.g_mul = j.a_mul = 1.f;
  }
  if ((rc = launch_wgrad((int)D, jobs, nj, wk.wg, s))) return rc;

  SmallWgradArgs sa{};
  sa.R = R; sa.D = (int)D;
  if (kind == MLP_SMALL_LN) {          // dW0[f][k] = sum_r g0[r][f] x[r][k] ; db0 = colsum g0
    sa.G = wk.g[0]; sa.S = x; sa.S_cols = (int)in_dim;
    sa.out = grads[0]; sa.os = 1; sa.of = in_dim; sa.colsum = grads[1];
    rc = launch_small_wgrad(sa, wk.sw, s);
  } else if (kind == MLP_ROWS_SMALL) {        // dW_H[c][f] = sum_r dy[r][c] a_{H-1}[r][f] ; db_H = colsum dy
    sa.G = sv.act[H - 1]; sa.S = grad_y; sa.S_cols = (int)out_dim;
    sa.out = grads[2 * H]; sa.os = D; sa.of = 1; sa.colsum_S = grads[2 * H + 1];
    rc = launch_small_wgrad(sa, wk.sw, s);
  }
  if (rc) return rc;
  if (flags & BSMS_BWD_DEFER_JOIN) {   // visible to bsms_side_lanes_join even if nothing else uses the lane afterwards
    SideLane* lane = nullptr;
    if ((rc = side_lane(&lane, 0, s)) || (rc = side_mark(lane, 0))) return rc;
  }
  return BSMS_OK;
}
