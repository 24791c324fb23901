// Shared host-side helpers for libbsms_hip.so (error reporting, plan struct, launch geometry).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/bsms_hip.h"

namespace bsms {

void set_error(const char* fmt, ...);

#define BSMS_FAIL(code, ...)        \
  do {                              \
    ::bsms::set_error(__VA_ARGS__); \
    return (code);                  \
  } while (0)

#define BSMS_REQUIRE(cond, code, ...)         \
  do {                                        \
    if (!(cond)) BSMS_FAIL(code, __VA_ARGS__); \
  } while (0)

#define BSMS_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) BSMS_FAIL(BSMS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

#define BSMS_LAUNCH_CHECK()                                                                  \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess) BSMS_FAIL(BSMS_E_HIP, "kernel launch: %s", hipGetErrorString(e_)); \
  } while (0)

inline hipStream_t as_stream(bsms_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Internal fork/join helper: one non-blocking side stream + two events per device, created lazily and kept for
// the life of the process.  A fork makes the side stream wait for everything queued on `main` so far; a join makes
// `main` wait for the side stream.  Stream-wait-event takes a snapshot, so the events are safely reused per call,
// and both operations are legal inside HIP-graph capture.
struct SideLane {
  hipStream_t stream = nullptr;
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  hipEvent_t done_ev[2] = {nullptr, nullptr};   // "everything queued on the lane up to here has run", two slots
};
constexpr int kSideLanes = 2;
int side_lane(SideLane** out, int which, hipStream_t caller);   // for the current device; `which` < kSideLanes; `caller`: see plan.hip (capture)
int side_fork(SideLane* lane, hipStream_t main);     // side waits for main
int side_join(SideLane* lane, hipStream_t main);     // main waits for side
// Deferred join: side_mark records "the lane's work so far" into slot 0/1; side_wait_mark makes `main` wait for what
// that slot recorded last (callers only wait on slots they marked themselves: a stale record from an earlier call
// would be an event from outside an ongoing graph capture).  The U-Net backward lets the weight gradients of block k run
// under the gradient chain of block k+1 and only waits for them before block k+2 reuses their scratch (bsgmp.hip).
int side_mark(SideLane* lane, int slot);
int side_wait_mark(SideLane* lane, int slot, hipStream_t main);
int side_mark_chain(SideLane* a, SideLane* b, int slot);   // b's mark covers a's: wait for b only
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Per-DEVICE launch state: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the device that is current
// when it is called, and the CU count is a property of that device -- a process driving several devices (not the one-process-per-GPU
// layout of dp.py, but legal at the C ABI) must not reuse the first device's answers.  Both are cached per device index; the
// races are benign (two threads may set the same attribute / read the same count twice).
constexpr int kMaxDevices = 64;
inline int current_device_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  return dev >= 0 && dev < kMaxDevices ? dev : 0;
}
struct DynLdsAttr {
  int state[kMaxDevices] = {};   // 0: not asked on this device yet; 1 + hipError_t afterwards
  hipError_t ensure(const void* fn, int bytes) {
    int& st = state[current_device_index()];
    if (st == 0) st = 1 + int(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return hipError_t(st - 1);
  }
};
inline int device_cu_count() {
  static int cus[kMaxDevices] = {};
  const int dev = current_device_index();
  if (cus[dev] <= 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
    cus[dev] = v;
  }
  return cus[dev];
}

// ---- internal entry points shared between translation units (not part of the C ABI)
// rowsum.hip: bsms_edge_conv + fused addend
int edge_conv_add(const bsms_plan* p, const float* x, int64_t B, int64_t D, const float* ew, int aggregating, int pooled,
                  float* out, const float* addend, hipStream_t stream);
// gmp.hip: the GMP block with the knobs the U-Net entry uses.  `packs_base` (nullable): where the weight packs of an
// INFERENCE call live (training keeps them in `saved`); `do_prepack` = false: the packs were filled by gmp_prepack
// earlier in the same step; `resid2` (nullable): the skip connection added to the block's output (ops/BSMS.py:102).
size_t gmp_pack_bytes(int64_t D, int hidden);
int gmp_prepack(int64_t B, int64_t N, int64_t E, int64_t D, int64_t p, int hidden, const float* const* params,
                void* saved, void* work, void* packs_base, hipStream_t stream, int precision = BSMS_F32);
size_t gmp_saved_bytes_p(int64_t B, int64_t N, int64_t E, int64_t D, int hidden, int precision);
int gmp_fwd_core(const bsms_plan* plan, const float* x, const float* pos, int64_t B, int64_t D, int64_t p,
                 int64_t pos_bstride, int hidden, const float* const* params, float* out, void* saved, void* work,
                 void* packs_base, bool do_prepack, const float* resid2, hipStream_t stream, int precision = BSMS_F32);
// `defer_slot` < 0: the side lanes are joined before returning (ABI semantics).  0/1: they are only MARKED in that slot;
// the caller joins later with side_wait_mark on both lanes and must not touch `work` or read `grads` before that.
bool gmp_marks_chained();   // gmp.hip: waiting for lane 1's mark of a slot is enough
int gmp_bwd_core(const bsms_plan* plan, const float* x, const float* pos, const float* grad_out, int64_t B, int64_t D,
                 int64_t p, int64_t pos_bstride, int hidden, const float* const* params, const void* saved, void* work,
                 float* grad_x, float* const* grads, int defer_slot, hipStream_t stream, int precision = BSMS_F32);
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// bsms_plan_concat (plan.hip: host side; rowsum.hip: the kernel -- plan.hip is also compiled as plain C++ for the sanitizer build):
// the index blocks of up to kCatParts plans copied into the blocks of their block-diagonal union, node / edge / pooled offsets added
constexpr size_t kIdxAlign = 64;   // int32 elements (256 bytes) between the arrays of a plan's index block
#ifdef __HIPCC__
#define BSMS_HD __host__ __device__
#else
#define BSMS_HD
#endif
BSMS_HD inline size_t idx_pad(size_t n) { return (n + kIdxAlign - 1) / kIdxAlign * kIdxAlign; }
constexpr int kCatParts = 16;
struct CatPart {
  const int32_t *blk, *pool;               // the part's two device blocks (pool: null when no part has one)
  int32_t N, E, Nk, Ek, Ep;                // its sizes ...
  int32_t n_off, e_off, k_off, ek_off, ep_off;   // ... and where it starts in the union
};
struct CatArgs {
  CatPart part[kCatParts];
  int nparts;
  int32_t *out_blk, *out_pool;             // blocks of the union, laid out for (N, E, Nk, Ek, Ep) below
  int32_t N, E, Nk, Ek, Ep;
  int has_w;                               // copy the gathered edge weights k_w / p_w too (every part is bound)
  int64_t* coo_out;                        // nullable: [2, E] int64 edge list of the union in the CALLER's edge order
  int64_t* ids_out;                        // nullable: [Nk] int64 kept ids of the union
};
int launch_plan_concat(const CatArgs& a, hipStream_t s) __attribute__((weak));   // rowsum.hip (absent from the host-only sanitizer library)

}  // namespace bsms

// One mesh level in HBM.  All arrays int32, owned by the plan.
//   dst-sorted CSR ("plan order" q = 0..E-1, stable w.r.t. the caller's edge order):
//     rowptr[N+1]; src[q], dst[q] node ids; perm[q] = caller's edge id of plan slot q
//   src-sorted transpose (slot t = 0..E-1, stable):
//     t_rowptr[N+1]; t_dst[t] target node; t_eid[t] caller's edge id; t_pos[t] plan slot q
//   pooling (optional): ids[Nk] kept fine ids (ascending), inv[N] fine -> coarse or -1
struct bsms_plan {
  int64_t N = 0, E = 0, Nk = 0;
  int64_t min_out_degree = 0, max_source = -1;
  int64_t max_in_degree = 0, max_out_degree = 0;   // bound the scatter sums of a bounded tensor (wgrad.hip: operand scales)
  int32_t *rowptr = nullptr, *src = nullptr, *dst = nullptr, *perm = nullptr;
  int32_t *t_rowptr = nullptr, *t_dst = nullptr, *t_eid = nullptr, *t_pos = nullptr;
  int32_t *ids = nullptr, *inv = nullptr;
  // pooled transitions, compacted (built by bsms_plan_set_pool; plan.hip).  restrict: CSR over the KEPT rows only,
  //   k_rowptr[Nk+1]; k_src[q] fine source node; k_eid[q] caller's edge id -- the slots of kept row ids[k] in plan order.
  // prolong: CSR over ALL fine rows by source with the slots whose target is not kept dropped (they add nothing),
  //   p_rowptr[N+1]; p_src[t] COARSE row of the target (inv[t_dst]); p_eid[t] caller's edge id.
  // k_w / p_w: the edge weights in those slot orders, filled by bsms_plan_bind_edge_weights (w_bound = the `ew` they were
  // gathered from): a transition kernel then reads index + weight as two coalesced streams instead of chasing
  // rows -> rowptr -> (xidx, widx) -> (xmap, w) through four dependent round trips.
  int32_t *k_rowptr = nullptr, *k_src = nullptr, *k_eid = nullptr, *p_rowptr = nullptr, *p_src = nullptr, *p_eid = nullptr;
  float *k_w = nullptr, *p_w = nullptr;
  int64_t Ek = 0, Ep = 0;
  const float* w_bound = nullptr;
  std::vector<int32_t> host;                         // host copy of the index block (set_pool derives the compact lists from it); released by the first
                                                     // bsms_plan_set_pool -- a later re-pool reads the block back from the device (host_words int32)
  size_t host_words = 0;
  int32_t *block = nullptr, *pool_block = nullptr;   // the two device allocations the pointers above point into
  size_t block_cap = 0, pool_cap = 0;                // their capacities in bytes (plan.hip recycles them)
  int device = 0;                                    // the device the blocks live on (current device at bsms_plan_create)
};
