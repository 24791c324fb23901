// Graph plan: COO [2,E] int64 (host) -> dst-sorted CSR + src-sorted transpose in HBM (int32).
// Replaces the per-call index broadcasting of the reference (utils/basic.py:312-343) and the
// advanced-indexing gathers x[:, i], x[:, j] (ops/basic.py:70,130).  Built once per mesh level.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>

#include <cstdlib>

#include "common.h"

namespace bsms {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace bsms

namespace bsms {
static std::mutex g_lane_mu;
static SideLane g_lanes[64][kSideLanes];

// Can work queued on `b` overtake work queued on `a`?  HIP maps its streams onto a few hardware queues (4 by default), by the
// reference counts at creation time; two streams on ONE queue run in order.  Found in round 5: in a process that had built eight
// meshes' plans first (their upload stream, torch's copy-stream pool) both side lanes landed on one hardware queue and the
// block-diagonal cylinder step ran 6-20 % slower than in a process where the lanes happened to be created earlier
// (profiles/r05_hw_queues.txt).  The probe: ~250 us of fills on `a`, one small fill + event on `b`; if that event completes while
// `a` is still busy the queues are distinct.  Runtime API only (this file is also built as plain C++ for the sanitizer tests).
static bool can_overtake(hipStream_t a, hipStream_t b) {
  constexpr size_t kBytes = size_t(64) << 20;
  char* buf = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), kBytes + 256) != hipSuccess) { (void)hipGetLastError(); return true; }   // cannot tell
  hipEvent_t ea = nullptr, eb = nullptr;
  bool distinct = true;
  if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&eb, hipEventDisableTiming) == hipSuccess) {
    bool ok = true;
    for (int i = 0; i < 16 && ok; ++i) ok = hipMemsetAsync(buf, 0, kBytes, a) == hipSuccess;
    ok = ok && hipEventRecord(ea, a) == hipSuccess && hipMemsetAsync(buf + kBytes, 0, 256, b) == hipSuccess && hipEventRecord(eb, b) == hipSuccess &&
         hipEventSynchronize(eb) == hipSuccess;
    if (ok) distinct = hipEventQuery(ea) == hipErrorNotReady;
    (void)hipEventSynchronize(ea);
    (void)hipGetLastError();
  }
  if (ea) (void)hipEventDestroy(ea);
  if (eb) (void)hipEventDestroy(eb);
  (void)hipFree(buf);
  return distinct;
}

// a non-blocking stream that shares its hardware queue neither with the legacy default stream (what PyTorch runs on) nor with the
// lanes that exist already; streams that failed the probe are kept until one passes (each raises the reference count of the queue
// it sits on, so the next creation goes elsewhere), then destroyed
static int create_lane_stream(hipStream_t* out, const SideLane* lanes, int which, bool probe) {
  if (!probe) {   // the caller is capturing a HIP graph: hipMalloc / legacy-stream fills / event synchronisation would invalidate the capture
    BSMS_HIP_CHECK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return BSMS_OK;
  }
  std::vector<hipStream_t> ballast;
  hipStream_t st = nullptr;
  for (int attempt = 0; attempt < 6; ++attempt) {
    BSMS_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));   // (a lower or higher stream priority for the lanes: +-0, profiles/README.md)
    // twice against the caller's default stream: one kernel trace of round 6 (profiles/r06_step_summary.md: a profiled run) shows a
    // lane that had passed ONE probe on the default stream's hardware queue after all -- its 38 launches per step then run in order
    // with the caller's, +0.7 ms per step; a second pass costs ~0.3 ms once per process
    bool good = can_overtake(nullptr, st) && can_overtake(nullptr, st);
    for (int k = 0; k < kSideLanes && good; ++k)
      if (k != which && lanes[k].stream) good = can_overtake(lanes[k].stream, st);
    if (good || attempt == 5) break;   // after six tries: keep the last one (four hardware queues cannot separate every stream of a process)
    ballast.push_back(st);
    st = nullptr;
  }
  for (hipStream_t b : ballast) (void)hipStreamDestroy(b);
  *out = st;
  return BSMS_OK;
}

// `caller`: the stream the entry point was called with.  If it is being CAPTURED the lane is created without the hardware-queue
// probe (the probe allocates, fills on the legacy stream and synchronises an event -- none of it legal under a
// capture); such a lane may share a hardware queue with another stream, which costs overlap, never correctness.  Callers that
// capture should run one eager call first (the Python paths do: their warm-up steps), as include/bsms_hip.h says.
int side_lane(SideLane** out, int which, hipStream_t caller) {
  int dev = 0;
  BSMS_HIP_CHECK(hipGetDevice(&dev));
  BSMS_REQUIRE(dev >= 0 && dev < 64 && which >= 0 && which < kSideLanes, BSMS_E_UNSUPPORTED, "side_lane: device %d lane %d", dev, which);
  std::lock_guard<std::mutex> lock(g_lane_mu);
  SideLane& l = g_lanes[dev][which];
  if (!l.stream) {
#ifdef BSMS_EXPERIMENTS
    if (const char* e = getenv("BSMS_LANE_PRIO")) {   // A/B: lanes below / above the caller's stream (hipDeviceGetStreamPriorityRange: least .. greatest)
      int least = 0, greatest = 0;
      BSMS_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
      BSMS_HIP_CHECK(hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, atoi(e) < 0 ? least : greatest));
    } else
#endif
    {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(caller, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }   // cannot tell: do not risk a capture
      int rc_ = create_lane_stream(&l.stream, g_lanes[dev], which, cap == hipStreamCaptureStatusNone);
      if (rc_) return rc_;
    }
    // same-device stream ordering only: no timing, and no system-scope fence when an event completes (the kernels'
    // own end-of-kernel release / start-of-kernel acquire make their data visible device-wide; the extra fence is for
    // hosts and other devices reading behind the event, which nothing here does)
    unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
#ifdef BSMS_EXPERIMENTS
    if (getenv("BSMS_EVENT_SYSTEM_FENCE")) flags = hipEventDisableTiming;   // A/B: 179.7 (with the fence) vs 182.0 steps/s
#endif
    BSMS_HIP_CHECK(hipEventCreateWithFlags(&l.fork_ev, flags));
    BSMS_HIP_CHECK(hipEventCreateWithFlags(&l.join_ev, flags));
    for (int k = 0; k < 2; ++k) BSMS_HIP_CHECK(hipEventCreateWithFlags(&l.done_ev[k], flags));
  }
  *out = &l;
  return BSMS_OK;
}
// record + wait are one critical section: the lanes (and their events) are shared by every caller stream / thread of
// the device, and a wait must see ITS record, not one a concurrent caller slipped in between
int side_fork(SideLane* lane, hipStream_t main) {
  std::lock_guard<std::mutex> lock(g_lane_mu);
  BSMS_HIP_CHECK(hipEventRecord(lane->fork_ev, main));
  BSMS_HIP_CHECK(hipStreamWaitEvent(lane->stream, lane->fork_ev, 0));
  return BSMS_OK;
}
int side_join(SideLane* lane, hipStream_t main) {
  std::lock_guard<std::mutex> lock(g_lane_mu);
  BSMS_HIP_CHECK(hipEventRecord(lane->join_ev, lane->stream));
  BSMS_HIP_CHECK(hipStreamWaitEvent(main, lane->join_ev, 0));
  return BSMS_OK;
}
int side_mark(SideLane* lane, int slot) {
  std::lock_guard<std::mutex> lock(g_lane_mu);
  BSMS_HIP_CHECK(hipEventRecord(lane->done_ev[slot & 1], lane->stream));
  return BSMS_OK;
}
// both lanes' work so far as ONE event: lane `b` waits for lane `a`'s mark and records its own; a later
// side_wait_mark(b, slot, main) then covers both (one barrier packet on the caller's stream instead of two)
int side_mark_chain(SideLane* a, SideLane* b, int slot) {
  std::lock_guard<std::mutex> lock(g_lane_mu);
  BSMS_HIP_CHECK(hipEventRecord(a->done_ev[slot & 1], a->stream));
  BSMS_HIP_CHECK(hipStreamWaitEvent(b->stream, a->done_ev[slot & 1], 0));
  BSMS_HIP_CHECK(hipEventRecord(b->done_ev[slot & 1], b->stream));
  return BSMS_OK;
}
int side_wait_mark(SideLane* lane, int slot, hipStream_t main) {
  std::lock_guard<std::mutex> lock(g_lane_mu);
  BSMS_HIP_CHECK(hipStreamWaitEvent(main, lane->done_ev[slot & 1], 0));
  return BSMS_OK;
}
}  // namespace bsms

using namespace bsms;

// 1: work queued on `b` can overtake work queued on `a` (the two streams sit on different hardware queues); 0: they run in order.
extern "C" int bsms_streams_overlap(bsms_stream_t a, bsms_stream_t b) {
  hipStream_t sa = as_stream(a), sb = as_stream(b);
  if (sa == sb) return 0;
  hipStreamCaptureStatus ca = hipStreamCaptureStatusNone, cb = hipStreamCaptureStatusNone;
  BSMS_REQUIRE(hipStreamIsCapturing(sa, &ca) == hipSuccess && hipStreamIsCapturing(sb, &cb) == hipSuccess &&
               ca == hipStreamCaptureStatusNone && cb == hipStreamCaptureStatusNone, BSMS_E_UNSUPPORTED,
               "streams_overlap: a stream is capturing (the probe allocates and synchronises)");
  return can_overtake(sa, sb) ? 1 : 0;
}

extern "C" int bsms_abi_version(void) { return 3; }  // 2: saved == NULL selects inference in *_fwd; 3: bsms_mlp_fwd_ex, larger saved buffers (bound slots)
extern "C" const char* bsms_last_error(void) { return bsms::g_err; }

namespace {
// Index uploads go through a private non-blocking stream: a plain hipMemcpy is ordered behind everything the caller has
// queued on the NULL stream (PyTorch's default), i.e. building the plans of a NEW mesh would drain the GPU first.  The
// copy is complete on the host's timeline when upload_block returns, so kernels launched afterwards on any stream see it.
constexpr int kMaxDevices = 64;
int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}
std::mutex g_stream_mu;
hipStream_t upload_stream() {   // one per device (a process normally drives one GPU; PyTorch callers may drive several)
  static hipStream_t streams[kMaxDevices] = {};
  static bool tried[kMaxDevices] = {};
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_stream_mu);
  if (!tried[dev]) {
    tried[dev] = true;
    if (hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) streams[dev] = nullptr;
  }
  return streams[dev];
}
// Device blocks of destroyed plans are kept for the next plan: hipFree waits for the whole device, and a variable-mesh
// training run retires L + 1 plans per step once the host-side cache is full.  (bsms_plan_destroy's contract: nothing that
// uses the plan is still in flight -- the recycled block is overwritten by the next upload without further ordering.)
struct PoolBlock { size_t cap; int32_t* ptr; int dev; };
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t kPoolMaxBytes = size_t(512) << 20, kPoolMaxBlocks = 512, kBlockGrain = size_t(64) << 10;

int alloc_block(int32_t** dev, size_t bytes, size_t* cap) {
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    int best = -1;
    const int cur = current_device();
    for (int i = 0; i < (int)g_pool.size(); ++i)
      if (g_pool[i].dev == cur && g_pool[i].cap >= bytes && g_pool[i].cap <= 2 * bytes + kBlockGrain &&
          (best < 0 || g_pool[i].cap < g_pool[best].cap)) best = i;
    if (best >= 0) {
      *dev = g_pool[best].ptr;
      *cap = g_pool[best].cap;
      g_pool_bytes -= g_pool[best].cap;
      g_pool.erase(g_pool.begin() + best);
      return BSMS_OK;
    }
  }
  *cap = (bytes + bytes / 8 + kBlockGrain - 1) / kBlockGrain * kBlockGrain;   // headroom: the next batch's level is rarely the same size
  BSMS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dev), *cap));
  return BSMS_OK;
}
void release_block(int32_t* ptr, size_t cap, int dev) {
  if (!ptr) return;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (g_pool.size() < kPoolMaxBlocks && g_pool_bytes + cap <= kPoolMaxBytes) {
      g_pool.push_back(PoolBlock{cap, ptr, dev});
      g_pool_bytes += cap;
      return;
    }
  }
  (void)hipFree(ptr);
}

// ONE allocation and ONE copy per plan (ten of each cost ~1 ms per level; a variable-mesh batch builds L + 1 plans per step)
int upload_block(int32_t** dev, size_t* cap, const std::vector<int32_t>& host) {
  const size_t bytes = std::max<size_t>(host.size(), 1) * sizeof(int32_t);
  int rc = alloc_block(dev, bytes, cap);
  if (rc) return rc;
  if (!host.empty()) {
    hipStream_t us = upload_stream();
    if (us) {
      BSMS_HIP_CHECK(hipMemcpyAsync(*dev, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice, us));
      BSMS_HIP_CHECK(hipStreamSynchronize(us));
    } else {
      BSMS_HIP_CHECK(hipMemcpy(*dev, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
  }
  return BSMS_OK;
}
}  // namespace

extern "C" int bsms_plan_create(const int64_t* coo, int64_t E, int64_t N, bsms_plan_t** out) {
  BSMS_REQUIRE(out != nullptr, BSMS_E_INVALID_ARG, "plan_create: out is null");
  *out = nullptr;
  BSMS_REQUIRE(E >= 0 && N >= 0, BSMS_E_SHAPE, "plan_create: negative size");
  BSMS_REQUIRE(E < (int64_t(1) << 31) && N < (int64_t(1) << 31), BSMS_E_UNSUPPORTED,
               "plan_create: indices must fit int32 (E=%lld N=%lld)", (long long)E, (long long)N);
  BSMS_REQUIRE(coo != nullptr || E == 0, BSMS_E_INVALID_ARG, "plan_create: coo is null");
  const int64_t* gi = coo;
  const int64_t* gj = coo + E;
  for (int64_t e = 0; e < E; ++e)
    BSMS_REQUIRE(gi[e] >= 0 && gi[e] < N && gj[e] >= 0 && gj[e] < N, BSMS_E_INVALID_ARG,
                 "plan_create: edge %lld = (%lld -> %lld) out of range for N=%lld", (long long)e,
                 (long long)gi[e], (long long)gj[e], (long long)N);

  // all eight index arrays in one host block (layout = the device block)
  const size_t nN = idx_pad(size_t(N) + 1), nE = idx_pad(size_t(E));
  std::vector<int32_t> blk(2 * nN + 6 * nE, 0);
  int32_t *rowptr = blk.data(), *t_rowptr = rowptr + nN, *src = t_rowptr + nN, *dst = src + nE, *perm = dst + nE,
          *t_dst = perm + nE, *t_eid = t_dst + nE, *t_pos = t_eid + nE;
  for (int64_t e = 0; e < E; ++e) {
    rowptr[gj[e] + 1]++;
    t_rowptr[gi[e] + 1]++;
  }
  int64_t min_deg = E > 0 || N > 0 ? INT64_MAX : 0, max_src = -1, max_in = 0, max_out = 0;
  for (int64_t n = 0; n < N; ++n) {
    min_deg = std::min<int64_t>(min_deg, t_rowptr[n + 1]);
    max_out = std::max<int64_t>(max_out, t_rowptr[n + 1]);
    max_in = std::max<int64_t>(max_in, rowptr[n + 1]);
  }
  if (N == 0) min_deg = 0;
  for (int64_t e = 0; e < E; ++e) max_src = std::max(max_src, gi[e]);
  std::partial_sum(rowptr, rowptr + N + 1, rowptr);
  std::partial_sum(t_rowptr, t_rowptr + N + 1, t_rowptr);

  // stable counting sorts: slots of equal key keep the caller's edge order
  std::vector<int32_t> slot_of_edge(E);
  {
    std::vector<int32_t> cur(rowptr, rowptr + N);
    for (int64_t e = 0; e < E; ++e) {
      int32_t q = cur[gj[e]]++;
      src[q] = (int32_t)gi[e];
      dst[q] = (int32_t)gj[e];
      perm[q] = (int32_t)e;
      slot_of_edge[e] = q;
    }
  }
  {
    std::vector<int32_t> cur(t_rowptr, t_rowptr + N);
    for (int64_t e = 0; e < E; ++e) {
      int32_t t = cur[gi[e]]++;
      t_dst[t] = (int32_t)gj[e];
      t_eid[t] = (int32_t)e;
      t_pos[t] = slot_of_edge[e];
    }
  }

  bsms_plan* p = new bsms_plan();
  p->device = current_device();
  p->N = N;
  p->E = E;
  p->min_out_degree = min_deg;
  p->max_source = max_src;
  p->max_in_degree = max_in;
  p->max_out_degree = max_out;
  int rc = upload_block(&p->block, &p->block_cap, blk);
  if (rc) {
    bsms_plan_destroy(p);
    return rc;
  }
  p->host.swap(blk);   // kept until the first bsms_plan_set_pool, which builds the compact transition lists from it
  p->host_words = p->host.size();
  p->rowptr = p->block;
  p->t_rowptr = p->rowptr + nN;
  p->src = p->t_rowptr + nN;
  p->dst = p->src + nE;
  p->perm = p->dst + nE;
  p->t_dst = p->perm + nE;
  p->t_eid = p->t_dst + nE;
  p->t_pos = p->t_eid + nE;
  *out = p;
  return BSMS_OK;
}

extern "C" int bsms_plan_set_pool(bsms_plan_t* p, const int64_t* ids, int64_t Nk) {
  BSMS_REQUIRE(p != nullptr, BSMS_E_INVALID_ARG, "plan_set_pool: plan is null");
  BSMS_REQUIRE(Nk >= 0 && Nk <= p->N && (ids != nullptr || Nk == 0), BSMS_E_SHAPE,
               "plan_set_pool: bad Nk=%lld for N=%lld", (long long)Nk, (long long)p->N);
  const size_t nK = idx_pad(size_t(Nk)), nNinv = idx_pad(size_t(p->N));
  std::vector<int32_t> head(nK + nNinv, -1);
  int32_t *h_ids = head.data(), *h_inv = h_ids + nK;
  for (int64_t k = 0; k < Nk; ++k) {
    BSMS_REQUIRE(ids[k] >= 0 && ids[k] < p->N, BSMS_E_INVALID_ARG, "plan_set_pool: id %lld out of range",
                 (long long)ids[k]);
    BSMS_REQUIRE(h_inv[ids[k]] < 0, BSMS_E_INVALID_ARG, "plan_set_pool: duplicate id %lld", (long long)ids[k]);
    h_ids[k] = (int32_t)ids[k];
    h_inv[ids[k]] = (int32_t)k;
  }
  // compact lists of the two pooled transitions (common.h), from the host copy of the CSR
  const size_t nN = idx_pad(size_t(p->N) + 1), nE = idx_pad(size_t(p->E));
  // the host copy of the CSR: kept since bsms_plan_create for the FIRST pooling only (a cache of 1024 plans used to carry ~2 MB each for
  // good: ADVICE round 4); a plan that is pooled again reads its index block back from the device
  std::vector<int32_t> readback;
  if (p->host.empty()) {
    BSMS_REQUIRE(p->block != nullptr && p->host_words > 0, BSMS_E_INVALID_ARG, "plan_set_pool: plan has no index block");
    readback.resize(p->host_words);
    int dev = 0;
    BSMS_HIP_CHECK(hipGetDevice(&dev));
    if (dev != p->device) BSMS_HIP_CHECK(hipSetDevice(p->device));
    const hipError_t e = hipMemcpy(readback.data(), p->block, p->host_words * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (dev != p->device) (void)hipSetDevice(dev);
    BSMS_REQUIRE(e == hipSuccess, BSMS_E_HIP, "plan_set_pool: reading the index block back: %s", hipGetErrorString(e));
  }
  const int32_t* const hostp = p->host.empty() ? readback.data() : p->host.data();
  const int32_t *rowptr = hostp, *t_rowptr = rowptr + nN, *src = t_rowptr + nN, *perm = src + 2 * nE,
                *t_dst = perm + nE, *t_eid = t_dst + nE;
  int64_t Ek = 0, Ep = 0;
  for (int64_t k = 0; k < Nk; ++k) Ek += rowptr[h_ids[k] + 1] - rowptr[h_ids[k]];
  for (int64_t t = 0; t < p->E; ++t) Ep += h_inv[t_dst[t]] >= 0;
  const size_t nK1 = idx_pad(size_t(Nk) + 1), nEk = idx_pad(size_t(Ek)), nEp = idx_pad(size_t(Ep));
  std::vector<int32_t> blk(nK + nNinv + nK1 + 2 * nEk + nN + 2 * nEp + nEk + nEp, 0);
  std::copy(head.begin(), head.end(), blk.begin());
  int32_t *k_rowptr = blk.data() + nK + nNinv, *k_src = k_rowptr + nK1, *k_eid = k_src + nEk, *p_rowptr = k_eid + nEk,
          *p_src = p_rowptr + nN, *p_eid = p_src + nEp;
  {
    int64_t q = 0;
    for (int64_t k = 0; k < Nk; ++k) {
      k_rowptr[k] = (int32_t)q;
      for (int32_t s = rowptr[h_ids[k]]; s < rowptr[h_ids[k] + 1]; ++s, ++q) { k_src[q] = src[s]; k_eid[q] = perm[s]; }
    }
    k_rowptr[Nk] = (int32_t)q;
    int64_t u = 0;
    for (int64_t i = 0; i < p->N; ++i) {
      p_rowptr[i] = (int32_t)u;
      for (int32_t t = t_rowptr[i]; t < t_rowptr[i + 1]; ++t)
        if (h_inv[t_dst[t]] >= 0) { p_src[u] = h_inv[t_dst[t]]; p_eid[u] = t_eid[t]; ++u; }
    }
    p_rowptr[p->N] = (int32_t)u;
  }
  release_block(p->pool_block, p->pool_cap, p->device);
  p->pool_block = p->ids = p->inv = nullptr;
  p->k_rowptr = p->k_src = p->k_eid = p->p_rowptr = p->p_src = p->p_eid = nullptr;
  p->k_w = p->p_w = nullptr;
  p->w_bound = nullptr;
  p->Nk = p->Ek = p->Ep = 0;
  int rc;
  if ((rc = upload_block(&p->pool_block, &p->pool_cap, blk))) return rc;
  p->ids = p->pool_block;
  p->inv = p->pool_block + nK;
  p->k_rowptr = p->inv + nNinv;
  p->k_src = p->k_rowptr + nK1;
  p->k_eid = p->k_src + nEk;
  p->p_rowptr = p->k_eid + nEk;
  p->p_src = p->p_rowptr + nN;
  p->p_eid = p->p_src + nEp;
  p->k_w = reinterpret_cast<float*>(p->p_eid + nEp);
  p->p_w = p->k_w + nEk;
  p->Nk = Nk;
  p->Ek = Ek;
  p->Ep = Ep;
  if (!p->host.empty()) {   // the compact lists are built: the host copy has served its purpose
    p->host_words = p->host.size();
    std::vector<int32_t>().swap(p->host);
  }
  return BSMS_OK;
}

// The pointers of a plan into its two blocks, from its sizes (the layouts bsms_plan_create / bsms_plan_set_pool write)
static void point_into_blocks(bsms_plan* p) {
  const size_t nN = idx_pad(size_t(p->N) + 1), nE = idx_pad(size_t(p->E));
  p->rowptr = p->block;
  p->t_rowptr = p->rowptr + nN;
  p->src = p->t_rowptr + nN;
  p->dst = p->src + nE;
  p->perm = p->dst + nE;
  p->t_dst = p->perm + nE;
  p->t_eid = p->t_dst + nE;
  p->t_pos = p->t_eid + nE;
  if (!p->pool_block) return;
  const size_t nK = idx_pad(size_t(p->Nk)), nNinv = idx_pad(size_t(p->N)), nK1 = idx_pad(size_t(p->Nk) + 1),
               nEk = idx_pad(size_t(p->Ek)), nEp = idx_pad(size_t(p->Ep));
  p->ids = p->pool_block;
  p->inv = p->pool_block + nK;
  p->k_rowptr = p->inv + nNinv;
  p->k_src = p->k_rowptr + nK1;
  p->k_eid = p->k_src + nEk;
  p->p_rowptr = p->k_eid + nEk;
  p->p_src = p->p_rowptr + nN;
  p->p_eid = p->p_src + nEp;
  p->k_w = reinterpret_cast<float*>(p->p_eid + nEp);
  p->p_w = p->k_w + nEk;
}

// Block-diagonal union of plans ON THE DEVICE (round 6).  The stable dst-sorted CSR, its transpose and
// the compact transition lists of an offset-concatenated edge list ARE the concatenations of the parts' arrays with node / edge /
// pooled-row offsets added (the parts' target ranges are disjoint and ordered), so a batch of different meshes -- the reference's
// cylinder_flow path, datasets/base.py:319-351 + PyG Batch, models/model.py:194-200 -- needs no host CSR build and no upload:
// one kernel per 16 parts.  Result == bsms_plan_create on the concatenated COO + bsms_plan_set_pool on the offset ids
// (+ bsms_plan_bind_edge_weights when `ew_cat` is given), array for array (tests/test_hip_host_builder.py).
extern "C" int bsms_plan_concat(const bsms_plan_t* const* parts, int nparts, const float* ew_cat, int64_t* coo_out, int64_t* ids_out,
                                bsms_stream_t stream, bsms_plan_t** out) {
  BSMS_REQUIRE(out != nullptr, BSMS_E_INVALID_ARG, "plan_concat: out is null");
  *out = nullptr;
  BSMS_REQUIRE(parts != nullptr && nparts >= 1, BSMS_E_INVALID_ARG, "plan_concat: no parts");
  BSMS_REQUIRE(launch_plan_concat != nullptr, BSMS_E_UNSUPPORTED, "plan_concat: this build has no device code");
  const int dev = current_device();
  int64_t N = 0, E = 0, Nk = 0, Ek = 0, Ep = 0;
  bool pooled = parts[0] && parts[0]->ids != nullptr, bound = ew_cat != nullptr;
  for (int b = 0; b < nparts; ++b) {
    const bsms_plan* q = parts[b];
    BSMS_REQUIRE(q != nullptr && q->block != nullptr, BSMS_E_INVALID_ARG, "plan_concat: part %d is null", b);
    BSMS_REQUIRE(q->device == dev, BSMS_E_INVALID_ARG, "plan_concat: part %d lives on device %d, the current device is %d", b, q->device, dev);
    BSMS_REQUIRE((q->ids != nullptr) == pooled, BSMS_E_INVALID_ARG, "plan_concat: part %d %s a pool, part 0 %s", b,
                 q->ids ? "has" : "has no", pooled ? "has one" : "has none");
    BSMS_REQUIRE(!bound || q->w_bound != nullptr, BSMS_E_INVALID_ARG, "plan_concat: ew_cat given but part %d has no bound edge weights", b);
    N += q->N; E += q->E; Nk += q->Nk; Ek += q->Ek; Ep += q->Ep;
  }
  BSMS_REQUIRE(!bound || pooled, BSMS_E_INVALID_ARG, "plan_concat: ew_cat given but the parts have no pools");
  BSMS_REQUIRE(E < (int64_t(1) << 31) && N < (int64_t(1) << 31), BSMS_E_UNSUPPORTED, "plan_concat: indices must fit int32 (E=%lld N=%lld)",
               (long long)E, (long long)N);
  bsms_plan* p = new bsms_plan();
  p->device = dev;
  p->N = N; p->E = E;
  p->min_out_degree = (N > 0 || E > 0) ? INT64_MAX : 0;
  const size_t nN = idx_pad(size_t(N) + 1), nE = idx_pad(size_t(E));
  const size_t words = 2 * nN + 6 * nE;
  int rc = alloc_block(&p->block, std::max<size_t>(words, 1) * sizeof(int32_t), &p->block_cap);
  if (rc) { bsms_plan_destroy(p); return rc; }
  p->host_words = words;
  if (pooled) {
    p->Nk = Nk; p->Ek = Ek; p->Ep = Ep;
    const size_t nK = idx_pad(size_t(Nk)), nNinv = idx_pad(size_t(N)), nK1 = idx_pad(size_t(Nk) + 1), nEk = idx_pad(size_t(Ek)), nEp = idx_pad(size_t(Ep));
    const size_t pw = nK + nNinv + nK1 + 2 * nEk + nN + 2 * nEp + nEk + nEp;
    if ((rc = alloc_block(&p->pool_block, std::max<size_t>(pw, 1) * sizeof(int32_t), &p->pool_cap))) { bsms_plan_destroy(p); return rc; }
  }
  point_into_blocks(p);
  int64_t n_off = 0, e_off = 0, k_off = 0, ek_off = 0, ep_off = 0;
  for (int b0 = 0; b0 < nparts; b0 += kCatParts) {
    CatArgs a{};
    a.nparts = std::min(kCatParts, nparts - b0);
    a.out_blk = p->block; a.out_pool = p->pool_block;
    a.N = (int32_t)N; a.E = (int32_t)E; a.Nk = (int32_t)Nk; a.Ek = (int32_t)Ek; a.Ep = (int32_t)Ep;
    a.has_w = bound ? 1 : 0;
    a.coo_out = coo_out; a.ids_out = pooled ? ids_out : nullptr;
    for (int i = 0; i < a.nparts; ++i) {
      const bsms_plan* q = parts[b0 + i];
      CatPart& c = a.part[i];
      c.blk = q->block; c.pool = q->pool_block;
      c.N = (int32_t)q->N; c.E = (int32_t)q->E; c.Nk = (int32_t)q->Nk; c.Ek = (int32_t)q->Ek; c.Ep = (int32_t)q->Ep;
      c.n_off = (int32_t)n_off; c.e_off = (int32_t)e_off; c.k_off = (int32_t)k_off; c.ek_off = (int32_t)ek_off; c.ep_off = (int32_t)ep_off;
      if (q->N > 0 || q->E > 0) p->min_out_degree = std::min(p->min_out_degree, q->min_out_degree);
      if (q->max_source >= 0) p->max_source = std::max(p->max_source, n_off + q->max_source);
      p->max_in_degree = std::max(p->max_in_degree, q->max_in_degree);
      p->max_out_degree = std::max(p->max_out_degree, q->max_out_degree);
      n_off += q->N; e_off += q->E; k_off += q->Nk; ek_off += q->Ek; ep_off += q->Ep;
    }
    if ((rc = launch_plan_concat(a, as_stream(stream)))) { bsms_plan_destroy(p); return rc; }
  }
  if (p->min_out_degree == INT64_MAX) p->min_out_degree = 0;
  if (bound) p->w_bound = ew_cat;
  *out = p;
  return BSMS_OK;
}

// test accessor: array `which` of a plan copied to the host as int32 words (k_w / p_w: their bit patterns); host == NULL: its length.
//  0 rowptr  1 src  2 dst  3 perm  4 t_rowptr  5 t_dst  6 t_eid  7 t_pos  8 ids  9 inv  10 k_rowptr  11 k_src  12 k_eid
//  13 p_rowptr  14 p_src  15 p_eid  16 k_w  17 p_w
extern "C" int64_t bsms_plan_export_ex(const bsms_plan_t* p, int which, int32_t* host) {
  if (!p) return -1;
  const int32_t* ptr = nullptr;
  int64_t n = 0;
  switch (which) {
    case 0: ptr = p->rowptr; n = p->N + 1; break;
    case 1: ptr = p->src; n = p->E; break;
    case 2: ptr = p->dst; n = p->E; break;
    case 3: ptr = p->perm; n = p->E; break;
    case 4: ptr = p->t_rowptr; n = p->N + 1; break;
    case 5: ptr = p->t_dst; n = p->E; break;
    case 6: ptr = p->t_eid; n = p->E; break;
    case 7: ptr = p->t_pos; n = p->E; break;
    case 8: ptr = p->ids; n = p->ids ? p->Nk : 0; break;
    case 9: ptr = p->inv; n = p->ids ? p->N : 0; break;
    case 10: ptr = p->k_rowptr; n = p->ids ? p->Nk + 1 : 0; break;
    case 11: ptr = p->k_src; n = p->Ek; break;
    case 12: ptr = p->k_eid; n = p->Ek; break;
    case 13: ptr = p->p_rowptr; n = p->ids ? p->N + 1 : 0; break;
    case 14: ptr = p->p_src; n = p->Ep; break;
    case 15: ptr = p->p_eid; n = p->Ep; break;
    case 16: ptr = reinterpret_cast<const int32_t*>(p->k_w); n = p->w_bound ? p->Ek : 0; break;
    case 17: ptr = reinterpret_cast<const int32_t*>(p->p_w); n = p->w_bound ? p->Ep : 0; break;
    default: return -1;
  }
  if (host && n > 0 && ptr) {
    if (hipMemcpy(host, ptr, size_t(n) * 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
  }
  return n;
}

extern "C" int bsms_plan_destroy(bsms_plan_t* p) {
  if (!p) return BSMS_OK;
  release_block(p->block, p->block_cap, p->device);
  release_block(p->pool_block, p->pool_cap, p->device);
  delete p;
  return BSMS_OK;
}

extern "C" int bsms_plan_pool_trim(void) {
  std::vector<PoolBlock> blocks;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    blocks.swap(g_pool);
    g_pool_bytes = 0;
  }
  for (const PoolBlock& b : blocks) (void)hipFree(b.ptr);
  return BSMS_OK;
}

extern "C" int64_t bsms_plan_num_nodes(const bsms_plan_t* p) { return p ? p->N : -1; }
extern "C" int64_t bsms_plan_num_edges(const bsms_plan_t* p) { return p ? p->E : -1; }
extern "C" int64_t bsms_plan_num_pooled(const bsms_plan_t* p) { return p ? p->Nk : -1; }
extern "C" int64_t bsms_plan_min_out_degree(const bsms_plan_t* p) { return p ? p->min_out_degree : -1; }
extern "C" int64_t bsms_plan_max_source(const bsms_plan_t* p) { return p ? p->max_source : -1; }

extern "C" int bsms_plan_export(const bsms_plan_t* p, int32_t* rowptr, int32_t* src_sorted, int32_t* perm,
                                int32_t* t_rowptr) {
  BSMS_REQUIRE(p != nullptr, BSMS_E_INVALID_ARG, "plan_export: plan is null");
  if (rowptr) BSMS_HIP_CHECK(hipMemcpy(rowptr, p->rowptr, (p->N + 1) * 4, hipMemcpyDeviceToHost));
  if (src_sorted && p->E) BSMS_HIP_CHECK(hipMemcpy(src_sorted, p->src, p->E * 4, hipMemcpyDeviceToHost));
  if (perm && p->E) BSMS_HIP_CHECK(hipMemcpy(perm, p->perm, p->E * 4, hipMemcpyDeviceToHost));
  if (t_rowptr) BSMS_HIP_CHECK(hipMemcpy(t_rowptr, p->t_rowptr, (p->N + 1) * 4, hipMemcpyDeviceToHost));
  return BSMS_OK;
}
