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

int side_lane(SideLane** out, int which) {
  int dev = 0;
  BSMS_HIP_CHECK(hipGetDevice(&dev));
  BSMS_REQUIRE(dev >= 0 && dev < 64 && which >= 0 && which < kSideLanes, BSMS_E_UNSUPPORTED, "side_lane: device %d lane %d", dev, which);
  std::lock_guard<std::mutex> lock(g_lane_mu);
  SideLane& l = g_lanes[dev][which];
  if (!l.stream) {
    BSMS_HIP_CHECK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));   // (a lower or higher stream priority for the lanes: +-0, profiles/README.md)
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

extern "C" int bsms_abi_version(void) { return 3; }  // 2: saved == NULL selects inference in *_fwd; 3: bsms_mlp_fwd_ex, larger saved buffers (bound slots)
extern "C" const char* bsms_last_error(void) { return bsms::g_err; }

namespace {
int upload(int32_t** dev, const std::vector<int32_t>& host) {
  size_t bytes = std::max<size_t>(host.size(), 1) * sizeof(int32_t);
  BSMS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dev), bytes));
  if (!host.empty())
    BSMS_HIP_CHECK(hipMemcpy(*dev, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice));
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

  std::vector<int32_t> rowptr(N + 1, 0), t_rowptr(N + 1, 0);
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
  std::partial_sum(rowptr.begin(), rowptr.end(), rowptr.begin());
  std::partial_sum(t_rowptr.begin(), t_rowptr.end(), t_rowptr.begin());

  // stable counting sorts: slots of equal key keep the caller's edge order
  std::vector<int32_t> src(E), dst(E), perm(E), slot_of_edge(E);
  {
    std::vector<int32_t> cur(rowptr.begin(), rowptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) {
      int32_t q = cur[gj[e]]++;
      src[q] = (int32_t)gi[e];
      dst[q] = (int32_t)gj[e];
      perm[q] = (int32_t)e;
      slot_of_edge[e] = q;
    }
  }
  std::vector<int32_t> t_dst(E), t_eid(E), t_pos(E);
  {
    std::vector<int32_t> cur(t_rowptr.begin(), t_rowptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) {
      int32_t t = cur[gi[e]]++;
      t_dst[t] = (int32_t)gj[e];
      t_eid[t] = (int32_t)e;
      t_pos[t] = slot_of_edge[e];
    }
  }

  bsms_plan* p = new bsms_plan();
  p->N = N;
  p->E = E;
  p->min_out_degree = min_deg;
  p->max_source = max_src;
  p->max_in_degree = max_in;
  p->max_out_degree = max_out;
  int rc = BSMS_OK;
  if ((rc = upload(&p->rowptr, rowptr)) || (rc = upload(&p->src, src)) || (rc = upload(&p->dst, dst)) ||
      (rc = upload(&p->perm, perm)) || (rc = upload(&p->t_rowptr, t_rowptr)) ||
      (rc = upload(&p->t_dst, t_dst)) || (rc = upload(&p->t_eid, t_eid)) || (rc = upload(&p->t_pos, t_pos))) {
    bsms_plan_destroy(p);
    return rc;
  }
  *out = p;
  return BSMS_OK;
}

extern "C" int bsms_plan_set_pool(bsms_plan_t* p, const int64_t* ids, int64_t Nk) {
  BSMS_REQUIRE(p != nullptr, BSMS_E_INVALID_ARG, "plan_set_pool: plan is null");
  BSMS_REQUIRE(Nk >= 0 && Nk <= p->N && (ids != nullptr || Nk == 0), BSMS_E_SHAPE,
               "plan_set_pool: bad Nk=%lld for N=%lld", (long long)Nk, (long long)p->N);
  std::vector<int32_t> h_ids(Nk), h_inv(p->N, -1);
  for (int64_t k = 0; k < Nk; ++k) {
    BSMS_REQUIRE(ids[k] >= 0 && ids[k] < p->N, BSMS_E_INVALID_ARG, "plan_set_pool: id %lld out of range",
                 (long long)ids[k]);
    BSMS_REQUIRE(h_inv[ids[k]] < 0, BSMS_E_INVALID_ARG, "plan_set_pool: duplicate id %lld", (long long)ids[k]);
    h_ids[k] = (int32_t)ids[k];
    h_inv[ids[k]] = (int32_t)k;
  }
  if (p->ids) (void)hipFree(p->ids);
  if (p->inv) (void)hipFree(p->inv);
  p->ids = p->inv = nullptr;
  int rc;
  if ((rc = upload(&p->ids, h_ids)) || (rc = upload(&p->inv, h_inv))) return rc;
  p->Nk = Nk;
  return BSMS_OK;
}

extern "C" int bsms_plan_destroy(bsms_plan_t* p) {
  if (!p) return BSMS_OK;
  int32_t* bufs[] = {p->rowptr, p->src, p->dst, p->perm, p->t_rowptr, p->t_dst, p->t_eid, p->t_pos, p->ids, p->inv};
  for (int32_t* b : bufs)
    if (b) (void)hipFree(b);
  delete p;
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
