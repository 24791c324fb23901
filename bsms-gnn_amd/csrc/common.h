// Shared host-side helpers for libbsms_hip.so (error reporting, plan struct, launch geometry).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

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
};
constexpr int kSideLanes = 2;
int side_lane(SideLane** out, int which = 0);        // for the current device; `which` < kSideLanes
int side_fork(SideLane* lane, hipStream_t main);     // side waits for main
int side_join(SideLane* lane, hipStream_t main);     // main waits for side
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

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
  int32_t *rowptr = nullptr, *src = nullptr, *dst = nullptr, *perm = nullptr;
  int32_t *t_rowptr = nullptr, *t_dst = nullptr, *t_eid = nullptr, *t_pos = nullptr;
  int32_t *ids = nullptr, *inv = nullptr;
};
