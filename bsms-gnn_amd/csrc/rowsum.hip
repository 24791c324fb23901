// HBM-bound index kernels of the BSMS path: CSR segmented row sums (edge aggregation, restrict /
// prolong transitions, gradient scatters), cal_ew, row gathers / scatters.
//
// One kernel shape serves all of them:
//     out[b, r, :] = sum_{q in [rowptr[v], rowptr[v+1])}  w[widx[q]] * X[b, xmap[xidx[q]], :]
// with v = rows[r] (or r).  Rows are contiguous runs of a sorted CSR, so a destination row is
// reduced by ONE group of lanes sequentially in the caller's edge order: no atomics, run-to-run
// reproducible, and bit-identical to a sequential scatter_add_ (utils/basic.py:324-343).
// Lanes run along the feature axis (16-byte loads, a 128-float row = one 512-byte burst of a
// half-wave), so every HBM access is a fully used, coalesced row segment.
#include "chain.h"

// hipcc defaults to -ffp-contract=fast, which would fuse the rounded product x*ew with the running sum
// into one FMA; the reference rounds the product first (x[:, i] *= ew, then scatter_add_), so keep them apart.
#pragma clang fp contract(off)

using namespace bsms;

namespace {

struct RowSumArgs {
  const int32_t* rowptr;  // [NV+1]
  const int32_t* rows;    // optional [n_out]: which CSR row feeds output row r
  const int32_t* xidx;    // optional [E]: slot -> x row (identity if null)
  const int32_t* xmap;    // optional: second-level map of the x row, negative = skip
  const float* w;         // optional weights
  const int32_t* widx;    // optional [E]: slot -> weight index (identity if null)
  const float* x;
  float* out;
  const float* addend;    // optional [B, n_out, D] (layout of `out`): out = row sum + addend (a fused residual / skip add)
  int64_t x_bstride, out_bstride;  // floats per batch item
  int32_t n_out, B, D;
};

// four consecutive features of a row tensor at ELEMENT offset e: fp32 storage, or (XBF) bf16 storage widened exactly.
// STREAM: the row is read exactly once by the whole launch (the plan-order edge aggregation: every message row belongs to
// one destination) -- a non-temporal load.  Measured on cold data (profiles/census/agg_rows.hip, airfoil L0 shape):
// fp32 31.8 -> 29.3 us (0.595 -> 0.647 of the HBM peak), bf16 messages 22.3 -> 20.4 us (0.486 -> 0.530); more rows per
// lane group with all loads hoisted were SLOWER in every variant (33.7 / 36.2 us with 2 / 4 rows).
using f32x4_t = __attribute__((ext_vector_type(4))) float;
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
template <bool XBF, bool STREAM = false>
__device__ __forceinline__ float4 ld4(const float* base, int64_t e) {
  if (XBF) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + e);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
  if (STREAM) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(base + e));
    return make_float4(v[0], v[1], v[2], v[3]);
  }
  return *reinterpret_cast<const float4*>(base + e);
}

template <bool WEIGHTED>
__device__ __forceinline__ float4 accum(float4 acc, float4 v, float w) {
  if (WEIGHTED) {  // product rounded, then added: same two roundings as x[i]*ew followed by scatter_add_
    const float px = v.x * w, py = v.y * w, pz = v.z * w, pw = v.w * w;  // contraction is off in this file
    acc.x = acc.x + px; acc.y = acc.y + py; acc.z = acc.z + pz; acc.w = acc.w + pw;
  } else {
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  return acc;
}

// LPR lanes cooperate on one output row; each lane owns float4 column groups c4, c4+LPR, ...
// One batch of U consecutive CSR slots, branch-free: all index loads, then all row loads, then the sums strictly in
// slot order.  (With the optional-pointer tests and the "skip" branch inside the unrolled loop hipcc serialises the
// batch -- an s_waitcnt vmcnt(0) before every row load -- and the coarse levels, which have too few waves to hide
// that, run at a fifth of the bandwidth.)  A skipped slot (MAPPED, negative map entry) reads row 0 and is dropped
// by a select, so the accumulator sees exactly the sequence of additions scatter_add_ performs.
template <int U, bool WEIGHTED, bool MAPPED, bool HAS_XIDX, bool HAS_WIDX, bool XBF = false>
__device__ __forceinline__ float4 rowsum_batch(const RowSumArgs& a, int64_t xcol, int q, float4 acc) {   // xcol: element offset of this lane's columns in batch item b
  int xr[U];
  float w4[U];
  float4 v4[U];
#pragma unroll
  for (int u = 0; u < U; ++u) xr[u] = HAS_XIDX ? a.xidx[q + u] : q + u;
  if (MAPPED) {
#pragma unroll
    for (int u = 0; u < U; ++u) xr[u] = a.xmap[xr[u]];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int safe = (MAPPED && xr[u] < 0) ? 0 : xr[u];
    v4[u] = ld4<XBF, !HAS_XIDX && !MAPPED>(a.x, xcol + int64_t(safe) * a.D);   // no index at all: a pure stream, each row read once
    w4[u] = WEIGHTED ? a.w[HAS_WIDX ? a.widx[q + u] : q + u] : 1.f;
  }
  __builtin_amdgcn_sched_barrier(0);   // all U row loads are in flight before the first add (hipcc would interleave)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float4 nxt = accum<WEIGHTED>(acc, v4[u], w4[u]);
    if (!MAPPED || xr[u] >= 0) acc = nxt;   // a select, not a branch
  }
  return acc;
}

// LPR lanes cooperate on one output row; each lane owns float4 column groups c4, c4+LPR, ...
// DEEP: coarse levels (few rows, up to ~170 edges each): the critical path is (edges / batch) dependent HBM round
// trips of the longest row, so batch 32 rows (128 registers; occupancy is irrelevant there).
template <int LPR, bool WEIGHTED, bool MAPPED, bool DEEP, bool HAS_XIDX, bool HAS_WIDX, bool XBF = false>
__device__ __forceinline__ void rowsum_body(const RowSumArgs& a) {
  const int64_t worker = (int64_t(blockIdx.x) * 256 + threadIdx.x) / LPR;
  const int lane = threadIdx.x % LPR;
  if (worker >= int64_t(a.B) * a.n_out) return;
  const int b = int(worker / a.n_out), r = int(worker % a.n_out);
  const int v = a.rows ? a.rows[r] : r;
  const int q0 = a.rowptr[v], q1 = a.rowptr[v + 1];
  float* ob = a.out + b * a.out_bstride + int64_t(r) * a.D;
  const int d4 = a.D >> 2;
  for (int c4 = lane; c4 < d4; c4 += LPR) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t xcol = b * a.x_bstride + c4 * 4;
    int q = q0;
    if (DEEP)
      for (; q + 32 <= q1; q += 32) acc = rowsum_batch<32, WEIGHTED, MAPPED, HAS_XIDX, HAS_WIDX, XBF>(a, xcol, q, acc);
    for (; q + 8 <= q1; q += 8) acc = rowsum_batch<8, WEIGHTED, MAPPED, HAS_XIDX, HAS_WIDX, XBF>(a, xcol, q, acc);
    // the remaining 1..7 slots as ONE batch of exactly that size: a fine-level row (6-7 edges) costs a single round
    // trip to memory instead of three or four dependent ones; the additions keep their slot order
#define BSMS_TAIL(K) case K: acc = rowsum_batch<K, WEIGHTED, MAPPED, HAS_XIDX, HAS_WIDX, XBF>(a, xcol, q, acc); break
    switch (q1 - q) { BSMS_TAIL(7); BSMS_TAIL(6); BSMS_TAIL(5); BSMS_TAIL(4); BSMS_TAIL(3); BSMS_TAIL(2); BSMS_TAIL(1); default: break; }
#undef BSMS_TAIL
    if (a.addend) {   // (row sum) + addend: one more rounded add, exactly what a separate elementwise add would do
      const float4 ad = *reinterpret_cast<const float4*>(a.addend + b * a.out_bstride + int64_t(r) * a.D + c4 * 4);
      acc.x += ad.x; acc.y += ad.y; acc.z += ad.z; acc.w += ad.w;
    }
    *reinterpret_cast<float4*>(ob + c4 * 4) = acc;
  }
}

template <int LPR, bool WEIGHTED, bool MAPPED, bool DEEP = false>
__global__ __launch_bounds__(256) void k_rowsum_v4(RowSumArgs a) {
  // the optional index arrays are tested once (uniform), not per slot
  const bool hx = a.xidx != nullptr, hw = WEIGHTED && a.widx != nullptr;
  if (hx) {
    if (hw) rowsum_body<LPR, WEIGHTED, MAPPED, DEEP, true, WEIGHTED>(a);
    else rowsum_body<LPR, WEIGHTED, MAPPED, DEEP, true, false>(a);
  } else {
    if (hw) rowsum_body<LPR, WEIGHTED, MAPPED, DEEP, false, WEIGHTED>(a);
    else rowsum_body<LPR, WEIGHTED, MAPPED, DEEP, false, false>(a);
  }
}

// the edge aggregation of the bf16 precision: messages stored as bf16, sums and output fp32.
// A lane owns EIGHT features (one 16-byte load per edge row, as in the fp32 kernel) and LPR = D / 8 lanes share an output
// row: with four features per lane (8-byte loads) a wave had half the bytes in flight and the launch ran at 0.48 of the HBM
// peak against 0.64 for the fp32 kernel on twice the bytes.  Every feature still sums its rows in slot order.
template <int U>
__device__ __forceinline__ void rowsum_batch_bf8(const unsigned short* x, int64_t xcol, int q, int D, float (&acc)[8]) {
  uint4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {   // every message row is read once: non-temporal (see ld4)
    const u32x4_t w = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(x + xcol + int64_t(q + u) * D));
    v[u] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __builtin_amdgcn_sched_barrier(0);   // all U row loads are in flight before the first add
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[2 * k] += __uint_as_float(w[k] << 16);
      acc[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
    }
  }
}
template <int LPR, bool DEEP>
__global__ __launch_bounds__(256) void k_rowsum_bf16in(RowSumArgs a) {   // D == 8 * LPR, plan order, no weights / maps / addend
  const int64_t worker = (int64_t(blockIdx.x) * 256 + threadIdx.x) / LPR;
  const int lane = threadIdx.x % LPR;
  if (worker >= int64_t(a.B) * a.n_out) return;
  const int b = int(worker / a.n_out), r = int(worker % a.n_out);
  const int q0 = a.rowptr[r], q1 = a.rowptr[r + 1];
  const unsigned short* x = reinterpret_cast<const unsigned short*>(a.x);
  const int64_t xcol = b * a.x_bstride + lane * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int q = q0;
  if (DEEP)
    for (; q + 16 <= q1; q += 16) rowsum_batch_bf8<16>(x, xcol, q, a.D, acc);
  for (; q + 8 <= q1; q += 8) rowsum_batch_bf8<8>(x, xcol, q, a.D, acc);
#define BSMS_TAIL(K) case K: rowsum_batch_bf8<K>(x, xcol, q, a.D, acc); break
  switch (q1 - q) { BSMS_TAIL(7); BSMS_TAIL(6); BSMS_TAIL(5); BSMS_TAIL(4); BSMS_TAIL(3); BSMS_TAIL(2); BSMS_TAIL(1); default: break; }
#undef BSMS_TAIL
  float4* ob = reinterpret_cast<float4*>(a.out + b * a.out_bstride + int64_t(r) * a.D + lane * 8);
  ob[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  ob[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// two unweighted, unmapped sums of the SAME shape in one launch (blockIdx.y picks the job): the scatters of the first
// edge gradient to its source and to its target rows
template <int LPR, bool DEEP>
__global__ __launch_bounds__(256) void k_rowsum_pair(RowSumArgs a0, RowSumArgs a1) {
  const RowSumArgs& a = blockIdx.y ? a1 : a0;
  if (a.xidx) rowsum_body<LPR, false, false, DEEP, true, false>(a);
  else rowsum_body<LPR, false, false, DEEP, false, false>(a);
}

// ---- the two scatters of the first edge gradient + the narrow weight gradient of the first edge Linear, ONE launch.
// blockIdx.y = 0: by source (transpose CSR, gathered rows), as k_rowsum_pair.  blockIdx.y = 1: by target -- the rows of
// a target are contiguous in plan order, and while a lane streams them it also accumulates, for the fiber columns s
// of the Linear (ops/basic.py:84-90), sum_e g[e][f] * fiber[e][s]: the gradient of those columns of W0 and (s = NS,
// the plain row sum) of its bias.  The 256 / LPR workers of a workgroup are combined in fixed order through LDS and
// written as one partial block; k_small_reduce (wgrad.hip) sums the blocks in fixed order.  This replaces a separate
// pass over the whole [B,E,D] gradient (k_small_wgrad read it a third time: 128 MB at airfoil L0).
template <int U, int NS, bool XBF = false>
__device__ __forceinline__ void fiber_batch(const float* xbase, int64_t xcol, const float* frow, int D, int ld, int q, float4& acc,
                                            float4 (&accf)[NS]) {
  float4 v[U], f0[U], f1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    v[u] = ld4<XBF>(xbase, xcol + int64_t(q + u) * D);
    f0[u] = *reinterpret_cast<const float4*>(frow + int64_t(q + u) * ld);
    if (NS > 4) f1[u] = *reinterpret_cast<const float4*>(frow + int64_t(q + u) * ld + 4);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
    const float fs[8] = {f0[u].x, f0[u].y, f0[u].z, f0[u].w, NS > 4 ? f1[u].x : 0.f, NS > 4 ? f1[u].y : 0.f,
                         NS > 4 ? f1[u].z : 0.f, NS > 4 ? f1[u].w : 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      accf[s].x = fmaf(v[u].x, fs[s], accf[s].x); accf[s].y = fmaf(v[u].y, fs[s], accf[s].y);
      accf[s].z = fmaf(v[u].z, fs[s], accf[s].z); accf[s].w = fmaf(v[u].w, fs[s], accf[s].w);
    }
  }
}

constexpr int kSmallRows = 10;   // partial block = kSmallRows x D floats: rows 0..7 narrow products, 8 colsum(G), 9 colsum(S) (wgrad.hip)

template <int LPR, int NS, bool DEEP, bool XBF = false>
__global__ __launch_bounds__(256) void k_rowsum_pair_fiber(RowSumArgs a0, RowSumArgs a1, const float* fiber, int ld, float* part,
                                                          int nsrc_blocks) {
  // blocks [0, nsrc_blocks): by source, one worker per output row (as k_rowsum_pair); the remaining gridDim.x -
  // nsrc_blocks blocks: by target, workers stride over the rows so that the number of partial blocks stays small
  if (int(blockIdx.x) < nsrc_blocks) {
    if (a0.xidx) rowsum_body<LPR, false, false, DEEP, true, false, XBF>(a0);
    else rowsum_body<LPR, false, false, DEEP, false, false, XBF>(a0);
    return;
  }
  __shared__ float4 red[(NS + 1) * 256];
  const RowSumArgs& a = a1;
  const int ntgt = int(gridDim.x) - nsrc_blocks, j = int(blockIdx.x) - nsrc_blocks;
  constexpr int WPB = 256 / LPR;   // workers per block
  const int lane = threadIdx.x % LPR;
  const int64_t total = int64_t(a.B) * a.n_out;
  float4 accs = make_float4(0.f, 0.f, 0.f, 0.f), accf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) accf[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t worker = int64_t(j) * WPB + threadIdx.x / LPR; worker < total; worker += int64_t(ntgt) * WPB) {
    // D == 4 * LPR: one float4 column group per lane
    const int b = int(worker / a.n_out), r = int(worker % a.n_out);
    const int q0 = a.rowptr[r], q1 = a.rowptr[r + 1];
    const int64_t xcol = b * a.x_bstride + lane * 4;
    const float* frow = fiber + int64_t(b) * (a.x_bstride / a.D) * ld;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int q = q0;
    if (DEEP)
      for (; q + 16 <= q1; q += 16) fiber_batch<16, NS, XBF>(a.x, xcol, frow, a.D, ld, q, acc, accf);
    for (; q + 8 <= q1; q += 8) fiber_batch<8, NS, XBF>(a.x, xcol, frow, a.D, ld, q, acc, accf);
#define BSMS_TAIL(K) case K: fiber_batch<K, NS, XBF>(a.x, xcol, frow, a.D, ld, q, acc, accf); break
    switch (q1 - q) { BSMS_TAIL(7); BSMS_TAIL(6); BSMS_TAIL(5); BSMS_TAIL(4); BSMS_TAIL(3); BSMS_TAIL(2); BSMS_TAIL(1); default: break; }
#undef BSMS_TAIL
    *reinterpret_cast<float4*>(a.out + b * a.out_bstride + int64_t(r) * a.D + lane * 4) = acc;
    accs.x += acc.x; accs.y += acc.y; accs.z += acc.z; accs.w += acc.w;
  }
  float* blk = part + int64_t(j) * kSmallRows * a.D;
  // combine the workers of the workgroup in fixed order: all NS + 1 sums through LDS in ONE round (one barrier instead of
  // 2 (NS + 1): a coarse-level launch of this kernel is nothing but its dependent round trips)
#pragma unroll
  for (int s = 0; s <= NS; ++s) red[s * 256 + threadIdx.x] = s < NS ? accf[s] : accs;
  __syncthreads();
  for (int idx = threadIdx.x; idx < (NS + 1) * LPR; idx += 256) {
    const int s = idx / LPR, t = idx % LPR;
    float4 v = red[s * 256 + t];
    for (int w = 1; w < WPB; ++w) {
      const float4 o = red[s * 256 + w * LPR + t];
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *reinterpret_cast<float4*>(blk + (s < NS ? s : 8) * a.D + t * 4) = v;
  }
}

// any D (positions: D = 2 or 3; 1-D sums): one thread per output element, slots in branch-free batches like rowsum_batch
// (a one-slot-at-a-time loop costs two dependent round trips per edge: the coarse position restricts -- up to 166
// edges per row -- took 60-160 us each)
template <int U, bool WEIGHTED, bool MAPPED, bool HAS_XIDX, bool HAS_WIDX>
__device__ __forceinline__ float scalar_batch(const RowSumArgs& a, const float* xcol, int q, float acc) {
  int xr[U];
  float w[U], v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) xr[u] = HAS_XIDX ? a.xidx[q + u] : q + u;
  if (MAPPED) {
#pragma unroll
    for (int u = 0; u < U; ++u) xr[u] = a.xmap[xr[u]];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int safe = (MAPPED && xr[u] < 0) ? 0 : xr[u];
    v[u] = xcol[int64_t(safe) * a.D];
    w[u] = WEIGHTED ? a.w[HAS_WIDX ? a.widx[q + u] : q + u] : 1.f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float nxt;
    if (WEIGHTED) { const float prod = v[u] * w[u]; nxt = acc + prod; }   // contraction is off in this file
    else nxt = acc + v[u];
    if (!MAPPED || xr[u] >= 0) acc = nxt;
  }
  return acc;
}

template <bool WEIGHTED, bool MAPPED, bool HAS_XIDX, bool HAS_WIDX>
__device__ __forceinline__ void scalar_body(const RowSumArgs& a) {
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int64_t total = int64_t(a.B) * a.n_out * a.D;
  if (t >= total) return;
  const int c = int(t % a.D);
  const int64_t br = t / a.D;
  const int b = int(br / a.n_out), r = int(br % a.n_out);
  const int v = a.rows ? a.rows[r] : r;
  const float* xcol = a.x + b * a.x_bstride + c;
  float acc = 0.f;
  int q = a.rowptr[v];
  const int q1 = a.rowptr[v + 1];
  for (; q + 16 <= q1; q += 16) acc = scalar_batch<16, WEIGHTED, MAPPED, HAS_XIDX, HAS_WIDX>(a, xcol, q, acc);
  for (; q + 4 <= q1; q += 4) acc = scalar_batch<4, WEIGHTED, MAPPED, HAS_XIDX, HAS_WIDX>(a, xcol, q, acc);
  for (; q < q1; ++q) acc = scalar_batch<1, WEIGHTED, MAPPED, HAS_XIDX, HAS_WIDX>(a, xcol, q, acc);
  const int64_t o = b * a.out_bstride + int64_t(r) * a.D + c;
  if (a.addend) acc += a.addend[o];
  a.out[o] = acc;
}

template <bool WEIGHTED, bool MAPPED>
__global__ __launch_bounds__(256) void k_rowsum_scalar(RowSumArgs a) {
  const bool hx = a.xidx != nullptr, hw = WEIGHTED && a.widx != nullptr;
  if (hx) {
    if (hw) scalar_body<WEIGHTED, MAPPED, true, WEIGHTED>(a);
    else scalar_body<WEIGHTED, MAPPED, true, false>(a);
  } else {
    if (hw) scalar_body<WEIGHTED, MAPPED, false, WEIGHTED>(a);
    else scalar_body<WEIGHTED, MAPPED, false, false>(a);
  }
}

// below this many lanes the chip is under-filled (256 CUs x 2048 threads) and latency, not bandwidth, is the limit
constexpr int64_t kDeepBelowThreads = 400 * 1024;

template <bool WEIGHTED, bool MAPPED>
int launch_rowsum_wm(const RowSumArgs& a, hipStream_t s) {
  const int64_t workers = int64_t(a.B) * a.n_out;
  if (workers == 0 || a.D == 0) return BSMS_OK;
  if (a.D % 4 != 0) {
    const int64_t total = workers * a.D;
    hipLaunchKernelGGL((k_rowsum_scalar<WEIGHTED, MAPPED>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, a);
  } else {
    const int d4 = a.D / 4;
#define BSMS_RS(L)                                                                                              \
  do {                                                                                                          \
    if (workers * L < kDeepBelowThreads)                                                                        \
      hipLaunchKernelGGL((k_rowsum_v4<L, WEIGHTED, MAPPED, true>), dim3((unsigned)ceil_div(workers * L, 256)),  \
                         dim3(256), 0, s, a);                                                                   \
    else                                                                                                        \
      hipLaunchKernelGGL((k_rowsum_v4<L, WEIGHTED, MAPPED, false>), dim3((unsigned)ceil_div(workers * L, 256)), \
                         dim3(256), 0, s, a);                                                                   \
  } while (0)
    if (d4 <= 1) BSMS_RS(1);
    else if (d4 <= 2) BSMS_RS(2);
    else if (d4 <= 4) BSMS_RS(4);
    else if (d4 <= 8) BSMS_RS(8);
    else if (d4 <= 16) BSMS_RS(16);
    else if (d4 <= 32) BSMS_RS(32);
    else BSMS_RS(64);
#undef BSMS_RS
  }
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

int launch_rowsum(const RowSumArgs& a, hipStream_t s) {
  const bool wt = a.w != nullptr, mp = a.xmap != nullptr;
  if (wt && mp) return launch_rowsum_wm<true, true>(a, s);
  if (wt) return launch_rowsum_wm<true, false>(a, s);
  if (mp) return launch_rowsum_wm<false, true>(a, s);
  return launch_rowsum_wm<false, false>(a, s);
}

// out[b, oidx[r], :] = in[b, iidx[r], :]   (either index optional)
template <typename IndexT>
__global__ __launch_bounds__(256) void k_copy_rows(const float* in, float* out, const IndexT* iidx,
                                                   const IndexT* oidx, int64_t in_bstride, int64_t out_bstride,
                                                   int32_t n_rows, int32_t B, int32_t D) {
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const bool v4 = (D % 4) == 0;
  const int per_row = v4 ? D / 4 : D;
  if (t >= int64_t(B) * n_rows * per_row) return;
  const int c = int(t % per_row);
  const int64_t br = t / per_row;
  const int b = int(br / n_rows), r = int(br % n_rows);
  const int64_t ir = iidx ? int64_t(iidx[r]) : r, orow = oidx ? int64_t(oidx[r]) : r;
  const float* src = in + b * in_bstride + ir * D;
  float* dst = out + b * out_bstride + orow * D;
  if (v4) reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(src)[c];
  else dst[c] = src[c];
}

__global__ __launch_bounds__(256) void k_cal_ew_nodes(const int32_t* rowptr, const int32_t* src,
                                                      const int32_t* t_rowptr, const float* w, float* aggr_w,
                                                      int32_t N) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  float s = 0.f;
  for (int q = rowptr[j]; q < rowptr[j + 1]; ++q) {
    const int i = src[q];
    const float deg = float(t_rowptr[i + 1] - t_rowptr[i]);       // degree(g[0])  utils/basic.py:305-309
    s = __fadd_rn(s, __fdiv_rn(w[i], deg));                        // ops/basic.py:160-164
  }
  aggr_w[j] = __fadd_rn(s, 1e-12f);
}

__global__ __launch_bounds__(256) void k_cal_ew_edges(const int32_t* src, const int32_t* dst, const int32_t* perm,
                                                      const int32_t* t_rowptr, const float* w, const float* aggr_w,
                                                      float* ec, int32_t E) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= E) return;
  const int i = src[q];
  const float deg = float(t_rowptr[i + 1] - t_rowptr[i]);
  ec[perm[q]] = __fdiv_rn(__fdiv_rn(w[i], deg), aggr_w[dst[q]]);   // ops/basic.py:165
}

}  // namespace

// Exposed to the other translation units of the library (gmp.hip): plan-order segment sums.
namespace bsms {
int rowsum_plan_order(const bsms_plan* p, const float* x, int64_t B, int64_t D, float* out, hipStream_t s) {
  RowSumArgs a{};
  a.rowptr = p->rowptr;
  a.x = x; a.out = out;
  a.x_bstride = p->E * D; a.out_bstride = p->N * D;
  a.n_out = (int32_t)p->N; a.B = (int32_t)B; a.D = (int32_t)D;
  return launch_rowsum(a, s);
}
// the same with bf16 messages (bf16 precision of the GMP block): D = 128 / 256 only
int rowsum_plan_order_bf16(const bsms_plan* p, const float* x_bf16, int64_t B, int64_t D, float* out, hipStream_t s) {
  BSMS_REQUIRE(D == 128 || D == 256, BSMS_E_UNSUPPORTED, "bf16 aggregation: D=%lld (128, 256)", (long long)D);
  RowSumArgs a{};
  a.rowptr = p->rowptr;
  a.x = x_bf16; a.out = out;
  a.x_bstride = p->E * D; a.out_bstride = p->N * D;
  a.n_out = (int32_t)p->N; a.B = (int32_t)B; a.D = (int32_t)D;
  const int64_t workers = B * p->N;
  if (workers == 0) return BSMS_OK;
  const int lpr = int(D / 8);   // eight features per lane
  const dim3 grid((unsigned)ceil_div(workers * lpr, 256));
  const bool deep = workers * (D / 4) < kDeepBelowThreads;   // the same levels as the fp32 kernel
  if (D == 128) {
    if (deep) hipLaunchKernelGGL((k_rowsum_bf16in<16, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_rowsum_bf16in<16, false>), grid, dim3(256), 0, s, a);
  } else {
    if (deep) hipLaunchKernelGGL((k_rowsum_bf16in<32, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_rowsum_bf16in<32, false>), grid, dim3(256), 0, s, a);
  }
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
// out[b, i, :] = sum over edges whose SOURCE is i of x[b, slot(e), :]   (x in plan order)
int rowsum_by_source(const bsms_plan* p, const float* x, int64_t B, int64_t D, float* out, hipStream_t s) {
  RowSumArgs a{};
  a.rowptr = p->t_rowptr; a.xidx = p->t_pos;
  a.x = x; a.out = out;
  a.x_bstride = p->E * D; a.out_bstride = p->N * D;
  a.n_out = (int32_t)p->N; a.B = (int32_t)B; a.D = (int32_t)D;
  return launch_rowsum(a, s);
}
// by source and by target at once + the narrow weight-gradient partials (see k_rowsum_pair_fiber); returns the number
// of partial blocks in *nwg, or 0 there if this shape is not built (the caller then runs the separate kernels)
int rowsum_source_target_fiber(const bsms_plan* p, const float* x, int64_t B, int64_t D, float* outS, float* outD,
                               const float* fiber, int ld, int ncols, float* part, int64_t part_blocks, int* nwg, hipStream_t s,
                               bool x_bf16) {
  *nwg = 0;
  const int64_t workers = B * p->N;
  const int lpr = int(D / 4);
  const int64_t blocks = ceil_div(workers * lpr, 256);
  const int64_t tgt_blocks = std::min<int64_t>(blocks, std::min<int64_t>(part_blocks, 1024));   // = partial blocks to reduce
  if (workers == 0 || (D != 128 && D != 256) || ncols < 1 || ncols > 4 || (ld & 3) || tgt_blocks < 1) return BSMS_OK;
  RowSumArgs a0{}, a1{};
  a0.rowptr = p->t_rowptr; a0.xidx = p->t_pos;
  a1.rowptr = p->rowptr;
  a0.x = a1.x = x; a0.out = outS; a1.out = outD;
  a0.x_bstride = a1.x_bstride = p->E * D; a0.out_bstride = a1.out_bstride = p->N * D;
  a0.n_out = a1.n_out = (int32_t)p->N; a0.B = a1.B = (int32_t)B; a0.D = a1.D = (int32_t)D;
  const dim3 grid((unsigned)(blocks + tgt_blocks));
  const bool deep = workers * lpr < kDeepBelowThreads;
  const int nsrc = (int)blocks;
#define BSMS_PF(L, NS)                                                                                                                 \
  do {                                                                                                                                 \
    if (x_bf16) {                                                                                                                      \
      if (deep) hipLaunchKernelGGL((k_rowsum_pair_fiber<L, NS, true, true>), grid, dim3(256), 0, s, a0, a1, fiber, ld, part, nsrc);    \
      else hipLaunchKernelGGL((k_rowsum_pair_fiber<L, NS, false, true>), grid, dim3(256), 0, s, a0, a1, fiber, ld, part, nsrc);        \
    } else if (deep) hipLaunchKernelGGL((k_rowsum_pair_fiber<L, NS, true>), grid, dim3(256), 0, s, a0, a1, fiber, ld, part, nsrc);     \
    else hipLaunchKernelGGL((k_rowsum_pair_fiber<L, NS, false>), grid, dim3(256), 0, s, a0, a1, fiber, ld, part, nsrc);                \
  } while (0)
  if (D == 128) {
    if (ncols == 1) BSMS_PF(32, 1); else if (ncols == 2) BSMS_PF(32, 2); else if (ncols == 3) BSMS_PF(32, 3); else BSMS_PF(32, 4);
  } else {
    if (ncols == 1) BSMS_PF(64, 1); else if (ncols == 2) BSMS_PF(64, 2); else if (ncols == 3) BSMS_PF(64, 3); else BSMS_PF(64, 4);
  }
#undef BSMS_PF
  BSMS_LAUNCH_CHECK();
  *nwg = (int)tgt_blocks;
  return BSMS_OK;
}
// by source and by target at once: outS[b,i,:] = sum_{e: src(e)=i} x[b,slot(e),:], outD[b,j,:] = sum_{e: dst(e)=j} x[b,slot(e),:]
int rowsum_source_and_target(const bsms_plan* p, const float* x, int64_t B, int64_t D, float* outS, float* outD, hipStream_t s) {
  if (D != 128) {   // the pair kernel is built for the 32-lanes-per-row shape only
    int rc = rowsum_by_source(p, x, B, D, outS, s);
    return rc ? rc : rowsum_plan_order(p, x, B, D, outD, s);
  }
  RowSumArgs a0{}, a1{};
  a0.rowptr = p->t_rowptr; a0.xidx = p->t_pos;
  a1.rowptr = p->rowptr;
  a0.x = a1.x = x; a0.out = outS; a1.out = outD;
  a0.x_bstride = a1.x_bstride = p->E * D; a0.out_bstride = a1.out_bstride = p->N * D;
  a0.n_out = a1.n_out = (int32_t)p->N; a0.B = a1.B = (int32_t)B; a0.D = a1.D = (int32_t)D;
  const int64_t workers = B * p->N;
  if (workers == 0) return BSMS_OK;
  const dim3 grid((unsigned)ceil_div(workers * 32, 256), 2);
  if (workers * 32 < kDeepBelowThreads) hipLaunchKernelGGL((k_rowsum_pair<32, true>), grid, dim3(256), 0, s, a0, a1);
  else hipLaunchKernelGGL((k_rowsum_pair<32, false>), grid, dim3(256), 0, s, a0, a1);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
}  // namespace bsms

extern "C" int bsms_segment_sum_fwd(const bsms_plan_t* p, const float* src, int64_t B, int64_t D, int plan_order,
                                    float* out, bsms_stream_t stream) {
  BSMS_REQUIRE(p && out && (src || p->E == 0), BSMS_E_INVALID_ARG, "segment_sum_fwd: null argument");
  BSMS_REQUIRE(B >= 0 && D >= 1, BSMS_E_SHAPE, "segment_sum_fwd: bad B=%lld D=%lld", (long long)B, (long long)D);
  RowSumArgs a{};
  a.rowptr = p->rowptr;
  a.xidx = plan_order ? nullptr : p->perm;
  a.x = src; a.out = out;
  a.x_bstride = p->E * D; a.out_bstride = p->N * D;
  a.n_out = (int32_t)p->N; a.B = (int32_t)B; a.D = (int32_t)D;
  return launch_rowsum(a, as_stream(stream));
}

extern "C" int bsms_segment_sum_bf16(const bsms_plan_t* p, const void* src_bf16, int64_t B, int64_t D, float* out,
                                     bsms_stream_t stream) {
  BSMS_REQUIRE(p && out && (src_bf16 || p->E == 0), BSMS_E_INVALID_ARG, "segment_sum_bf16: null argument");
  BSMS_REQUIRE(B >= 0, BSMS_E_SHAPE, "segment_sum_bf16: bad B=%lld", (long long)B);
  return bsms::rowsum_plan_order_bf16(p, reinterpret_cast<const float*>(src_bf16), B, D, out, as_stream(stream));
}

extern "C" int bsms_segment_sum_bwd(const bsms_plan_t* p, const float* grad_out, int64_t B, int64_t D,
                                    float* grad_src, bsms_stream_t stream) {
  BSMS_REQUIRE(p && grad_out && (grad_src || p->E == 0), BSMS_E_INVALID_ARG, "segment_sum_bwd: null argument");
  BSMS_REQUIRE(B >= 0 && D >= 1, BSMS_E_SHAPE, "segment_sum_bwd: bad B/D");
  const int per_row = (D % 4 == 0) ? int(D / 4) : int(D);
  const int64_t total = B * p->E * per_row;
  if (total == 0) return BSMS_OK;
  hipLaunchKernelGGL((k_copy_rows<int32_t>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                     grad_out, grad_src, (const int32_t*)p->dst, (const int32_t*)p->perm, p->N * D, p->E * D,
                     (int32_t)p->E, (int32_t)B, (int32_t)D);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

extern "C" int bsms_cal_ew(const bsms_plan_t* p, const float* w, float* ec, float* aggr_w, bsms_stream_t stream) {
  BSMS_REQUIRE(p && w && aggr_w && (ec || p->E == 0), BSMS_E_INVALID_ARG, "cal_ew: null argument");
  hipStream_t s = as_stream(stream);
  if (p->N > 0) {
    hipLaunchKernelGGL(k_cal_ew_nodes, dim3((unsigned)ceil_div(p->N, 256)), dim3(256), 0, s, p->rowptr, p->src,
                       p->t_rowptr, w, aggr_w, (int32_t)p->N);
    BSMS_LAUNCH_CHECK();
  }
  if (p->E > 0) {
    hipLaunchKernelGGL(k_cal_ew_edges, dim3((unsigned)ceil_div(p->E, 256)), dim3(256), 0, s, p->src, p->dst, p->perm,
                       p->t_rowptr, w, aggr_w, ec, (int32_t)p->E);
    BSMS_LAUNCH_CHECK();
  }
  return BSMS_OK;
}

// (lives here, not in plan.hip: that file is host-only C++ -- the sanitizer build compiles it with g++)
namespace {
__global__ __launch_bounds__(256) void k_bind_ew(const float* ew, const int32_t* k_eid, float* k_w, int32_t Ek, const int32_t* p_eid,
                                                 float* p_w, int32_t Ep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < Ek) k_w[i] = ew[k_eid[i]];
  if (i < Ep) p_w[i] = ew[p_eid[i]];
}
}  // namespace

// ---- bsms_plan_concat (host side in plan.hip): blockIdx.y = part, the threads of the x dimension stride over the longest array of
// the part; every array of the part is copied to its place in the union with the offsets of common.h: CatPart added.  Entry N_b of
// a part's rowptr arrays coincides with entry 0 of the next part's (same value): a benign double write.
namespace {
struct PlanView {
  const int32_t *rowptr, *t_rowptr, *src, *dst, *perm, *t_dst, *t_eid, *t_pos;
  const int32_t *ids, *inv, *k_rowptr, *k_src, *k_eid, *p_rowptr, *p_src, *p_eid, *k_w, *p_w;
};
__device__ __forceinline__ PlanView view_of(const int32_t* blk, const int32_t* pool, int64_t N, int64_t E, int64_t Nk, int64_t Ek, int64_t Ep) {
  PlanView v;
  const size_t nN = idx_pad(size_t(N) + 1), nE = idx_pad(size_t(E));
  v.rowptr = blk; v.t_rowptr = blk + nN; v.src = v.t_rowptr + nN; v.dst = v.src + nE; v.perm = v.dst + nE;
  v.t_dst = v.perm + nE; v.t_eid = v.t_dst + nE; v.t_pos = v.t_eid + nE;
  const size_t nK = idx_pad(size_t(Nk)), nNinv = idx_pad(size_t(N)), nK1 = idx_pad(size_t(Nk) + 1), nEk = idx_pad(size_t(Ek)), nEp = idx_pad(size_t(Ep));
  v.ids = pool; v.inv = pool + nK; v.k_rowptr = v.inv + nNinv; v.k_src = v.k_rowptr + nK1; v.k_eid = v.k_src + nEk;
  v.p_rowptr = v.k_eid + nEk; v.p_src = v.p_rowptr + nN; v.p_eid = v.p_src + nEp; v.k_w = v.p_eid + nEp; v.p_w = v.k_w + nEk;
  return v;
}
__global__ __launch_bounds__(256) void k_plan_concat(bsms::CatArgs a) {
  const bsms::CatPart c = a.part[blockIdx.y];
  const PlanView in = view_of(c.blk, c.pool, c.N, c.E, c.Nk, c.Ek, c.Ep);
  const PlanView o = view_of(a.out_blk, a.out_pool, a.N, a.E, a.Nk, a.Ek, a.Ep);
  int32_t* const* ow = reinterpret_cast<int32_t* const*>(&o);   // (the union's arrays are written: same layout, mutable block)
  auto W = [&](const int32_t* field) { return const_cast<int32_t*>(field); };
  (void)ow;
  const int longest = max(max(c.N + 1, c.E), max(max(c.Nk + 1, c.Ek), c.Ep));
  const bool pooled = a.out_pool != nullptr;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < longest; i += gridDim.x * blockDim.x) {
    if (i <= c.N) {
      W(o.rowptr)[c.n_off + i] = in.rowptr[i] + c.e_off;
      W(o.t_rowptr)[c.n_off + i] = in.t_rowptr[i] + c.e_off;
      if (pooled) W(o.p_rowptr)[c.n_off + i] = in.p_rowptr[i] + c.ep_off;
    }
    if (i < c.N && pooled) {
      const int32_t v = in.inv[i];
      W(o.inv)[c.n_off + i] = v < 0 ? -1 : v + c.k_off;
    }
    if (i < c.E) {
      const int32_t sq = in.src[i] + c.n_off, dq = in.dst[i] + c.n_off, pq = in.perm[i] + c.e_off;
      W(o.src)[c.e_off + i] = sq;
      W(o.dst)[c.e_off + i] = dq;
      W(o.perm)[c.e_off + i] = pq;
      W(o.t_dst)[c.e_off + i] = in.t_dst[i] + c.n_off;
      W(o.t_eid)[c.e_off + i] = in.t_eid[i] + c.e_off;
      W(o.t_pos)[c.e_off + i] = in.t_pos[i] + c.e_off;
      if (a.coo_out) { a.coo_out[pq] = sq; a.coo_out[int64_t(a.E) + pq] = dq; }   // plan slot -> the caller's edge id
    }
    if (pooled) {
      if (i <= c.Nk) W(o.k_rowptr)[c.k_off + i] = in.k_rowptr[i] + c.ek_off;
      if (i < c.Nk) {
        const int32_t v = in.ids[i] + c.n_off;
        W(o.ids)[c.k_off + i] = v;
        if (a.ids_out) a.ids_out[c.k_off + i] = v;
      }
      if (i < c.Ek) {
        W(o.k_src)[c.ek_off + i] = in.k_src[i] + c.n_off;
        W(o.k_eid)[c.ek_off + i] = in.k_eid[i] + c.e_off;
        if (a.has_w) W(o.k_w)[c.ek_off + i] = in.k_w[i];
      }
      if (i < c.Ep) {
        W(o.p_src)[c.ep_off + i] = in.p_src[i] + c.k_off;   // COARSE row of the target
        W(o.p_eid)[c.ep_off + i] = in.p_eid[i] + c.e_off;
        if (a.has_w) W(o.p_w)[c.ep_off + i] = in.p_w[i];
      }
    }
  }
}
}  // namespace

int bsms::launch_plan_concat(const CatArgs& a, hipStream_t s) {
  int longest = 1;
  for (int b = 0; b < a.nparts; ++b)
    longest = std::max(longest, std::max(std::max(a.part[b].N + 1, a.part[b].E), std::max(std::max(a.part[b].Nk + 1, a.part[b].Ek), a.part[b].Ep)));
  hipLaunchKernelGGL(k_plan_concat, dim3((unsigned)std::min<int64_t>(ceil_div(longest, 256), 64), (unsigned)a.nparts), dim3(256), 0, s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

extern "C" int bsms_plan_bind_edge_weights(bsms_plan_t* p, const float* ew, bsms_stream_t stream) {
  BSMS_REQUIRE(p != nullptr, BSMS_E_INVALID_ARG, "plan_bind_edge_weights: plan is null");
  if (!ew) { p->w_bound = nullptr; return BSMS_OK; }
  BSMS_REQUIRE(p->ids != nullptr, BSMS_E_INVALID_ARG, "plan_bind_edge_weights: the plan has no pool (bsms_plan_set_pool)");
  // Already bound to ANOTHER weight tensor (the same coarse plan under a different level-0 plan): the gathered copies stay as
  // they are -- kernels still queued on other streams and captured HIP graphs have their pointers baked in --
  // and this tensor takes the unbound path of bsms_edge_conv (same sums in the same order, two dependent loads more per row).
  if (p->w_bound && p->w_bound != ew) return BSMS_OK;
  if (p->w_bound == ew) return BSMS_OK;   // nothing to do: the copies were gathered from this tensor (mesh-static weights; refresh = bind NULL, then bind again)
  const int64_t n = std::max(p->Ek, p->Ep);
  if (n > 0) {
    hipLaunchKernelGGL(k_bind_ew, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), ew, p->k_eid, p->k_w,
                       (int32_t)p->Ek, p->p_eid, p->p_w, (int32_t)p->Ep);
    BSMS_LAUNCH_CHECK();
  }
  p->w_bound = ew;
  return BSMS_OK;
}

extern "C" const float* bsms_plan_bound_edge_weights(const bsms_plan_t* p) { return p ? p->w_bound : nullptr; }

extern "C" int bsms_edge_conv(const bsms_plan_t* p, const float* x, int64_t B, int64_t D, const float* ew,
                              int aggregating, int pooled, float* out, bsms_stream_t stream) {
  return bsms::edge_conv_add(p, x, B, D, ew, aggregating, pooled, out, nullptr, as_stream(stream));
}

// bsms_edge_conv with an optional fused `+ addend` (same layout as `out`): the U-Net's backward adds the gradient that
// arrived at a skip connection to the adjoint of the restriction (csrc/bsgmp.hip) without a separate elementwise pass.
int bsms::edge_conv_add(const bsms_plan* p, const float* x, int64_t B, int64_t D, const float* ew, int aggregating,
                        int pooled, float* out, const float* addend, hipStream_t stream) {
  BSMS_REQUIRE(p && x && ew && out, BSMS_E_INVALID_ARG, "edge_conv: null argument");
  BSMS_REQUIRE(B >= 0 && D >= 1, BSMS_E_SHAPE, "edge_conv: bad B=%lld D=%lld", (long long)B, (long long)D);
  BSMS_REQUIRE(!pooled || p->ids, BSMS_E_INVALID_ARG, "edge_conv: pooled=1 but the plan has no pool (bsms_plan_set_pool)");
  RowSumArgs a{};
  a.w = ew; a.x = x; a.out = out;
  a.B = (int32_t)B; a.D = (int32_t)D;
  if (pooled && ew == p->w_bound && p->w_bound) {
    // the mesh-static pooled transitions with BOUND weights (bsms_plan_bind_edge_weights): compact CSR + weights in slot
    // order (common.h) -- per output row one rowptr pair, then index and weight as two coalesced streams, then the rows:
    // two dependent round trips fewer than rows -> rowptr -> (xidx, widx) -> (xmap, w).  Same additions in the same order
    // (the dropped slots of the prolongation were skipped by a select before): bit-identical.
    a.addend = addend;
    if (aggregating) {
      a.rowptr = p->k_rowptr; a.xidx = p->k_src; a.w = p->k_w;
      a.n_out = (int32_t)p->Nk; a.x_bstride = p->N * D;
    } else {
      a.rowptr = p->p_rowptr; a.xidx = p->p_src; a.w = p->p_w;
      a.n_out = (int32_t)p->N; a.x_bstride = p->Nk * D;
    }
    a.out_bstride = int64_t(a.n_out) * D;
    return launch_rowsum(a, stream);
  }
  if (aggregating) {            // by target: fine x -> all rows or kept rows
    a.rowptr = p->rowptr; a.xidx = p->src; a.widx = p->perm;
    a.rows = pooled ? p->ids : nullptr;
    a.n_out = (int32_t)(pooled ? p->Nk : p->N);
    a.x_bstride = p->N * D;
  } else {                      // by source: (un-pooled coarse | fine) x -> all fine rows
    a.rowptr = p->t_rowptr; a.xidx = p->t_dst; a.widx = p->t_eid;
    a.xmap = pooled ? p->inv : nullptr;
    a.n_out = (int32_t)p->N;
    a.x_bstride = (pooled ? p->Nk : p->N) * D;
  }
  a.out_bstride = int64_t(a.n_out) * D;
  a.addend = addend;
  return launch_rowsum(a, stream);
}

extern "C" int bsms_scatter_rows(const float* h, int64_t B, int64_t Nk, int64_t D, const int64_t* idx, int64_t N,
                                 float* out, bsms_stream_t stream) {
  BSMS_REQUIRE(out && (h || Nk == 0) && (idx || Nk == 0), BSMS_E_INVALID_ARG, "scatter_rows: null argument");
  BSMS_REQUIRE(B >= 0 && Nk >= 0 && N >= 0 && D >= 1, BSMS_E_SHAPE, "scatter_rows: bad shape");
  hipStream_t s = as_stream(stream);
  if (B * N * D > 0) BSMS_HIP_CHECK(hipMemsetAsync(out, 0, size_t(B) * N * D * sizeof(float), s));
  const int per_row = (D % 4 == 0) ? int(D / 4) : int(D);
  const int64_t total = B * Nk * per_row;
  if (total == 0) return BSMS_OK;
  hipLaunchKernelGGL((k_copy_rows<int64_t>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, h, out,
                     (const int64_t*)nullptr, idx, Nk * D, N * D, (int32_t)Nk, (int32_t)B, (int32_t)D);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

extern "C" int bsms_gather_rows(const float* x, int64_t B, int64_t N, int64_t D, const int64_t* idx, int64_t Nk,
                                float* out, bsms_stream_t stream) {
  BSMS_REQUIRE((out && x && idx) || Nk == 0, BSMS_E_INVALID_ARG, "gather_rows: null argument");
  BSMS_REQUIRE(B >= 0 && Nk >= 0 && N >= 0 && D >= 1, BSMS_E_SHAPE, "gather_rows: bad shape");
  const int per_row = (D % 4 == 0) ? int(D / 4) : int(D);
  const int64_t total = B * Nk * per_row;
  if (total == 0) return BSMS_OK;
  hipLaunchKernelGGL((k_copy_rows<int64_t>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), x,
                     out, idx, (const int64_t*)nullptr, N * D, Nk * D, (int32_t)Nk, (int32_t)B, (int32_t)D);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
