// Weight-gradient kernels: dW = G^T A with the reduction over (up to millions of) edge / node rows.
//
// k_wgrad: split-K f32-MFMA GEMM.  A workgroup owns one 128x128 block of dW and a contiguous slab of
// rows; 32-row chunks of G and A stream HBM -> registers -> LDS (2-deep ring, row-major exactly as
// they sit in HBM, so the loads are full 512-byte bursts) and feed v_mfma_f32_32x32x2_f32 with
// conflict-free ds_read_b32 (lanes run along the feature axis for both operands).  Partial blocks go
// to a workspace and are summed in slab order by k_reduce (deterministic; no float atomics).  Several
// layers' gradients are batched into one launch so the small coarse levels still fill the chip.
// The bias gradient (column sums of G) rides along for free from the LDS tile.
#include "chain.h"

using namespace bsms;

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int TB = 128;  // dW block edge
constexpr int RC = 32;   // rows per chunk

struct WgradTable {
  int njobs, D, nblk;
  int rows_per_wg[kMaxWgradJobs];
  int nsplit[kMaxWgradJobs];
  int first_tile[kMaxWgradJobs + 1];  // tile = ((split * nblk) + bi) * nblk + bj, offset by first_tile[job]
  WgradJob job[kMaxWgradJobs];
  float* partials;   // [tiles][TB*TB]
  float* colsums;    // [tiles][TB]
};

__global__ __launch_bounds__(256) void k_wgrad(WgradTable tab) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][G|A][RC][TB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  int j = 0;
  while (j + 1 < tab.njobs && int(blockIdx.x) >= tab.first_tile[j + 1]) ++j;
  const WgradJob job = tab.job[j];
  const int local = blockIdx.x - tab.first_tile[j];
  const int bj = local % tab.nblk, bi = (local / tab.nblk) % tab.nblk, split = local / (tab.nblk * tab.nblk);
  const int D = tab.D;
  const int64_t r0 = int64_t(split) * tab.rows_per_wg[j];
  const int64_t r1 = min(job.R, r0 + tab.rows_per_wg[j]);
  const int nchunk = int((r1 - r0 + RC - 1) / RC);
  const int n0 = bi * TB, k0 = bj * TB;

  // staging: each thread moves 4 float4 of G and 4 of A per chunk (row = tid/32 + 8 i, col4 = tid%32)
  const int srow = tid >> 5, scol = (tid & 31) * 4;
  float4 sg[4], sa[4];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + int64_t(chunk) * RC + srow + 8 * i;
      const bool rv = r < r1;
      sg[i] = (rv && n0 + scol < D) ? *reinterpret_cast<const float4*>(job.G + r * job.ldg + n0 + scol)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      sa[i] = (rv && k0 + scol < D) ? *reinterpret_cast<const float4*>(job.A + r * job.lda + k0 + scol)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&](int buf) {
    float* g = lds + buf * (2 * RC * TB);
    float* a = g + RC * TB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(g + (srow + 8 * i) * TB + scol) = sg[i];
      *reinterpret_cast<float4*>(a + (srow + 8 * i) * TB + scol) = sa[i];
    }
  };

  const int wr = wave >> 1, wc = wave & 1;  // wave owns dW rows [64 wr, +64) x cols [64 wc, +64)
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
  float csum = 0.f;  // thread tid < TB sums column tid of the G tile

  if (nchunk > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const float* g = lds + (c & 1) * (2 * RC * TB);
    const float* a = g + RC * TB;
    if (c + 1 < nchunk) fetch(c + 1);
#pragma unroll
    for (int s = 0; s < RC / 2; ++s) {
      const int rr = 2 * s + hh;
      const float a0 = g[rr * TB + 64 * wr + l31], a1 = g[rr * TB + 64 * wr + 32 + l31];
      const float b0 = a[rr * TB + 64 * wc + l31], b1 = a[rr * TB + 64 * wc + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (job.db && bj == 0 && tid < TB) {
#pragma unroll 8
      for (int rr = 0; rr < RC; ++rr) csum += g[rr * TB + tid];
    }
    if (c + 1 < nchunk) stash((c + 1) & 1);
    __syncthreads();
  }

  float* part = tab.partials + int64_t(blockIdx.x) * (TB * TB);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
        part[n * TB + 64 * wc + 32 * jj + l31] = acc[i][jj][r];
      }
  if (job.db && bj == 0 && tid < TB) tab.colsums[int64_t(blockIdx.x) * TB + tid] = csum;
}

// dW[n][col0+k] = sum over slabs (in order) of the partial blocks; db likewise.
__global__ __launch_bounds__(256) void k_wgrad_reduce(WgradTable tab) {
  const int j = blockIdx.y;
  const WgradJob job = tab.job[j];
  const int D = tab.D, nblk = tab.nblk;
  const int total = D * D + (job.db ? D : 0);
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    if (o < D * D) {
      const int n = o / D, k = o % D;
      const int bi = n / TB, bj = k / TB;
      float s = 0.f;
      for (int sp = 0; sp < tab.nsplit[j]; ++sp) {
        const int tile = tab.first_tile[j] + (sp * nblk + bi) * nblk + bj;
        s += tab.partials[int64_t(tile) * (TB * TB) + (n % TB) * TB + (k % TB)];
      }
      job.dW[int64_t(n) * job.ldw + job.col0 + k] = s;
    } else {
      const int n = o - D * D, bi = n / TB;
      float s = 0.f;
      for (int sp = 0; sp < tab.nsplit[j]; ++sp) {
        const int tile = tab.first_tile[j] + (sp * nblk + bi) * nblk;
        s += tab.colsums[int64_t(tile) * TB + (n % TB)];
      }
      job.db[n] = s;
    }
  }
}

constexpr int kMaxTiles = 1024;

// ------------------------------------------------------------------ narrow-side weight gradients
constexpr int SW_MAXS = 8;
constexpr int SW_WGS = 256;

__global__ __launch_bounds__(256) void k_small_wgrad(SmallWgradArgs a, float* part /* [WGS][(S+2)][D] */,
                                                     int rows_per_wg) {
  // thread = (feature f, row lane rl); row lanes split the slab, combined through LDS at the end
  __shared__ float red[256 * (SW_MAXS + 1)];
  __shared__ float s_sh[8][SW_MAXS];  // per row-lane broadcast of the narrow row
  const int D = a.D, tid = threadIdx.x;
  const int nrl = 256 / D > 0 ? 256 / D : 1;
  const int f = tid % D, rl = tid / D;
  const bool active = tid < nrl * D;
  const int S = a.S_cols;
  float acc[SW_MAXS + 1];
#pragma unroll
  for (int s = 0; s <= SW_MAXS; ++s) acc[s] = 0.f;
  const int64_t r0 = int64_t(blockIdx.x) * rows_per_wg, r1 = min(a.R, r0 + rows_per_wg);
  for (int64_t base = r0; base < r1; base += nrl) {
    const int64_t r = base + rl;
    const bool rv = active && r < r1;
    // narrow row -> LDS (one thread per (row lane, s))
    if (active && f < S && r < r1) {
      float v;
      if (a.S) {
        v = a.S[r * S + f];
      } else {  // fiber [pos_i - pos_j, |pos_i - pos_j|]  (ops/basic.py:83-85)
        const int b = int(r / a.E), q = int(r - int64_t(b) * a.E);
        const int i = a.src[q], jn = a.dst[q];
        const float* pb = a.pos + b * a.pos_bstride;
        if (f < a.p) {
          v = pb[int64_t(i) * a.p + f] - pb[int64_t(jn) * a.p + f];
        } else {
          float n2 = 0.f;
          for (int c = 0; c < a.p; ++c) {
            const float rel = pb[int64_t(i) * a.p + c] - pb[int64_t(jn) * a.p + c];
            n2 = fmaf(rel, rel, n2);
          }
          v = sqrtf(n2);
        }
      }
      s_sh[rl][f] = v;
    }
    __syncthreads();
    if (rv) {
      const float g = a.G[r * D + f];
#pragma unroll
      for (int s = 0; s < SW_MAXS; ++s)
        if (s < S) acc[s] = fmaf(g, s_sh[rl][s], acc[s]);
      acc[SW_MAXS] += g;
    }
    __syncthreads();
  }
  // combine row lanes
#pragma unroll
  for (int s = 0; s <= SW_MAXS; ++s) red[s * 256 + tid] = active ? acc[s] : 0.f;
  __syncthreads();
  if (tid < D) {
    float* out = part + int64_t(blockIdx.x) * (SW_MAXS + 1) * D;
    for (int s = 0; s <= SW_MAXS; ++s) {
      float v = 0.f;
      for (int l = 0; l < nrl; ++l) v += red[s * 256 + l * D + tid];
      out[s * D + tid] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_small_reduce(SmallWgradArgs a, const float* part, int nwg) {
  const int D = a.D, S = a.S_cols;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= (SW_MAXS + 1) * D) return;
  const int s = o / D, f = o % D;
  if (s < SW_MAXS && s >= S) return;
  float v = 0.f;
  for (int w = 0; w < nwg; ++w) v += part[(int64_t(w) * (SW_MAXS + 1) + s) * D + f];
  if (s < S) a.out[s * a.os + f * a.of] = v;
  else if (a.colsum) a.colsum[f] = v;
}

// colsum of the narrow matrix itself (decoder output bias): tiny, one workgroup
__global__ __launch_bounds__(256) void k_colsum_small(const float* S, int64_t R, int cols, float* out) {
  __shared__ float red[256];
  for (int c = 0; c < cols; ++c) {
    float v = 0.f;
    for (int64_t r = threadIdx.x; r < R; r += 256) v += S[r * cols + c];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
    __syncthreads();
  }
}

}  // namespace

namespace bsms {

size_t wgrad_work_bytes(int D, int njobs) {
  (void)D;
  (void)njobs;
  return size_t(kMaxTiles) * (TB * TB + TB) * sizeof(float);
}

int launch_wgrad(int D, const WgradJob* jobs, int njobs, void* work, hipStream_t s) {
  BSMS_REQUIRE(njobs >= 0 && njobs <= kMaxWgradJobs, BSMS_E_INVALID_ARG, "wgrad: %d jobs (max %d)", njobs, kMaxWgradJobs);
  BSMS_REQUIRE(D % 4 == 0 && D <= 256, BSMS_E_UNSUPPORTED, "wgrad: D=%d", D);
  if (njobs == 0) return BSMS_OK;
  WgradTable tab{};
  tab.njobs = njobs;
  tab.D = D;
  tab.nblk = (int)ceil_div(D, TB);
  const int blocks = tab.nblk * tab.nblk;
  int64_t total_rows = 0;
  for (int j = 0; j < njobs; ++j) total_rows += jobs[j].R;
  // aim for ~768 workgroups over all jobs; slabs are multiples of RC rows, at least 128
  int64_t rows_per = std::max<int64_t>(128, ceil_div(total_rows * blocks, 768));
  rows_per = ceil_div(rows_per, RC) * RC;
  for (;;) {
    int64_t tiles = 0;
    for (int j = 0; j < njobs; ++j) tiles += std::max<int64_t>(1, ceil_div(jobs[j].R, rows_per)) * blocks;
    if (tiles <= kMaxTiles) break;
    rows_per *= 2;
  }
  int first = 0;
  for (int j = 0; j < njobs; ++j) {
    tab.job[j] = jobs[j];
    tab.rows_per_wg[j] = (int)rows_per;
    tab.nsplit[j] = (int)std::max<int64_t>(1, ceil_div(jobs[j].R, rows_per));
    tab.first_tile[j] = first;
    first += tab.nsplit[j] * blocks;
  }
  tab.first_tile[njobs] = first;
  tab.partials = reinterpret_cast<float*>(work);
  tab.colsums = tab.partials + size_t(kMaxTiles) * TB * TB;
  const size_t lds = size_t(2) * 2 * RC * TB * sizeof(float);
  hipLaunchKernelGGL(k_wgrad, dim3(first), dim3(256), lds, s, tab);
  BSMS_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)std::min<int64_t>(ceil_div(D * D + D, 256), 64), njobs), dim3(256), 0, s, tab);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

size_t small_wgrad_work_bytes(int D) { return size_t(SW_WGS) * (SW_MAXS + 1) * D * sizeof(float); }

int launch_small_wgrad(const SmallWgradArgs& a, void* work, hipStream_t s) {
  BSMS_REQUIRE(a.S_cols >= 1 && a.S_cols <= SW_MAXS, BSMS_E_UNSUPPORTED, "small_wgrad: narrow width %d (max %d)", a.S_cols,
               SW_MAXS);
  BSMS_REQUIRE(a.D >= 32 && a.D <= 256, BSMS_E_UNSUPPORTED, "small_wgrad: D=%d", a.D);
  float* part = reinterpret_cast<float*>(work);
  const int nrl = std::max(1, 256 / a.D);
  int64_t rows_per = std::max<int64_t>(64, ceil_div(a.R, SW_WGS));
  rows_per = ceil_div(rows_per, nrl) * nrl;
  const int nwg = (int)std::max<int64_t>(1, ceil_div(a.R, rows_per));
  hipLaunchKernelGGL(k_small_wgrad, dim3(nwg), dim3(256), 0, s, a, part, (int)rows_per);
  BSMS_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_small_reduce, dim3((unsigned)ceil_div((SW_MAXS + 1) * a.D, 256)), dim3(256), 0, s, a, (const float*)part, nwg);
  BSMS_LAUNCH_CHECK();
  if (a.colsum_S && a.S) {
    hipLaunchKernelGGL(k_colsum_small, dim3(1), dim3(256), 0, s, a.S, a.R, a.S_cols, a.colsum_S);
    BSMS_LAUNCH_CHECK();
  }
  return BSMS_OK;
}

}  // namespace bsms
