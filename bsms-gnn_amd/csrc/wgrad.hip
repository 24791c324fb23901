// Weight-gradient kernels: dW = G^T A with the reduction over (up to millions of) edge / node rows.
//
// k_wgrad: split-K GEMM on v_mfma_f32_16x16x32_bf16 with the same exact three-way bf16 split as the chain kernels
// (chain.h): both operands are fp32 in HBM, split ONCE per element while they are staged into LDS, and every
// product is accumulated in fp32 as six bf16 partial products.  The reduction index (rows) is the MFMA's K, so an
// operand fragment is 8 consecutive ROWS of one column: a lane stages exactly that -- one column of G and one of A
// for 8 rows (4-byte loads, 16 lanes = one 64-byte row segment), splits its 16 values and writes three 16-byte
// fragments per matrix, already in MFMA order ([plane][column][row group], a wave's reads and writes are
// contiguous 1 KB: conflict-free).  A workgroup owns one 128x128 block of dW and a contiguous slab of rows, 32 rows
// per chunk, next chunk's loads in flight during the MFMAs; 48 KB of LDS -> two workgroups per CU alternate
// staging and MFMA phases.  At this rate the kernel is HBM bound (two fp32 streams, each read once).
// Partial blocks go to a workspace and are summed in slab order by k_wgrad_reduce (deterministic; no float
// atomics).  Several layers' gradients are batched into one launch so the small coarse levels still fill the
// chip.  The bias gradient (column sums of G) is accumulated in fp32 by the staging lanes.
#include "chain.h"

using namespace bsms;

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int TB = 128;  // dW block edge
constexpr int RC = 32;   // rows per chunk (= K of one MFMA)

struct WgradTable {
  int njobs, D, nblk;
  int rows_per_wg[kMaxWgradJobs];
  int nsplit[kMaxWgradJobs];
  int first_tile[kMaxWgradJobs + 1];  // tile = ((split * nblk) + bi) * nblk + bj, offset by first_tile[job]
  WgradJob job[kMaxWgradJobs];
  float* partials;   // [tiles][TB*TB]
  float* colsums;    // [tiles][TB]
};

__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {  // exact: x = hi + mid + lo
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid));
}

// 8 fp32 (consecutive rows of one column) -> three fragments of 8 bf16
__device__ __forceinline__ void split_column(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3(v[2 * q], h0, m0, l0);
    split3(v[2 * q + 1], h1, m1, l1);
    h[q] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    m[q] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    l[q] = __builtin_amdgcn_perm(l1, l0, 0x07060302u);
  }
}

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_wgrad(WgradTable tab) {
  extern __shared__ __attribute__((aligned(16))) u32x4 frag[];  // [G|A][plane][column 0..127][row group 0..3]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
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

  // staging: this lane owns column 16 wave + m of the G block and of the A block, rows 8 g .. 8 g + 7 of a chunk
  const int col = 16 * wave + m;
  const bool vg = n0 + col < D, va = k0 + col < D;
  const float* gcol = job.G + n0 + col;
  const float* acol = job.A + k0 + col;
  float gv[8], av[8];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t r = r0 + int64_t(chunk) * RC + 8 * g + i;
      const bool rv = r < r1;
      gv[i] = (rv && vg) ? gcol[r * job.ldg] : 0.f;
      av[i] = (rv && va) ? acol[r * job.lda] : 0.f;
    }
  };
  const bool want_db = job.db && bj == 0;
  float csum = 0.f;  // partial column sum of G over this lane's rows
  auto stash = [&]() {
    u32x4 h, mm, l;
    split_column(gv, h, mm, l);
    frag[(0 * TB + col) * 4 + g] = h;
    frag[(1 * TB + col) * 4 + g] = mm;
    frag[(2 * TB + col) * 4 + g] = l;
    if (want_db) csum += ((gv[0] + gv[1]) + (gv[2] + gv[3])) + ((gv[4] + gv[5]) + (gv[6] + gv[7]));
    split_column(av, h, mm, l);
    frag[(3 * TB + col) * 4 + g] = h;
    frag[(4 * TB + col) * 4 + g] = mm;
    frag[(5 * TB + col) * 4 + g] = l;
  };

  // 8 waves: wave owns dW rows [32 wr, +32) x cols [64 wc, +64) = 2 x 4 MFMA blocks
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nchunk > 0) fetch(0);
  for (int c = 0; c < nchunk; ++c) {
    stash();                          // split + store this chunk's fragments (waits for its loads)
    if (c + 1 < nchunk) fetch(c + 1);  // next chunk's rows fly during the MFMAs
    wg_barrier();
    u32x4 gh[2], gm[2], gl[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int o = (32 * wr + 16 * a + m) * 4 + g;
      gh[a] = frag[0 * TB * 4 + o]; gm[a] = frag[1 * TB * 4 + o]; gl[a] = frag[2 * TB * 4 + o];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int o = (64 * wc + 16 * b + m) * 4 + g;
      const u32x4 ah = frag[3 * TB * 4 + o], am = frag[4 * TB * 4 + o], al = frag[5 * TB * 4 + o];
      acc[0][b] = mma(gl[0], ah, acc[0][b]);
      acc[1][b] = mma(gl[1], ah, acc[1][b]);
      acc[0][b] = mma(gh[0], al, acc[0][b]);
      acc[1][b] = mma(gh[1], al, acc[1][b]);
      acc[0][b] = mma(gm[0], am, acc[0][b]);
      acc[1][b] = mma(gm[1], am, acc[1][b]);
      acc[0][b] = mma(gm[0], ah, acc[0][b]);
      acc[1][b] = mma(gm[1], ah, acc[1][b]);
      acc[0][b] = mma(gh[0], am, acc[0][b]);
      acc[1][b] = mma(gh[1], am, acc[1][b]);
      acc[0][b] = mma(gh[0], ah, acc[0][b]);
      acc[1][b] = mma(gh[1], ah, acc[1][b]);
    }
    wg_barrier();                     // everyone is done reading before the next stash overwrites
  }

  // D[row = 4 g + r][col = m] of block (a, b) = dW[32 wr + 16 a + 4 g + r][64 wc + 16 b + m]
  float* part = tab.partials + int64_t(blockIdx.x) * (TB * TB);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(32 * wr + 16 * a + 4 * g + r) * TB + 64 * wc + 16 * b + m] = acc[a][b][r];
  if (want_db) {
    csum += __shfl_xor(csum, 16, 64);
    csum += __shfl_xor(csum, 32, 64);
    if (g == 0) tab.colsums[int64_t(blockIdx.x) * TB + col] = csum;
  }
}

// dW[n][col0+k] = sum over slabs of the partial blocks; db likewise.  A block owns 64 float4 outputs; its 4
// "slab lanes" each sum every 4th slab (independent 16-byte loads in flight), then combine in a fixed order
// through LDS, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void k_wgrad_reduce(WgradTable tab) {
  __shared__ float4 red[256];
  const int j = blockIdx.y;
  const WgradJob job = tab.job[j];
  const int D = tab.D, nblk = tab.nblk, ns = tab.nsplit[j];
  const int o4 = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  const int nmat = D * D / 4, nall = nmat + (job.db ? D / 4 : 0);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (o4 < nall) {
    const float* base;
    int64_t stride;
    if (o4 < nmat) {
      const int n = (o4 * 4) / D, k = (o4 * 4) % D;
      const int bi = n / TB, bj = k / TB;
      base = tab.partials + int64_t(tab.first_tile[j] + bi * nblk + bj) * (TB * TB) + (n % TB) * TB + (k % TB);
      stride = int64_t(nblk) * nblk * (TB * TB);
    } else {
      const int n = (o4 - nmat) * 4, bi = n / TB;
      base = tab.colsums + int64_t(tab.first_tile[j] + bi * nblk) * TB + (n % TB);
      stride = int64_t(nblk) * nblk * TB;
    }
    for (int sp = sl; sp < ns; sp += 4) {
      const float4 v = *reinterpret_cast<const float4*>(base + sp * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0 && o4 < nall) {
#pragma unroll
    for (int l = 1; l < 4; ++l) {
      const float4 v = red[l * 64 + threadIdx.x];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (o4 < nmat) {
      const int n = (o4 * 4) / D, k = (o4 * 4) % D;
      float* dst = job.dW + int64_t(n) * job.ldw + job.col0 + k;
      if (((job.ldw | job.col0) & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = acc;
      } else {  // sub-block of a wider matrix (edge W0: ld = 2D+p+1, column offset p+1): only 4-byte aligned
        dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w;
      }
    } else {
      *reinterpret_cast<float4*>(job.db + (o4 - nmat) * 4) = acc;
    }
  }
}

constexpr int kMaxTiles = 1024;

// ------------------------------------------------------------------ narrow-side weight gradients
// out[s][f] = sum_r G[r][f] * S[r][s] with S at most 8 columns wide (the fiber columns of the first edge Linear,
// the encoder's first and the decoder's last Linear) plus colsum(G).  HBM bound on G: a workgroup streams a slab
// of rows with 16-byte loads (a row of D floats = one burst of D/4 lanes), 256/(D/4) rows in flight per pass;
// the narrow row is broadcast.  Per-workgroup partials are combined in fixed order by k_small_reduce.
constexpr int SW_MAXS = 8;
constexpr int SW_WGS = 1024;

__device__ __forceinline__ void small_row(const SmallWgradArgs& a, int64_t r, float (&sv)[SW_MAXS]) {
  if (a.S) {
#pragma unroll
    for (int s = 0; s < SW_MAXS; ++s) sv[s] = s < a.S_cols ? a.S[r * a.S_cols + s] : 0.f;
  } else {  // fiber [pos_i - pos_j, |pos_i - pos_j|]  (ops/basic.py:83-85)
    const int b = int(r / a.E), q = int(r - int64_t(b) * a.E);
    const int i = a.src[q], jn = a.dst[q];
    const float* pb = a.pos + b * a.pos_bstride;
    float n2 = 0.f;
#pragma unroll
    for (int c = 0; c < SW_MAXS; ++c) {
      float rel = 0.f;
      if (c < a.p) {
        rel = pb[int64_t(i) * a.p + c] - pb[int64_t(jn) * a.p + c];
        n2 = fmaf(rel, rel, n2);
      }
      sv[c] = rel;
    }
    const float nrm = sqrtf(n2);
#pragma unroll
    for (int c = 0; c < SW_MAXS; ++c)
      if (c == a.p) sv[c] = nrm;
  }
}

template <int NS>  // NS = number of narrow columns actually accumulated (compile-time for register allocation)
__global__ __launch_bounds__(256) void k_small_wgrad(SmallWgradArgs a, float* part /* [nwg][SW_MAXS+1][D] */,
                                                     int rows_per_wg) {
  __shared__ float4 red[256];
  const int D = a.D, tid = threadIdx.x, d4 = D >> 2;
  const int nrl = 256 / d4;                  // rows in flight per pass (D=128: 8)
  const int c4 = tid % d4, rl = tid / d4;
  const bool active = rl < nrl;
  float4 acc[NS + 1];
#pragma unroll
  for (int s = 0; s <= NS; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t r0 = int64_t(blockIdx.x) * rows_per_wg, r1 = min(a.R, r0 + rows_per_wg);
  if (active) {
    for (int64_t r = r0 + rl; r < r1; r += nrl) {
      const float4 g = *reinterpret_cast<const float4*>(a.G + r * D + c4 * 4);
      float sv[SW_MAXS];
      small_row(a, r, sv);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        acc[s].x = fmaf(g.x, sv[s], acc[s].x); acc[s].y = fmaf(g.y, sv[s], acc[s].y);
        acc[s].z = fmaf(g.z, sv[s], acc[s].z); acc[s].w = fmaf(g.w, sv[s], acc[s].w);
      }
      acc[NS].x += g.x; acc[NS].y += g.y; acc[NS].z += g.z; acc[NS].w += g.w;
    }
  }
  float* out = part + int64_t(blockIdx.x) * (SW_MAXS + 1) * D;
#pragma unroll
  for (int s = 0; s <= NS; ++s) {  // combine the row lanes in fixed order
    red[tid] = active ? acc[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tid < d4) {
      float4 v = red[tid];
      for (int l = 1; l < nrl; ++l) {
        const float4 w = red[l * d4 + tid];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      *reinterpret_cast<float4*>(out + (s == NS ? SW_MAXS : s) * D + tid * 4) = v;
    }
    __syncthreads();
  }
}

// one block per 8 output floats; 32 partial-lanes each sum every 32nd workgroup partial, fixed-order combine
__global__ __launch_bounds__(256) void k_small_reduce(SmallWgradArgs a, const float* part, int nwg) {
  __shared__ float red[256];
  const int D = a.D, S = a.S_cols;
  const int o = blockIdx.x * 8 + (threadIdx.x & 7), pl = threadIdx.x >> 3;
  const int s = o / D, f = o % D;
  const bool valid = o < (SW_MAXS + 1) * D && (s < S || s == SW_MAXS);
  float v = 0.f;
  if (valid)
    for (int w = pl; w < nwg; w += 32) v += part[(int64_t(w) * (SW_MAXS + 1) + s) * D + f];
  red[threadIdx.x] = v;
  __syncthreads();
  if (pl == 0 && valid) {
    for (int l = 1; l < 32; ++l) v += red[l * 8 + threadIdx.x];
    if (s < S) a.out[s * a.os + f * a.of] = v;
    else if (a.colsum) a.colsum[f] = v;
  }
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
  const size_t lds = size_t(6) * TB * 4 * sizeof(u32x4);   // 48 KB: [G|A][3 planes][128 columns][4 row groups] x 16 B
  hipLaunchKernelGGL(k_wgrad, dim3(first), dim3(512), lds, s, tab);
  BSMS_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)ceil_div((D * D + D) / 4, 64), njobs), dim3(256), 0, s, tab);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

size_t small_wgrad_work_bytes(int D) { return size_t(SW_WGS) * (SW_MAXS + 1) * D * sizeof(float); }

int launch_small_wgrad(const SmallWgradArgs& a, void* work, hipStream_t s) {
  BSMS_REQUIRE(a.S_cols >= 1 && a.S_cols <= SW_MAXS, BSMS_E_UNSUPPORTED, "small_wgrad: narrow width %d (max %d)", a.S_cols,
               SW_MAXS);
  BSMS_REQUIRE(a.D >= 32 && a.D <= 256 && a.D % 4 == 0, BSMS_E_UNSUPPORTED, "small_wgrad: D=%d", a.D);
  BSMS_REQUIRE(a.S != nullptr || a.p + 1 == a.S_cols, BSMS_E_INVALID_ARG, "small_wgrad: fiber mode needs S_cols = p+1");
  float* part = reinterpret_cast<float*>(work);
  const int nrl = 256 / (a.D / 4);
  int64_t rows_per = std::max<int64_t>(4 * nrl, ceil_div(a.R, SW_WGS));
  rows_per = ceil_div(rows_per, nrl) * nrl;
  const int nwg = (int)std::max<int64_t>(1, ceil_div(a.R, rows_per));
#define BSMS_SW(NS)                                                                                   \
  case NS:                                                                                            \
    hipLaunchKernelGGL((k_small_wgrad<NS>), dim3(nwg), dim3(256), 0, s, a, part, (int)rows_per); \
    break
  switch (a.S_cols) {
    BSMS_SW(1); BSMS_SW(2); BSMS_SW(3); BSMS_SW(4); BSMS_SW(5); BSMS_SW(6); BSMS_SW(7); BSMS_SW(8);
  }
#undef BSMS_SW
  BSMS_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_small_reduce, dim3((unsigned)ceil_div((SW_MAXS + 1) * a.D, 8)), dim3(256), 0, s, a, (const float*)part, nwg);
  BSMS_LAUNCH_CHECK();
  if (a.colsum_S && a.S) {
    hipLaunchKernelGGL(k_colsum_small, dim3(1), dim3(256), 0, s, a.S, a.R, a.S_cols, a.colsum_S);
    BSMS_LAUNCH_CHECK();
  }
  return BSMS_OK;
}

}  // namespace bsms
