// Weight-gradient kernels: dW = G^T A with the reduction over (up to millions of) edge / node rows.
//
// k_wgrad: split-K GEMM on the 16x16x32 matrix instructions with fp32 operands in HBM, split ONCE per element while they
// are staged into LDS.  Two arithmetics (WgradJob::g_bound / a_bound, chain.h):
//   H2   both operands come with a magnitude bound: fp16 x 2 pieces with one power-of-two scale per TENSOR (the
//        reduction index is the row, so the chain kernels' per-row scales cannot be used) and three partial products
//        h l + l h + h h per fragment pair -- the arithmetic of the chain kernels (chain.h);
//   BF3  no bounds: the exact three-way bf16 split of rounds 1-2, six partial products (needs no range information).
// A workgroup owns one 128x128 block of dW and a contiguous slab of rows, 32 rows per chunk:
//   stage   1024 threads load the chunk of G and of A row-major (16-byte loads, full 512-byte row bursts, three
//           chunks ahead in registers), split every value into its 16-bit pieces (H2: h, l; BF3: hi, mid, lo) and store
//           one row-major plane per piece and matrix (row pitch 288 B so that 8 consecutive rows hit disjoint banks);
//   MFMA    the reduction index (rows) is the MFMA's K, i.e. an operand fragment is a COLUMN of 8 rows per lane:
//           ds_read_b64_tr_b16 delivers exactly that from the row-major planes (a 16-lane group transposes a
//           4-row x 16-column block), two reads per fragment, no bank conflicts.
// The planes are double buffered (108 KB, one 16-wave workgroup per CU, one barrier per chunk): a wave stages chunk
// c+1, then multiplies chunk c, and the four waves of a SIMD overlap each other's phases.  At this rate the kernel is HBM bound
// (two fp32 streams, each read once).  Partial blocks go to a workspace and are summed in slab order by
// k_wgrad_reduce (deterministic; no float atomics).  Several layers' gradients are batched into one launch so the
// small coarse levels still fill the chip.  The bias gradient (column sums of G) is accumulated in fp32 by the
// staging threads.
// D = 256 (round 6): four 128 x 128 blocks per slab would read -- and split -- every row of G and A twice.  Launches of edge-level
// size take k_wgrad_wide / k_wgrad_bf64_wide (a workgroup owns 128 x 256 of dW: A staged once), and in every tiling the workgroups
// of one slab are placed on the same XCD (tile_of / wide_tile) so that the remaining duplicate reads are L2 hits: 1.04 x the
// algorithmic bytes from HBM (profiles/r06_step_pmc_surface.txt).
#include "chain.h"
#include <cstdlib>
#include <type_traits>

using namespace bsms;

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int TB = 128;    // dW block edge
constexpr int RC = 32;     // rows per chunk (= K of one MFMA)
constexpr int LROW = 144;  // bf16 per LDS row: 128 columns + 16 pad (288 B: 8 banks further per row)
constexpr int PLANE = RC * LROW;

struct WgradTable {
  int njobs, D, nblk;
  int rows_per_wg[kMaxWgradJobs];
  int nsplit[kMaxWgradJobs];
  int first_tile[kMaxWgradJobs + 1];  // tile = ((split * nblk) + bi) * nblk + bj, offset by first_tile[job]
  WgradJob job[kMaxWgradJobs];
  float* partials;   // [tiles][TB*TB]
  float* colsums;    // [tiles][TB]
  unsigned long long* timing;  // experiments only: per-workgroup s_memtime stamps (null in production)
  int wide;          // k_wgrad_wide launch (D = 256): a tile is 128 rows x ALL 256 columns of dW, tile = split * nblk + bi
};

__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {  // exact: x = hi + mid + lo
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid));
}

// 4 fp32 (consecutive columns of one row) -> 4 bf16 per plane
__device__ __forceinline__ void split_quad(const f32x4& v, u32x2& h, u32x2& m, u32x2& l) {
  unsigned h0, m0, l0, h1, m1, l1;
  split3(v[0], h0, m0, l0);
  split3(v[1], h1, m1, l1);
  h[0] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  m[0] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
  l[0] = __builtin_amdgcn_perm(l1, l0, 0x07060302u);
  split3(v[2], h0, m0, l0);
  split3(v[3], h1, m1, l1);
  h[1] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  m[1] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
  l[1] = __builtin_amdgcn_perm(l1, l0, 0x07060302u);
}

// fp16 pieces of s * x, two elements per dword (chain.hip: split_h2)
__device__ __forceinline__ void split_h2(float x0, float x1, float s, unsigned& h, unsigned& l) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ void split_quad_h2(const f32x4& v, float s, u32x2& h, u32x2& l) {
  unsigned h0, l0, h1, l1;
  split_h2(v[0], v[1], s, h0, l0);
  split_h2(v[2], v[3], s, h1, l1);
  h = u32x2{h0, h1};
  l = u32x2{l0, l1};
}
// biased exponent E of a bound (clamped so that 2^(141 - E) is a normal float); bound * 2^(141 - E) lies in [2^14, 2^15):
// the largest piece stays below fp16's 65504, and two octaves more of the tensor's range keep a normal low piece than with the
// chain kernels' [2^12, 2^13) (there the headroom covers the row sums of the LayerNorm; here nothing is added before the MFMA)
__device__ __forceinline__ int bound_exp(float b) {
  int E = int(__float_as_uint(b) >> 23) & 0xff;
  return E < 14 ? 14 : E;
}

// MFMA operand of one 16-column block from a row-major plane: lane (c = lane & 15, q = lane >> 4) receives column
// `col0 + c`, rows 4q..4q+3 (slots 0-3) and 16+4q..16+4q+3 (slots 4-7).  ds_read_b64_tr_b16: within a 16-lane group
// lane i' supplies the 8 bytes at its address and lane i receives element (i & 3) of suppliers 4k + (i >> 2), k = 0..3
// (profiles/census/tr_test.hip), so supplier i' points at row (i' >> 2), columns 4 (i' & 3)...
__device__ __forceinline__ bf16x8 column_fragment(const short* plane, int col0, int lane) {
  const int q = lane >> 4, ip = lane & 15;
  const short* p = plane + (4 * q + (ip >> 2)) * LROW + col0 + 4 * (ip & 3);
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * LROW));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma_h(const bf16x8& a, const bf16x8& b, f32x4 c) {   // the same 16-bit lanes read as fp16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Tile of a launch -> (slab, block row, block column).  At D = 256 (nblk = 2) the four workgroups of a slab read the same rows of G
// and A (two of them each half): they are placed EIGHT workgroup indices apart -- same XCD (workgroups go round-robin over the 8
// XCDs), same dispatch round -- so that three of the four reads of a chunk are served by that XCD's L2 (round 6; measured on the
// 128 x 256 tiles of k_wgrad_wide: 1.04 x the algorithmic bytes from HBM instead of 2 x).  nblk = 1: the identity.  `tile` is where
// k_wgrad_reduce looks for the block's partial sums (slab-major, as before).
struct TileRef { int split, bi, bj; int64_t tile; };
__device__ __forceinline__ TileRef tile_of(const WgradTable& tab, int j, int block) {
  const int local = block - tab.first_tile[j];
  TileRef t;
  if (tab.nblk == 2) {
    const int full = (tab.nsplit[j] >> 3) << 3;
    int blk;
    if (local < 4 * full) {
      const int r = local & 31;
      blk = r >> 3;
      t.split = ((local >> 5) << 3) + (r & 7);
    } else {
      const int l2 = local - 4 * full;
      t.split = full + (l2 >> 2);
      blk = l2 & 3;
    }
    t.bi = blk >> 1;
    t.bj = blk & 1;
  } else {
    t.bj = local % tab.nblk;
    t.bi = (local / tab.nblk) % tab.nblk;
    t.split = local / (tab.nblk * tab.nblk);
  }
  t.tile = tab.first_tile[j] + (int64_t(t.split) * tab.nblk + t.bi) * tab.nblk + t.bj;
  return t;
}

constexpr int WG_THREADS = 1024;  // 16 waves = 4 per SIMD, one workgroup per CU
constexpr int NPF = 3;            // chunks in flight in registers beyond the one being staged
static_assert(NPF == 3, "the step schedule in k_wgrad is written out for three register sets");

// TIMING: experiments (profiles/wgrad_timeline.py); the production instantiation carries no stamps
// JB: every job of the launch is a bf16 job (WgradJob::bf16); H2: every job carries operand bounds (fp16 x 2 pieces);
// compile-time switches so that every instantiation keeps its branch-free load schedule
template <bool TIMING, bool JB = false, bool H2 = false>
__global__ __launch_bounds__(WG_THREADS) void k_wgrad(WgradTable tab) {
  extern __shared__ __attribute__((aligned(16))) short planes[];  // [2 buffers][G|A][hi|mid|lo][32 rows][LROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int j = 0;
  while (j + 1 < tab.njobs && int(blockIdx.x) >= tab.first_tile[j + 1]) ++j;
  const WgradJob job = tab.job[j];
  const TileRef tr = tile_of(tab, j, blockIdx.x);
  const int bj = tr.bj, bi = tr.bi, split = tr.split;
  const int D = tab.D;
  const int64_t r0 = int64_t(split) * tab.rows_per_wg[j];
  const int64_t r1 = min(job.R, r0 + tab.rows_per_wg[j]);
  const int nchunk = int((r1 - r0 + RC - 1) / RC);
  const int n0 = bi * TB, k0 = bj * TB;
  // H2: one power-of-two scale per operand tensor from its magnitude bound
  int Eg = 139, Ea = 139;
  float sG = 1.f, sA = 1.f;
  if (H2) {   // bound = largest entry of the operand's bound slot (chain.h: kBoundWidth = 4 entries per thread)
    static_assert(kBoundWidth == 8 * WG_THREADS, "two float4 of each bound slot per thread");
    __shared__ unsigned bred[2][WG_THREADS / 64];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const u32x4* gs = reinterpret_cast<const u32x4*>(job.g_bound);
    const u32x4* as = reinterpret_cast<const u32x4*>(job.a_bound);
    const u32x4 gv = gs[tid], gw = gs[tid + WG_THREADS], av = as[tid], aw = as[tid + WG_THREADS];
    unsigned gm = max(max(max(gv[0], gv[1]), max(gv[2], gv[3])), max(max(gw[0], gw[1]), max(gw[2], gw[3])));   // non-negative floats order like integers
    unsigned am = max(max(max(av[0], av[1]), max(av[2], av[3])), max(max(aw[0], aw[1]), max(aw[2], aw[3])));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      gm = max(gm, (unsigned)__shfl_xor((int)gm, o, 64));
      am = max(am, (unsigned)__shfl_xor((int)am, o, 64));
    }
    if (lane == 0) { bred[0][wave] = gm; bred[1][wave] = am; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WG_THREADS / 64; ++w) { gm = max(gm, bred[0][w]); am = max(am, bred[1][w]); }
    Eg = bound_exp(__uint_as_float(gm) * job.g_mul);
    Ea = bound_exp(__uint_as_float(am) * job.a_mul);
    sG = __uint_as_float(unsigned(268 - Eg) << 23);
    sA = __uint_as_float(unsigned(268 - Ea) << 23);
  }

  // staging: 1024 threads move one 32-row chunk of G and of A, one float4 of each per thread (row = tid / 32,
  // columns 4 (tid % 32) ..): full 512-byte row bursts, NPF chunks ahead in registers
  const int srow = tid >> 5, scol = (tid & 31) * 4;
  const bool vg = n0 + scol < D, va = k0 + scol < D;
  // loads are unconditional (a predicated load costs a branch and a vmcnt(0)): out-of-range rows / columns read a
  // valid address and are zeroed when they are staged
  // bf16 job (bf16 precision of a GMP block: edge gradients and edge activations are stored as bf16): both matrices are
  // read as 4 x bf16 = 8 bytes per thread, staged into the `hi` plane as they are and multiplied with ONE product
  constexpr bool jb = JB;
  const float* gsrc = job.G + (vg ? n0 + scol : 0);
  const float* asrc = job.A + (va ? k0 + scol : 0);
  const unsigned short* gsrc16 = reinterpret_cast<const unsigned short*>(job.G) + (vg ? n0 + scol : 0);
  const unsigned short* asrc16 = reinterpret_cast<const unsigned short*>(job.A) + (va ? k0 + scol : 0);
  const int64_t rlast = r1 - 1;
  f32x4 sg[NPF], sa[NPF];
  bool live[NPF];
  auto fetch = [&](int chunk, int set) {   // chunks past the slab re-read its last rows (never staged)
    const int64_t r = r0 + int64_t(chunk) * RC + srow;
    live[set] = r < r1;
    const int64_t rc = r < r1 ? r : rlast;
    if (jb) {
      const u32x2 g2 = *reinterpret_cast<const u32x2*>(gsrc16 + rc * job.ldg), a2 = *reinterpret_cast<const u32x2*>(asrc16 + rc * job.lda);
      sg[set] = f32x4{__uint_as_float(g2[0]), __uint_as_float(g2[1]), 0.f, 0.f};
      sa[set] = f32x4{__uint_as_float(a2[0]), __uint_as_float(a2[1]), 0.f, 0.f};
    } else {
      sg[set] = *reinterpret_cast<const f32x4*>(gsrc + rc * job.ldg);
      sa[set] = *reinterpret_cast<const f32x4*>(asrc + rc * job.lda);
    }
  };
  const bool want_db = job.db && bj == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};  // column sums of G over this thread's rows
  auto stash = [&](int set, int buf) {
    short* dst = planes + buf * (6 * PLANE) + srow * LROW + scol;
    u32x2 h, m, l;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 gq = (live[set] && vg) ? sg[set] : zero, aq = (live[set] && va) ? sa[set] : zero;
    if (jb) {   // the raw bf16 quads go to the hi planes; mid / lo are not read for this job
      const unsigned g0 = __float_as_uint(gq[0]), g1 = __float_as_uint(gq[1]);
      *reinterpret_cast<u32x2*>(dst + 0 * PLANE) = u32x2{g0, g1};
      *reinterpret_cast<u32x2*>(dst + 3 * PLANE) = u32x2{__float_as_uint(aq[0]), __float_as_uint(aq[1])};
      if (want_db)
        csum += f32x4{__uint_as_float(g0 << 16), __uint_as_float(g0 & 0xffff0000u), __uint_as_float(g1 << 16), __uint_as_float(g1 & 0xffff0000u)};
      return;
    }
    if (H2) {   // planes 0 / 1: G pieces h / l; planes 3 / 4: A pieces h / l
      split_quad_h2(gq, sG, h, l);
      *reinterpret_cast<u32x2*>(dst + 0 * PLANE) = h;
      *reinterpret_cast<u32x2*>(dst + 1 * PLANE) = l;
      if (want_db) csum += gq;
      split_quad_h2(aq, sA, h, l);
      *reinterpret_cast<u32x2*>(dst + 3 * PLANE) = h;
      *reinterpret_cast<u32x2*>(dst + 4 * PLANE) = l;
      return;
    }
    split_quad(gq, h, m, l);
    *reinterpret_cast<u32x2*>(dst + 0 * PLANE) = h;
    *reinterpret_cast<u32x2*>(dst + 1 * PLANE) = m;
    *reinterpret_cast<u32x2*>(dst + 2 * PLANE) = l;
    if (want_db) csum += gq;
    split_quad(aq, h, m, l);
    *reinterpret_cast<u32x2*>(dst + 3 * PLANE) = h;
    *reinterpret_cast<u32x2*>(dst + 4 * PLANE) = m;
    *reinterpret_cast<u32x2*>(dst + 5 * PLANE) = l;
  };

  // 16 waves: wave owns dW rows [32 wr, +32) x cols [32 wc, +32) = 2 x 2 MFMA blocks
  const int wr = wave >> 2, wc = wave & 3;
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // chunk c is staged from register set c % NPF into LDS buffer c & 1 one iteration before its MFMAs; one barrier
  // per chunk orders "buffer written" and "buffer free" at once
  if (nchunk == 0) {   // cannot happen for a launched slab; keeps the unconditional loads in range
    float* part0 = tab.partials + tr.tile * (TB * TB);
    for (int o = tid; o < TB * TB; o += WG_THREADS) part0[o] = 0.f;
    if (want_db && tid < TB) tab.colsums[tr.tile * TB + tid] = 0.f;
    return;
  }
#pragma unroll
  for (int c = 0; c < NPF; ++c) fetch(c, c);
  stash(0, 0);
  fetch(NPF, 0);
  wg_barrier();
  // one chunk: stage chunk c + 1 (register set SET1 -> buffer BUF ^ 1), refill that set, multiply chunk c from BUF.
  // (Running the two halves in opposite order on every other wave of a SIMD, so that its VALU and matrix pipes
  // overlap, measured no gain: profiles/wgrad_timeline.py shows a step of ~4.1k cycles = issue-bound, 25-30 % of it
  // barrier skew.)
  auto stage_next = [&](int c, auto set1_tag, auto buf_tag) {
    constexpr int SET1 = decltype(set1_tag)::value, BUF = decltype(buf_tag)::value;
    stash(SET1, BUF ^ 1);          // past the last chunk this stages zeros nobody reads
    fetch(c + 1 + NPF, SET1);
  };
  auto multiply = [&](auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
    const short* buf = planes + BUF * (6 * PLANE);
    if (jb) {
      bf16x8 g1[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) g1[a] = column_fragment(buf + 0 * PLANE, 32 * wr + 16 * a, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bf16x8 a1 = column_fragment(buf + 3 * PLANE, 32 * wc + 16 * b, lane);
        acc[0][b] = mma(g1[0], a1, acc[0][b]);
        acc[1][b] = mma(g1[1], a1, acc[1][b]);
      }
      return;
    }
    if (H2) {
      bf16x8 g_h[2], g_l[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        g_h[a] = column_fragment(buf + 0 * PLANE, 32 * wr + 16 * a, lane);
        g_l[a] = column_fragment(buf + 1 * PLANE, 32 * wr + 16 * a, lane);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bf16x8 a_h = column_fragment(buf + 3 * PLANE, 32 * wc + 16 * b, lane);
        const bf16x8 a_l = column_fragment(buf + 4 * PLANE, 32 * wc + 16 * b, lane);
        acc[0][b] = mma_h(g_h[0], a_l, acc[0][b]);
        acc[1][b] = mma_h(g_h[1], a_l, acc[1][b]);
        acc[0][b] = mma_h(g_l[0], a_h, acc[0][b]);
        acc[1][b] = mma_h(g_l[1], a_h, acc[1][b]);
        acc[0][b] = mma_h(g_h[0], a_h, acc[0][b]);
        acc[1][b] = mma_h(g_h[1], a_h, acc[1][b]);
      }
      return;
    }
    bf16x8 gh[2], gm[2], gl[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      gh[a] = column_fragment(buf + 0 * PLANE, 32 * wr + 16 * a, lane);
      gm[a] = column_fragment(buf + 1 * PLANE, 32 * wr + 16 * a, lane);
      gl[a] = column_fragment(buf + 2 * PLANE, 32 * wr + 16 * a, lane);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const bf16x8 ah = column_fragment(buf + 3 * PLANE, 32 * wc + 16 * b, lane);
      const bf16x8 am = column_fragment(buf + 4 * PLANE, 32 * wc + 16 * b, lane);
      const bf16x8 al = column_fragment(buf + 5 * PLANE, 32 * wc + 16 * b, lane);
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
  };
  int stamp_i = 0;
  auto stamp = [&]() {  // experiments: waves 0 and 4 record the shader clock (64 slots each)
    if (TIMING && tab.timing && (tid == 0 || tid == 256) && stamp_i < 64)
      tab.timing[(int64_t(blockIdx.x) * 2 + (tid >> 8)) * 64 + stamp_i++] = __builtin_amdgcn_s_memtime();
  };
  auto step = [&](int c, auto set1_tag, auto buf_tag) {
    stamp();
    stage_next(c, set1_tag, buf_tag);
    stamp();
    multiply(buf_tag);
    stamp();
    wg_barrier();
  };
  using std::integral_constant;
  // 6 = lcm(register sets, buffers): both are compile-time in every step.  The body has no conditional step (a
  // skipped step would make hipcc's vmcnt bookkeeping pessimistic: vmcnt(0) instead of leaving two chunks in
  // flight); chunks past the slab are zeros and add nothing -- the launcher makes slabs multiples of 6 chunks.
  for (int c = 0; c < nchunk; c += 6) {
    step(c, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    step(c + 1, integral_constant<int, 2>{}, integral_constant<int, 1>{});
    step(c + 2, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    step(c + 3, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    step(c + 4, integral_constant<int, 2>{}, integral_constant<int, 0>{});
    step(c + 5, integral_constant<int, 0>{}, integral_constant<int, 1>{});
  }

  // D[row = 4 q + r][col = c] of block (a, b) = dW[32 wr + 16 a + 4 q + r][32 wc + 16 b + c]
  const int q = lane >> 4, cc = lane & 15;
  float* part = tab.partials + tr.tile * (TB * TB);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)   // H2: un-scale by the exact power of two 2^-(k_G + k_A) (ldexp: exact over the whole range)
        part[(32 * wr + 16 * a + 4 * q + r) * TB + 32 * wc + 16 * b + cc] = H2 ? ldexpf(acc[a][b][r], Eg + Ea - 282) : acc[a][b][r];
  if (want_db) {  // combine the 32 row-threads of each column quad in fixed order through LDS
    f32x4* red = reinterpret_cast<f32x4*>(planes);
    red[tid] = csum;
    __syncthreads();
    if (tid < 32) {
      f32x4 v = red[tid];
      for (int l = 1; l < 32; ++l) v += red[l * 32 + tid];
      *reinterpret_cast<f32x4*>(tab.colsums + tr.tile * TB + tid * 4) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// k_wgrad_wide (round 6): the fp16 x 2 jobs at D = 256.  k_wgrad cuts dW into four 128 x 128 blocks per slab there, so every
// row of G and of A is read from HBM -- and split into its pieces -- TWICE (profiles/r06_surface_notes.txt: level 0 of the surface
// step, 1.2 GB algorithmic, 730 us on 128 workgroups).  Here a workgroup owns 128 rows of dW (columns of G) x ALL 256 columns (of
// A): A is read and split once, G twice; a wave owns 32 x 64 of dW = 2 x 4 MFMA blocks, i.e. 24 products per 12 fragment reads
// where k_wgrad has 12 per 8.  Same staging, same step schedule, same products in the same order per accumulator (chunk after
// chunk, rows ascending; h l, l h, h h) -- the results are bit-identical to k_wgrad<., false, true>.  LDS: 2 x [G h | G l | A h | A l]
// = 2 x (2 x 9 KB + 2 x 17 KB) = 104 KB; the A planes have a 544-byte row pitch (136 dwords: again 8 banks further per row).
constexpr int LROW_W = 272;
constexpr int PLANE_W = RC * LROW_W;
constexpr int BUF_W = 2 * PLANE + 2 * PLANE_W;

__device__ __forceinline__ bf16x8 column_fragment_w(const short* plane, int col0, int lane) {   // column_fragment on a 272-short pitch
  const int q = lane >> 4, ip = lane & 15;
  const short* p = plane + (4 * q + (ip >> 2)) * LROW_W + col0 + 4 * (ip & 3);
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * LROW_W));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// Tile of a wide launch -> (slab, block of G columns).  The two workgroups of a slab read the same rows of A: they are placed EIGHT
// workgroup indices apart, i.e. on the same XCD (workgroups go round-robin over the 8 XCDs) and in the same dispatch round, so that the
// second read of an A chunk is served by that XCD's L2 instead of HBM.  Whole groups of 8 slabs; the last slabs pair up as neighbours.
__device__ __forceinline__ void wide_tile(int local, int nsplit, int& split, int& bi) {
  const int full = (nsplit >> 3) << 3;
  if (local < 2 * full) {
    const int r = local & 15;
    bi = r >> 3;
    split = ((local >> 4) << 3) + (r & 7);
  } else {
    const int l2 = local - 2 * full;
    split = full + (l2 >> 1);
    bi = l2 & 1;
  }
}

__global__ __launch_bounds__(WG_THREADS) void k_wgrad_wide(WgradTable tab) {
  extern __shared__ __attribute__((aligned(16))) short planes[];  // [2 buffers][G h | G l | A h | A l]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int j = 0;
  while (j + 1 < tab.njobs && int(blockIdx.x) >= tab.first_tile[j + 1]) ++j;
  const WgradJob job = tab.job[j];
  int split, bi;
  wide_tile(blockIdx.x - tab.first_tile[j], tab.nsplit[j], split, bi);
  const int64_t tile = tab.first_tile[j] + split * 2 + bi;   // where k_wgrad_reduce looks for this block's partial sums
  const int64_t r0 = int64_t(split) * tab.rows_per_wg[j];
  const int64_t r1 = min(job.R, r0 + tab.rows_per_wg[j]);
  const int nchunk = int((r1 - r0 + RC - 1) / RC);
  const int n0 = bi * TB;
  constexpr int TILE = TB * 2 * TB;   // floats per partial tile: [128][256]
  // one power-of-two scale per operand tensor from its magnitude bound (k_wgrad, H2)
  int Eg, Ea;
  float sG, sA;
  {
    static_assert(kBoundWidth == 8 * WG_THREADS, "two float4 of each bound slot per thread");
    __shared__ unsigned bred[2][WG_THREADS / 64];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const u32x4* gs = reinterpret_cast<const u32x4*>(job.g_bound);
    const u32x4* as = reinterpret_cast<const u32x4*>(job.a_bound);
    const u32x4 gv = gs[tid], gw = gs[tid + WG_THREADS], av = as[tid], aw = as[tid + WG_THREADS];
    unsigned gm = max(max(max(gv[0], gv[1]), max(gv[2], gv[3])), max(max(gw[0], gw[1]), max(gw[2], gw[3])));
    unsigned am = max(max(max(av[0], av[1]), max(av[2], av[3])), max(max(aw[0], aw[1]), max(aw[2], aw[3])));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      gm = max(gm, (unsigned)__shfl_xor((int)gm, o, 64));
      am = max(am, (unsigned)__shfl_xor((int)am, o, 64));
    }
    if (lane == 0) { bred[0][wave] = gm; bred[1][wave] = am; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WG_THREADS / 64; ++w) { gm = max(gm, bred[0][w]); am = max(am, bred[1][w]); }
    Eg = bound_exp(__uint_as_float(gm) * job.g_mul);
    Ea = bound_exp(__uint_as_float(am) * job.a_mul);
    sG = __uint_as_float(unsigned(268 - Eg) << 23);
    sA = __uint_as_float(unsigned(268 - Ea) << 23);
  }
  // staging: one float4 of G (columns n0 + 4 (tid % 32) ..) and two of A (columns 4 (tid % 32) .. and 128 further) per thread and
  // chunk, row tid / 32: every load instruction covers full 512-byte bursts of 32 rows
  const int srow = tid >> 5, scol = (tid & 31) * 4;
  const float* gsrc = job.G + n0 + scol;
  const float* asrc = job.A + scol;
  const int64_t rlast = r1 - 1;
  constexpr int NPW = 2;   // register sets: two 64 KB chunks in flight per workgroup (k_wgrad: three of 32 KB); a third set spills at 128 VGPRs
  f32x4 sg[NPW], sa[NPW][2];
  bool live[NPW];
  auto fetch = [&](int chunk, int set) {   // chunks past the slab re-read its last rows (never staged)
    const int64_t r = r0 + int64_t(chunk) * RC + srow;
    live[set] = r < r1;
    const int64_t rc = r < r1 ? r : rlast;
    sg[set] = *reinterpret_cast<const f32x4*>(gsrc + rc * job.ldg);
    sa[set][0] = *reinterpret_cast<const f32x4*>(asrc + rc * job.lda);
    sa[set][1] = *reinterpret_cast<const f32x4*>(asrc + rc * job.lda + TB);
  };
  const bool want_db = job.db != nullptr;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  auto stash = [&](int set, int buf) {
    short* dst = planes + buf * BUF_W;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    u32x2 h, l;
    const f32x4 gq = live[set] ? sg[set] : zero;
    split_quad_h2(gq, sG, h, l);
    *reinterpret_cast<u32x2*>(dst + srow * LROW + scol) = h;
    *reinterpret_cast<u32x2*>(dst + PLANE + srow * LROW + scol) = l;
    if (want_db) csum += gq;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 aq = live[set] ? sa[set][u] : zero;
      split_quad_h2(aq, sA, h, l);
      *reinterpret_cast<u32x2*>(dst + 2 * PLANE + srow * LROW_W + scol + TB * u) = h;
      *reinterpret_cast<u32x2*>(dst + 2 * PLANE + PLANE_W + srow * LROW_W + scol + TB * u) = l;
    }
  };
  // 16 waves: wave owns dW rows [32 wr, +32) x cols [64 wc, +64) = 2 x 4 MFMA blocks
  const int wr = wave >> 2, wc = wave & 3;
  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nchunk == 0) {   // cannot happen for a launched slab; keeps the unconditional loads in range
    float* part0 = tab.partials + tile * TILE;
    for (int o = tid; o < TILE; o += WG_THREADS) part0[o] = 0.f;
    if (want_db && tid < TB) tab.colsums[tile * TB + tid] = 0.f;
    return;
  }
#pragma unroll
  for (int c = 0; c < NPW; ++c) fetch(c, c);
  stash(0, 0);
  fetch(NPW, 0);
  wg_barrier();
  auto multiply = [&](auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
    const short* buf = planes + BUF * BUF_W;
    bf16x8 g_h[2], g_l[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      g_h[a] = column_fragment(buf, 32 * wr + 16 * a, lane);
      g_l[a] = column_fragment(buf + PLANE, 32 * wr + 16 * a, lane);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const bf16x8 a_h = column_fragment_w(buf + 2 * PLANE, 64 * wc + 16 * b, lane);
      const bf16x8 a_l = column_fragment_w(buf + 2 * PLANE + PLANE_W, 64 * wc + 16 * b, lane);
      acc[0][b] = mma_h(g_h[0], a_l, acc[0][b]);
      acc[1][b] = mma_h(g_h[1], a_l, acc[1][b]);
      acc[0][b] = mma_h(g_l[0], a_h, acc[0][b]);
      acc[1][b] = mma_h(g_l[1], a_h, acc[1][b]);
      acc[0][b] = mma_h(g_h[0], a_h, acc[0][b]);
      acc[1][b] = mma_h(g_h[1], a_h, acc[1][b]);
    }
  };
  auto step = [&](int c, auto set1_tag, auto buf_tag) {
    constexpr int SET1 = decltype(set1_tag)::value, BUF = decltype(buf_tag)::value;
    stash(SET1, BUF ^ 1);          // past the last chunk this stages zeros nobody reads
    fetch(c + 1 + NPW, SET1);
    multiply(buf_tag);
    wg_barrier();
  };
  using std::integral_constant;
  for (int c = 0; c < nchunk; c += 2) {   // chunk c: register set c % 2, buffer c & 1; no conditional step (slabs are multiples of 6 chunks)
    step(c, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    step(c + 1, integral_constant<int, 0>{}, integral_constant<int, 1>{});
  }
  // D[row = 4 q + r][col = c] of block (a, b) = dW[n0 + 32 wr + 16 a + 4 q + r][64 wc + 16 b + c]
  const int q = lane >> 4, cc = lane & 15;
  float* part = tab.partials + tile * TILE;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        part[(32 * wr + 16 * a + 4 * q + r) * (2 * TB) + 64 * wc + 16 * b + cc] = ldexpf(acc[a][b][r], Eg + Ea - 282);
  if (want_db) {  // combine the 32 row-threads of each column quad in fixed order through LDS
    f32x4* red = reinterpret_cast<f32x4*>(planes);
    red[tid] = csum;
    __syncthreads();
    if (tid < 32) {
      f32x4 v = red[tid];
      for (int l = 1; l < 32; ++l) v += red[l * 32 + tid];
      *reinterpret_cast<f32x4*>(tab.colsums + tile * TB + tid * 4) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// k_wgrad_bf64: the bf16 jobs (WgradJob::bf16: gradient and activation tensors stored as bf16, ONE product per fragment
// pair) with 64-row chunks.  In k_wgrad<., true> a chunk is 32 rows = 8 KB per operand and a thread moves 8 bytes of each:
// half the bytes in flight per CU of the fp32 form at the same step overhead, and the chunk step is paced by HBM latency
// x bytes in flight -- 2.6-2.9 TB/s on 132 workgroups.  Here a thread moves 16 bytes of each operand per
// chunk (rows tid / 16, columns 8 (tid % 16) ..), a chunk is two K steps of the MFMA: the same 32 KB per step and 96 KB
// in flight as the fp32 kernel.  Same products in the same order per accumulator (chunk after chunk, rows ascending).
constexpr int RC2 = 64;
constexpr int PLANE2 = RC2 * LROW;

__global__ __launch_bounds__(WG_THREADS) void k_wgrad_bf64(WgradTable tab) {
  extern __shared__ __attribute__((aligned(16))) short planes[];  // [2 buffers][G|A][64 rows][LROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int j = 0;
  while (j + 1 < tab.njobs && int(blockIdx.x) >= tab.first_tile[j + 1]) ++j;
  const WgradJob job = tab.job[j];
  const TileRef tr = tile_of(tab, j, blockIdx.x);
  const int bj = tr.bj, bi = tr.bi, split = tr.split;
  const int D = tab.D;
  const int64_t r0 = int64_t(split) * tab.rows_per_wg[j];
  const int64_t r1 = min(job.R, r0 + tab.rows_per_wg[j]);
  const int nchunk = int((r1 - r0 + RC2 - 1) / RC2);
  const int n0 = bi * TB, k0 = bj * TB;
  float* part = tab.partials + tr.tile * (TB * TB);
  const bool want_db = job.db && bj == 0;
  if (nchunk == 0) {   // cannot happen for a launched slab
    for (int o = tid; o < TB * TB; o += WG_THREADS) part[o] = 0.f;
    if (want_db && tid < TB) tab.colsums[tr.tile * TB + tid] = 0.f;
    return;
  }
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  const int srow = tid >> 4, scol = (tid & 15) * 8;
  const bool vg = n0 + scol < D, va = k0 + scol < D;
  const unsigned short* gsrc = reinterpret_cast<const unsigned short*>(job.G) + (vg ? n0 + scol : 0);
  const unsigned short* asrc = reinterpret_cast<const unsigned short*>(job.A) + (va ? k0 + scol : 0);
  const int64_t rlast = r1 - 1;
  u32x4 sg[NPF], sa[NPF];
  bool live[NPF];
  auto fetch = [&](int chunk, int set) {   // unconditional loads: chunks past the slab re-read its last row (never staged)
    const int64_t r = r0 + int64_t(chunk) * RC2 + srow;
    live[set] = r < r1;
    const int64_t rc = r < r1 ? r : rlast;
    sg[set] = *reinterpret_cast<const u32x4*>(gsrc + rc * job.ldg);
    sa[set] = *reinterpret_cast<const u32x4*>(asrc + rc * job.lda);
  };
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums of G over this thread's rows
  auto stash = [&](int set, int buf) {
    short* dst = planes + buf * (2 * PLANE2) + srow * LROW + scol;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const u32x4 gq = (live[set] && vg) ? sg[set] : zero, aq = (live[set] && va) ? sa[set] : zero;
    *reinterpret_cast<u32x4*>(dst) = gq;
    *reinterpret_cast<u32x4*>(dst + PLANE2) = aq;
    if (want_db) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        csum[2 * k] += __uint_as_float(gq[k] << 16);
        csum[2 * k + 1] += __uint_as_float(gq[k] & 0xffff0000u);
      }
    }
  };
  const int wr = wave >> 2, wc = wave & 3;
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](int buf) {
    const short* base = planes + buf * (2 * PLANE2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {   // two K steps of 32 rows
      const short* pg = base + ks * 32 * LROW;
      const short* pa = base + PLANE2 + ks * 32 * LROW;
      bf16x8 g1[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) g1[a] = column_fragment(pg, 32 * wr + 16 * a, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bf16x8 a1 = column_fragment(pa, 32 * wc + 16 * b, lane);
        acc[0][b] = mma(g1[0], a1, acc[0][b]);
        acc[1][b] = mma(g1[1], a1, acc[1][b]);
      }
    }
  };
#pragma unroll
  for (int c = 0; c < NPF; ++c) fetch(c, c);
  stash(0, 0);
  fetch(NPF, 0);
  wg_barrier();
  auto step = [&](int c, auto set1_tag, auto buf_tag) {
    constexpr int SET1 = decltype(set1_tag)::value, BUF = decltype(buf_tag)::value;
    stash(SET1, BUF ^ 1);          // chunk c + 1 -> the other buffer (zeros past the slab)
    fetch(c + 1 + NPF, SET1);
    multiply(BUF);
    wg_barrier();
  };
  using std::integral_constant;
  for (int c = 0; c < nchunk; c += 6) {   // the schedule of k_wgrad: register sets and buffers compile-time in every step
    step(c, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    step(c + 1, integral_constant<int, 2>{}, integral_constant<int, 1>{});
    step(c + 2, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    step(c + 3, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    step(c + 4, integral_constant<int, 2>{}, integral_constant<int, 0>{});
    step(c + 5, integral_constant<int, 0>{}, integral_constant<int, 1>{});
  }
  const int q = lane >> 4, cc = lane & 15;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(32 * wr + 16 * a + 4 * q + r) * TB + 32 * wc + 16 * b + cc] = acc[a][b][r];
  if (want_db) {  // the 64 row-threads of each column octet in fixed order through LDS
    float* red = reinterpret_cast<float*>(planes);   // [64 row lanes][128 columns]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[srow * TB + scol + k] = csum[k];
    __syncthreads();
    if (tid < TB) {
      float v = red[tid];
      for (int l = 1; l < 64; ++l) v += red[l * TB + tid];
      tab.colsums[tr.tile * TB + tid] = v;
    }
  }
}

// k_wgrad_bf64_wide (round 6): k_wgrad_bf64 at D = 256 with the 128 x 256 tiles of k_wgrad_wide -- A (bf16 rows) is read once instead
// of twice.  64-row chunks: a thread moves 16 bytes of G and 2 x 16 bytes of A; same products in the same order per accumulator.
constexpr int PLANE2_W = RC2 * LROW_W;
constexpr int BUF2_W = PLANE2 + PLANE2_W;   // [G | A]: 18 KB + 34 KB

__global__ __launch_bounds__(WG_THREADS) void k_wgrad_bf64_wide(WgradTable tab) {
  extern __shared__ __attribute__((aligned(16))) short planes[];  // [2 buffers][G 64 x LROW | A 64 x LROW_W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int j = 0;
  while (j + 1 < tab.njobs && int(blockIdx.x) >= tab.first_tile[j + 1]) ++j;
  const WgradJob job = tab.job[j];
  int split, bi;
  wide_tile(blockIdx.x - tab.first_tile[j], tab.nsplit[j], split, bi);
  const int64_t tile = tab.first_tile[j] + split * 2 + bi;
  const int64_t r0 = int64_t(split) * tab.rows_per_wg[j];
  const int64_t r1 = min(job.R, r0 + tab.rows_per_wg[j]);
  const int nchunk = int((r1 - r0 + RC2 - 1) / RC2);
  const int n0 = bi * TB;
  constexpr int TILE = TB * 2 * TB;
  float* part = tab.partials + tile * TILE;
  const bool want_db = job.db != nullptr;
  if (nchunk == 0) {   // cannot happen for a launched slab
    for (int o = tid; o < TILE; o += WG_THREADS) part[o] = 0.f;
    if (want_db && tid < TB) tab.colsums[tile * TB + tid] = 0.f;
    return;
  }
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  const int srow = tid >> 4, scol = (tid & 15) * 8;
  const unsigned short* gsrc = reinterpret_cast<const unsigned short*>(job.G) + n0 + scol;
  const unsigned short* asrc = reinterpret_cast<const unsigned short*>(job.A) + scol;
  const int64_t rlast = r1 - 1;
  constexpr int NPW = 2;   // register sets: two 48 KB chunks in flight (a third set spills at 128 VGPRs)
  u32x4 sg[NPW], sa[NPW][2];
  bool live[NPW];
  auto fetch = [&](int chunk, int set) {   // unconditional loads: chunks past the slab re-read its last row (never staged)
    const int64_t r = r0 + int64_t(chunk) * RC2 + srow;
    live[set] = r < r1;
    const int64_t rc = r < r1 ? r : rlast;
    sg[set] = *reinterpret_cast<const u32x4*>(gsrc + rc * job.ldg);
    sa[set][0] = *reinterpret_cast<const u32x4*>(asrc + rc * job.lda);
    sa[set][1] = *reinterpret_cast<const u32x4*>(asrc + rc * job.lda + TB);
  };
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums of G over this thread's rows
  auto stash = [&](int set, int buf) {
    short* dst = planes + buf * BUF2_W;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const u32x4 gq = live[set] ? sg[set] : zero;
    *reinterpret_cast<u32x4*>(dst + srow * LROW + scol) = gq;
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<u32x4*>(dst + PLANE2 + srow * LROW_W + scol + TB * u) = live[set] ? sa[set][u] : zero;
    if (want_db) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        csum[2 * k] += __uint_as_float(gq[k] << 16);
        csum[2 * k + 1] += __uint_as_float(gq[k] & 0xffff0000u);
      }
    }
  };
  const int wr = wave >> 2, wc = wave & 3;   // dW rows [32 wr, +32) x cols [64 wc, +64)
  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](int buf) {
    const short* base = planes + buf * BUF2_W;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {   // two K steps of 32 rows
      const short* pg = base + ks * 32 * LROW;
      const short* pa = base + PLANE2 + ks * 32 * LROW_W;
      bf16x8 g1[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) g1[a] = column_fragment(pg, 32 * wr + 16 * a, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bf16x8 a1 = column_fragment_w(pa, 64 * wc + 16 * b, lane);
        acc[0][b] = mma(g1[0], a1, acc[0][b]);
        acc[1][b] = mma(g1[1], a1, acc[1][b]);
      }
    }
  };
#pragma unroll
  for (int c = 0; c < NPW; ++c) fetch(c, c);
  stash(0, 0);
  fetch(NPW, 0);
  wg_barrier();
  auto step = [&](int c, auto set1_tag, auto buf_tag) {
    constexpr int SET1 = decltype(set1_tag)::value, BUF = decltype(buf_tag)::value;
    stash(SET1, BUF ^ 1);          // chunk c + 1 -> the other buffer (zeros past the slab)
    fetch(c + 1 + NPW, SET1);
    multiply(BUF);
    wg_barrier();
  };
  using std::integral_constant;
  for (int c = 0; c < nchunk; c += 2) {   // chunk c: register set c % 2, buffer c & 1
    step(c, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    step(c + 1, integral_constant<int, 0>{}, integral_constant<int, 1>{});
  }
  const int q = lane >> 4, cc = lane & 15;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(32 * wr + 16 * a + 4 * q + r) * (2 * TB) + 64 * wc + 16 * b + cc] = acc[a][b][r];
  if (want_db) {  // the 64 row-threads of each column octet in fixed order through LDS
    float* red = reinterpret_cast<float*>(planes);   // [64 row lanes][128 columns]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[srow * TB + scol + k] = csum[k];
    __syncthreads();
    if (tid < TB) {
      float v = red[tid];
      for (int l = 1; l < 64; ++l) v += red[l * TB + tid];
      tab.colsums[tile * TB + tid] = v;
    }
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
      if (tab.wide) {   // k_wgrad_wide: tile = split * nblk + bi, [128][256] floats
        base = tab.partials + int64_t(tab.first_tile[j] + bi) * (TB * 2 * TB) + (n % TB) * (2 * TB) + k;
        stride = int64_t(nblk) * (TB * 2 * TB);
      } else {
        base = tab.partials + int64_t(tab.first_tile[j] + bi * nblk + bj) * (TB * TB) + (n % TB) * TB + (k % TB);
        stride = int64_t(nblk) * nblk * (TB * TB);
      }
    } else {
      const int n = (o4 - nmat) * 4, bi = n / TB;
      base = tab.colsums + int64_t(tab.first_tile[j] + (tab.wide ? bi : bi * nblk)) * TB + (n % TB);
      stride = int64_t(nblk) * (tab.wide ? 1 : nblk) * TB;
    }
    int sp = sl;
    for (; sp + 12 < ns; sp += 16) {   // four slabs per round: the loads first (one at a time this loop is 8-11 dependent round trips), the adds in slab order
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + (sp + 4 * u) * stride);
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; sp < ns; sp += 4) {
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
constexpr int SW_WGS = 512;   // workgroups (= partial blocks the reduce sums); each keeps SW_UNROLL rows per lane in flight
constexpr int SW_UNROLL = 4;

__device__ __forceinline__ void small_row(const SmallWgradArgs& a, int64_t r, float (&sv)[SW_MAXS]) {
  if (a.S) {
    const int ld = a.S_ld ? a.S_ld : a.S_cols;
    if ((ld & 3) == 0) {   // padded rows (the saved fiber): one or two 16-byte loads
      const float4 lo = *reinterpret_cast<const float4*>(a.S + r * ld);
      const float4 hi = ld > 4 ? *reinterpret_cast<const float4*>(a.S + r * ld + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      sv[0] = lo.x; sv[1] = lo.y; sv[2] = lo.z; sv[3] = lo.w; sv[4] = hi.x; sv[5] = hi.y; sv[6] = hi.z; sv[7] = hi.w;
    } else {
#pragma unroll
      for (int s = 0; s < SW_MAXS; ++s) sv[s] = s < a.S_cols ? a.S[r * ld + s] : 0.f;
    }
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
  float cs[SW_MAXS];   // column sums of the narrow matrix itself (decoder output bias), kept by the c4 == 0 lanes
#pragma unroll
  for (int s = 0; s < SW_MAXS; ++s) cs[s] = 0.f;
  const int64_t r0 = int64_t(blockIdx.x) * rows_per_wg, r1 = min(a.R, r0 + rows_per_wg);
  if (active && r0 < r1) {
    // SW_UNROLL rows of this lane in flight: all loads first (unconditional: rows past the slab re-read its last row and
    // are dropped by a select), then the products in row order -- the sums see the same sequence as a one-row loop
    for (int64_t r = r0 + rl; r < r1; r += int64_t(SW_UNROLL) * nrl) {
      float4 g[SW_UNROLL];
      float sv[SW_UNROLL][SW_MAXS];
      bool ok[SW_UNROLL];
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) {
        const int64_t rr = r + int64_t(u) * nrl;
        ok[u] = rr < r1;
        const int64_t rc = ok[u] ? rr : r1 - 1;
        g[u] = *reinterpret_cast<const float4*>(a.G + rc * D + c4 * 4);
        small_row(a, rc, sv[u]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          acc[s].x = fmaf(g[u].x, sv[u][s], acc[s].x); acc[s].y = fmaf(g[u].y, sv[u][s], acc[s].y);
          acc[s].z = fmaf(g[u].z, sv[u][s], acc[s].z); acc[s].w = fmaf(g[u].w, sv[u][s], acc[s].w);
        }
        acc[NS].x += g[u].x; acc[NS].y += g[u].y; acc[NS].z += g[u].z; acc[NS].w += g[u].w;
#pragma unroll
        for (int s = 0; s < NS; ++s) cs[s] += sv[u][s];
      }
    }
  }
  float* out = part + int64_t(blockIdx.x) * (SW_MAXS + 2) * D;
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
  if (a.colsum_S) {   // row SW_MAXS + 1 of the partial block: colsum of S in its first SW_MAXS entries
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      red[tid] = (active && c4 == 0) ? make_float4(cs[4 * h], cs[4 * h + 1], cs[4 * h + 2], cs[4 * h + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
      if (tid == 0) {
        float4 v = red[0];
        for (int l = 1; l < nrl; ++l) {
          const float4 w = red[l * d4];
          v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        *reinterpret_cast<float4*>(out + (SW_MAXS + 1) * D + 4 * h) = v;
      }
      __syncthreads();
    }
  }
}

// one block per 8 output floats; 32 partial-lanes each sum every 32nd workgroup partial, fixed-order combine
__global__ __launch_bounds__(256) void k_small_reduce(SmallWgradArgs a, const float* part, int nwg) {
  __shared__ float red[256];
  const int D = a.D, S = a.S_cols;
  const int o = blockIdx.x * 8 + (threadIdx.x & 7), pl = threadIdx.x >> 3;
  const int s = o / D, f = o % D;
  // rows 0..S-1: narrow products; row SW_MAXS: colsum(G); row SW_MAXS+1, entries < S: colsum(S)
  const bool valid = o < (SW_MAXS + 2) * D && (s < S || s == SW_MAXS || (s == SW_MAXS + 1 && f < S && a.colsum_S));
  float v = 0.f;
  if (valid) {
    const float* col = part + int64_t(s) * D + f;
    const int64_t pitch = int64_t(SW_MAXS + 2) * D;
    float v4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent loads in flight per lane, combined in fixed order
    int w = pl;
    for (; w + 96 < nwg; w += 128) {
      v4[0] += col[int64_t(w) * pitch]; v4[1] += col[int64_t(w + 32) * pitch];
      v4[2] += col[int64_t(w + 64) * pitch]; v4[3] += col[int64_t(w + 96) * pitch];
    }
    for (; w < nwg; w += 32) v4[0] += col[int64_t(w) * pitch];
    v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
  }
  red[threadIdx.x] = v;
  __syncthreads();
  if (pl == 0 && valid) {
    for (int l = 1; l < 32; ++l) v += red[l * 8 + threadIdx.x];
    if (s < S) a.out[s * a.os + f * a.of] = v;
    else if (s == SW_MAXS) { if (a.colsum) a.colsum[f] = v; }
    else a.colsum_S[f] = v;
  }
}

}  // namespace

#ifdef BSMS_EXPERIMENTS
static unsigned long long* g_wgrad_timing = nullptr;
extern "C" void bsms_debug_set_wgrad_timing(unsigned long long* dev_buf) { g_wgrad_timing = dev_buf; }  // experiments only
#else
constexpr unsigned long long* g_wgrad_timing = nullptr;
#endif

namespace bsms {

size_t wgrad_work_bytes(int D, int njobs) {
  (void)D;
  (void)njobs;
  return size_t(kMaxTiles) * (TB * TB + TB) * sizeof(float);
}

static int64_t knob_rows(const char* name, int64_t dflt) {   // experiment builds read launch-shape knobs from the environment
#ifdef BSMS_EXPERIMENTS
  const char* e = getenv(name);
  return e ? atoll(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}
static int launch_wgrad_same(int D, const WgradJob* jobs, int njobs, void* work, hipStream_t s, bool bf, bool h2) {
  BSMS_REQUIRE(njobs >= 0 && njobs <= kMaxWgradJobs, BSMS_E_INVALID_ARG, "wgrad: %d jobs (max %d)", njobs, kMaxWgradJobs);
  BSMS_REQUIRE(D % 4 == 0 && D <= 256, BSMS_E_UNSUPPORTED, "wgrad: D=%d", D);
  if (njobs == 0) return BSMS_OK;
  WgradTable tab{};
  tab.njobs = njobs;
  tab.D = D;
  int64_t total_rows = 0;
  for (int j = 0; j < njobs; ++j) total_rows += jobs[j].R;
  tab.nblk = (int)ceil_div(D, TB);
#ifdef BSMS_EXPERIMENTS
  static const bool wide_on = [] { const char* e = getenv("BSMS_WGRAD_WIDE"); return !e || atoi(e) != 0; }();   // same-box A/B against the 128 x 128 blocks
#else
  constexpr bool wide_on = true;
#endif
  static const int64_t wide_min_rows = knob_rows("BSMS_WGRAD_WIDE_MIN", 262144);
  // D = 256: 128 x 256 tiles, A read once (k_wgrad_wide) -- for launches of edge-level size; short launches keep the 128 x 128 blocks
  const bool bf64 = bf && [] { const char* e = getenv("BSMS_WGRAD_BF64"); return !e || atoi(e) != 0; }();
  const bool wide = wide_on && (bf ? bf64 : h2) && D == 2 * TB && !g_wgrad_timing && total_rows >= wide_min_rows;
  tab.wide = wide ? 1 : 0;
  const int blocks = wide ? tab.nblk : tab.nblk * tab.nblk;
  const int tile_units = wide ? 2 : 1;   // partial tiles of 128 x 128 floats a workgroup writes
  // Slab length: aim for ~128 workgroups over all jobs = half the CUs (one workgroup per CU is resident).  Measured
  // (same-box A/B, steps/s of the training step): 64 -> 110.7, 96 -> 125.5, 128 -> 131.0, 160 -> 124.8, 192 -> 126.4,
  // 256 -> 130.5, 512 -> 128.0, 768 -> 126.5: long slabs amortise the pipeline fill and the partial-block traffic, a
  // grid that spills into a second, nearly empty round loses, and half the chip is left to the kernels that run
  // concurrently on the caller's stream.  Slabs are multiples of 6 chunks, at least 128 rows.
#ifdef BSMS_EXPERIMENTS
  static const int target_h2 = [] { const char* e = getenv("BSMS_WGRAD_WGS"); return e ? atoi(e) : 128; }();
  static const int target_bf3 = [] { const char* e = getenv("BSMS_WGRAD_WGS_BF3"); return e ? atoi(e) : 128; }();   // the range-free jobs (second side lane)
  const int target_wgs = (h2 || bf) ? target_h2 : target_bf3;
#else
  constexpr int target_wgs = 128;
#endif
  int64_t rows_per = std::max<int64_t>(128, ceil_div(total_rows * blocks, target_wgs));
  const int chunk_rows = bf ? RC2 : RC;                // bf16 jobs: 64-row chunks (k_wgrad_bf64)
  rows_per = ceil_div(rows_per, 6 * chunk_rows) * (6 * chunk_rows);   // whole groups of six chunks (the kernels' step schedule)
  for (;;) {
    int64_t tiles = 0;
    for (int j = 0; j < njobs; ++j) tiles += std::max<int64_t>(1, ceil_div(jobs[j].R, rows_per)) * blocks;
    if (tiles * tile_units <= kMaxTiles) break;
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
  tab.timing = g_wgrad_timing;
  const size_t lds = size_t(2) * 6 * PLANE * sizeof(short);   // 108 KB: 2 x [G|A][3 planes][32 rows][288 B]
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_wgrad<false>), (int)lds);
  static DynLdsAttr attr_t_dev;
  const hipError_t attr_t = attr_t_dev.ensure(reinterpret_cast<const void*>(&k_wgrad<true>), (int)lds);
  BSMS_REQUIRE(attr == hipSuccess && attr_t == hipSuccess, BSMS_E_HIP, "wgrad: cannot reserve %zu bytes of LDS", lds);
  if (wide && bf) {
    const size_t lds_bw = size_t(2) * BUF2_W * sizeof(short);   // 104 KB
    static DynLdsAttr attr_bw_dev;
    const hipError_t attr_bw = attr_bw_dev.ensure(reinterpret_cast<const void*>(&k_wgrad_bf64_wide), (int)lds_bw);
    BSMS_REQUIRE(attr_bw == hipSuccess, BSMS_E_HIP, "wgrad: cannot reserve %zu bytes of LDS (bf16 build, 128 x 256 tiles)", lds_bw);
    hipLaunchKernelGGL(k_wgrad_bf64_wide, dim3(first), dim3(WG_THREADS), lds_bw, s, tab);
  } else if (wide) {
    const size_t lds_w = size_t(2) * BUF_W * sizeof(short);   // 104 KB
    static DynLdsAttr attr_w_dev;
    const hipError_t attr_w = attr_w_dev.ensure(reinterpret_cast<const void*>(&k_wgrad_wide), (int)lds_w);
    BSMS_REQUIRE(attr_w == hipSuccess, BSMS_E_HIP, "wgrad: cannot reserve %zu bytes of LDS (128 x 256 tiles)", lds_w);
    hipLaunchKernelGGL(k_wgrad_wide, dim3(first), dim3(WG_THREADS), lds_w, s, tab);
  } else if (h2) {
    static DynLdsAttr attr_h_dev;
  const hipError_t attr_h = attr_h_dev.ensure(reinterpret_cast<const void*>(&k_wgrad<false, false, true>), (int)lds);
    BSMS_REQUIRE(attr_h == hipSuccess, BSMS_E_HIP, "wgrad: cannot reserve %zu bytes of LDS (fp16 x 2 build)", lds);
    hipLaunchKernelGGL((k_wgrad<false, false, true>), dim3(first), dim3(WG_THREADS), lds, s, tab);
  } else if (bf64) {
    const size_t lds_b = size_t(2) * 2 * PLANE2 * sizeof(short);   // 72 KB: 2 x [G|A][64 rows][288 B]
    static DynLdsAttr attr_b64_dev;
  const hipError_t attr_b64 = attr_b64_dev.ensure(reinterpret_cast<const void*>(&k_wgrad_bf64), (int)lds_b);
    BSMS_REQUIRE(attr_b64 == hipSuccess, BSMS_E_HIP, "wgrad: cannot reserve %zu bytes of LDS (bf16 build, 64-row chunks)", lds_b);
    hipLaunchKernelGGL(k_wgrad_bf64, dim3(first), dim3(WG_THREADS), lds_b, s, tab);
  } else if (bf) {
    static DynLdsAttr attr_b_dev;
  const hipError_t attr_b = attr_b_dev.ensure(reinterpret_cast<const void*>(&k_wgrad<false, true>), (int)lds);
    BSMS_REQUIRE(attr_b == hipSuccess, BSMS_E_HIP, "wgrad: cannot reserve %zu bytes of LDS (bf16 build)", lds);
    hipLaunchKernelGGL((k_wgrad<false, true>), dim3(first), dim3(WG_THREADS), lds, s, tab);
  } else if (tab.timing) hipLaunchKernelGGL(k_wgrad<true>, dim3(first), dim3(WG_THREADS), lds, s, tab);
  else hipLaunchKernelGGL(k_wgrad<false>, dim3(first), dim3(WG_THREADS), lds, s, tab);
  BSMS_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)ceil_div((D * D + D) / 4, 64), njobs), dim3(256), 0, s, tab);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

// jobs of both storage types may be mixed in one call: they go out as (at most) two launches on the same stream, the
// second reusing the partial-block workspace after the first one's reduction
int launch_wgrad(int D, const WgradJob* jobs, int njobs, void* work, hipStream_t s) {
  BSMS_REQUIRE(njobs >= 0 && njobs <= kMaxWgradJobs, BSMS_E_INVALID_ARG, "wgrad: %d jobs (max %d)", njobs, kMaxWgradJobs);
#ifdef BSMS_EXPERIMENTS
  static const bool skip = getenv("BSMS_SKIP_WGRAD") != nullptr;   // timing experiments: what do the weight gradients cost the step?
  if (skip) return BSMS_OK;
#endif
  WgradJob part[kMaxWgradJobs];
#ifdef BSMS_EXPERIMENTS
  static const bool force_bf3 = [] { const char* e = getenv("BSMS_WGRAD_BF3"); return e && atoi(e) != 0; }();   // A/B: range-free six-product arithmetic for every fp32 job
#else
  constexpr bool force_bf3 = false;
#endif
  for (int mode = 0; mode < 3; ++mode) {   // 0: fp32 with bounds (fp16 x 2), 1: fp32 without (bf16 x 3), 2: bf16 tensors
    int n = 0;
    for (int j = 0; j < njobs; ++j) {
      const int m = jobs[j].bf16 ? 2 : ((jobs[j].g_bound && jobs[j].a_bound && !force_bf3) ? 0 : 1);
      if (m == mode) part[n++] = jobs[j];
    }
    if (n) {
      int rc = launch_wgrad_same(D, part, n, work, s, mode == 2, mode == 0);
      if (rc) return rc;
    }
  }
  return BSMS_OK;
}

size_t small_wgrad_work_bytes(int D) { return size_t(SW_WGS) * (SW_MAXS + 2) * D * sizeof(float); }
size_t small_wgrad_work_bytes_rows(int D, int64_t workers) {
  (void)workers;   // the fused scatter kernel strides its workers over the rows: at most 1024 partial blocks
  return size_t(std::max(SW_WGS, 1024)) * (SW_MAXS + 2) * D * sizeof(float);
}
int64_t small_wgrad_part_blocks(size_t work_bytes, int D) { return int64_t(work_bytes / (size_t(SW_MAXS + 2) * D * sizeof(float))); }

int launch_small_reduce(const SmallWgradArgs& a, const void* work, int nwg, hipStream_t s) {
  hipLaunchKernelGGL(k_small_reduce, dim3((unsigned)ceil_div((SW_MAXS + 2) * a.D, 8)), dim3(256), 0, s, a,
                     reinterpret_cast<const float*>(work), nwg);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

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
  hipLaunchKernelGGL(k_small_reduce, dim3((unsigned)ceil_div((SW_MAXS + 2) * a.D, 8)), dim3(256), 0, s, a, (const float*)part, nwg);
  BSMS_LAUNCH_CHECK();
  BSMS_REQUIRE(!a.colsum_S || a.S, BSMS_E_INVALID_ARG, "small_wgrad: colsum_S needs an explicit narrow matrix");
  return BSMS_OK;
}

}  // namespace bsms
