// Device-side building blocks of the chain kernels (register-tile helpers, the fp16 x 2 split, the LDS weight ring and its
// loader, the MFMA stage of one Linear): shared by chain.hip and efuse32.hip.  Everything here has internal linkage (anonymous
// namespace); the including file sets `#pragma clang fp contract(off)` BEFORE the include (see chain.hip for why).
#pragma once
#include "chain.h"

using namespace bsms;

namespace {


using f32x4 = __attribute__((ext_vector_type(4))) float;
#ifndef BSMS_CHAIN_WPE
#define BSMS_CHAIN_WPE 4   // generic chain kernels, multi-round launches: 128-VGPR budget (two 8-wave workgroups per CU); single-round (LONE) variants take 256
#endif
constexpr int kPL = 2;   // fp16 planes per weight of the fp32 path (chain.h); the bf16 precision has one

// Register layout ("chain layout", see chain.h): a wave owns 16 rows; lane l <-> row (l & 15), group g = l >> 4.
// For every 16-feature block t the lane holds features 16 t + 4 g + {0,1,2,3} as one f32x4.

// ---- bf16 storage (BSMS_BF16 precision: edge-level tensors are kept as bf16 in HBM, chain.h)
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// rows of a bf16 tensor [R, D]: lane (row, g) owns features 16 t + 4 g + {0..3} = 8 bytes per 16-feature block
template <int NB>
__device__ __forceinline__ void load_rows_bf16(f32x4 (&v)[NB], const void* base, int64_t row, int g) {
  const uint2* p = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + row * (NB * 16) + 4 * g);
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const uint2 u = p[4 * t];   // 16 features = 32 bytes = 4 uint2 further
    v[t] = f32x4{bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y)};
  }
}
template <int NB>
__device__ __forceinline__ void store_block_bf16(const f32x4 (&v)[NB], void* base, int64_t row, int g, int t) {
  uint2* p = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + row * (NB * 16) + 4 * g);
  p[4 * t] = make_uint2(pk_bf16(v[t][0], v[t][1]), pk_bf16(v[t][2], v[t][3]));
}
template <int NB>
__device__ __forceinline__ void store_rows_bf16(const f32x4 (&v)[NB], void* base, int64_t off, int g) {
  if (!base || off < 0) return;
#pragma unroll
  for (int t = 0; t < NB; ++t) store_block_bf16<NB>(v, base, off / (NB * 16), g, t);
}

// ---------------------------------------------------------------------------------- prepack ----
// fp32 -> two fp16 pieces of s * x (s a power of two; chain.h), two elements at a time: dword h = the hi pieces, dword l
// the lo pieces, element 0 in the low half.  v_fma_mix{lo,hi}_f16 computes fma(a, b, c) from fp32 / fp16 sources with ONE
// rounding to fp16: h = fp16(x s), l = fp16(x s - h) -- scaling, subtraction and rounding in one operation per piece.
__device__ __forceinline__ void split_h2(float x0, float x1, float s, unsigned& h, unsigned& l) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}

// Power-of-two scale of a row / matrix with maximum magnitude `amax` (>= 0): biased exponent E of amax, clamped so that
// s = 2^(139 - E) is a normal float; amax * s lies in [2^12, 2^13).  (E = 255: inf / nan rows propagate as such.)
struct RowScale { float s; int E; };
__device__ __forceinline__ RowScale scale_of(float amax, int emin = 12) {
  int E = int(__float_as_uint(amax) >> 23);
  E = E < emin ? emin : E;
  return RowScale{__uint_as_float(unsigned(266 - E) << 23), E};
}

// FRAG packs (layout in chain.h): per 32-feature K block c a chunk of kChunkHdrFloats + NB * 256 * planes dwords;
//   body dword ((t*planes + plane)*64 + lane)*4 + v  =  16-bit pair (slots 2v, 2v+1) of plane `plane` of
//   M[16 t + (lane & 15)][16 (2c + (i >> 2)) + 4 (lane >> 4) + (i & 3)],  i = slot
// with M[n][k] = W[row0+n][col0+k] (FRAG) or W[row0+k][col0+n] (FRAG_T).  fp32 path: planes = {h, l} of M * 2^k_w with
// k_w from the largest |M| of this matrix (and of its mate); bf16 precision: one plane, the rounded weight.
__device__ __forceinline__ float pack_elem(const PackDesc& d, int n, int k) {
  return (d.kind == PACK_FRAG_T) ? d.W[int64_t(d.row0 + k) * d.ld + d.col0 + n] : d.W[int64_t(d.row0 + n) * d.ld + d.col0 + k];
}
// PACK_ROWS_BF16 (chain.h): dword o of the 36 KB image = row n = o / 72, dword w = o % 72 of the row (64 data + 8 padding)
__device__ __forceinline__ void pack_rows_bf16(const PackDesc& d, int first, int stride) {
  unsigned* dst = reinterpret_cast<unsigned*>(d.dst);
  for (int o = first; o < d.N * 72; o += stride) {
    const int n = o / 72, w = o - n * 72, slot = w >> 1;
    unsigned v = 0u;
    if (slot < 32) {
      const int k = 4 * (slot ^ ((n >> 2) & 3)) + 2 * (w & 1);
      v = pk_bf16(pack_elem(d, n, k), pack_elem(d, n, k + 1));
    }
    dst[o] = v;
  }
}

// ------------------------------------------------------------------------ register-tile helpers
template <int NB>
__device__ __forceinline__ void zero_tile(f32x4 (&v)[NB]) {
#pragma unroll
  for (int t = 0; t < NB; ++t) v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// lane's row pointer; 4 lanes of a row read 64 contiguous bytes.  UNCONDITIONAL on purpose: a predicated load costs
// a branch, a zero-fill and a vmcnt wait per 16 bytes (hipcc then serialises a row into eight dependent round
// trips).  Lanes past the last row are pointed at row 0 by the caller; their results are never stored.
template <int NB>
__device__ __forceinline__ void load_rows(f32x4 (&v)[NB], const float* row, int g) {
#pragma unroll
  for (int t = 0; t < NB; ++t) v[t] = *reinterpret_cast<const f32x4*>(row + 16 * t + 4 * g);
}

// `base` is the tensor (wave-uniform, nullable), `off` this lane's row offset in floats (negative: lane past the last
// row, nothing is stored).  Callers keep ONE 64-bit per-lane value (the offset) instead of a pointer per tensor.
template <int NB, bool ACCUM>
__device__ __forceinline__ void store_rows(const f32x4 (&v)[NB], float* base, int64_t off, int g, int mode = 0) {
  if (!base || off < 0) return;
  float* row = base + off;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    f32x4* p = reinterpret_cast<f32x4*>(row + 16 * t + 4 * g);
    f32x4 x = v[t];
    if (ACCUM) x += *p;
    if (mode == 1) {
      __builtin_nontemporal_store(x, p);
    } else if (mode == 2) {
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
    } else {
      *p = x;
    }
  }
}

// Non-temporal (streaming) stores of a saved tensor, in 128-byte pieces.  In the chain layout one store instruction
// writes, per row, the 64 bytes held by that row's four lanes; streaming stores of 64-byte pieces run at 3.1 TB/s,
// of 128-byte pieces at 5.3 TB/s (profiles/census/store_bw.hip).  So feature blocks t, t+1 are paired: the lower
// eight rows of the wave keep block t and fetch their partner lane's (row + 8) block t+1 with one DPP half-row
// rotation; instruction A then writes rows 0-7 (both blocks = 128 contiguous bytes per row), instruction B rows 8-15.
// `row` is this lane's own row (may be >= nrows: the lane still carries its partner's data).  One pair per call: the
// chain issues pair c inside chunk c of the stage, because eight stores at once fill the CU's store path (its share
// of the chip's write bandwidth is ~10 B/clk) and the in-order wave then sits on the next store instead of its MFMAs.
template <int NB>
__device__ __forceinline__ void store_pair_stream(const f32x4 (&v)[NB], float* base, int64_t row, int64_t nrows, int lane,
                                                  int t, bool streaming = true) {   // feature blocks t, t + 1 (t even)
  if (!base) return;
  constexpr int D = NB * 16;
  const int g = lane >> 4;
  const bool hi = (lane & 8) != 0;
  const int64_t rowA = row - (lane & 8), rowB = rowA + 8;
  float* pA = base + rowA * D + 4 * g + (hi ? 16 : 0) + 16 * t;
  float* pB = base + rowB * D + 4 * g + (hi ? 0 : 16) + 16 * t;
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  const i32x4 own = __builtin_bit_cast(i32x4, v[t + 1]);
  i32x4 got;   // partner's block t + 1 (row_ror:8 = swap the two halves of a 16-lane row)
  got[0] = __builtin_amdgcn_update_dpp(own[0], own[0], 0x128, 0xf, 0xf, false);
  got[1] = __builtin_amdgcn_update_dpp(own[1], own[1], 0x128, 0xf, 0xf, false);
  got[2] = __builtin_amdgcn_update_dpp(own[2], own[2], 0x128, 0xf, 0xf, false);
  got[3] = __builtin_amdgcn_update_dpp(own[3], own[3], 0x128, 0xf, 0xf, false);
  const f32x4 x = __builtin_bit_cast(f32x4, got);
  const f32x4 dA = hi ? x : v[t], dB = hi ? v[t] : x;
#ifndef BSMS_EXPERIMENTS
  // production: the tensor is allocated for whole tiles (chain.h: pad_rows), no bounds test and no exec-mask branches
  (void)nrows;
  __builtin_nontemporal_store(dA, reinterpret_cast<f32x4*>(pA));
  __builtin_nontemporal_store(dB, reinterpret_cast<f32x4*>(pB));
  return;
#endif
  if (streaming) {
    if (rowA < nrows) __builtin_nontemporal_store(dA, reinterpret_cast<f32x4*>(pA));
    if (rowB < nrows) __builtin_nontemporal_store(dB, reinterpret_cast<f32x4*>(pB));
  } else {   // same 128-byte pieces, but allowed to stay in L2 / the memory-side cache for a reader that follows soon
    if (rowA < nrows) *reinterpret_cast<f32x4*>(pA) = dA;
    if (rowB < nrows) *reinterpret_cast<f32x4*>(pB) = dB;
  }
}

// ReLU sign bits of this lane's 4 NB values: bit (4 t + r) of word (4 t + r) / 32
template <int NB>
constexpr int mask_words() { return NB <= 8 ? 1 : NB / 8; }

// `BF`: the activation block in front of the bits is bf16 (R * D * 2 bytes) instead of fp32
template <int NB, bool BF = false>
__device__ __forceinline__ void store_mask_bits(const f32x4 (&v)[NB], float* act_base, int64_t R, int64_t off, int g) {
  if (!act_base || off < 0) return;
  constexpr int D = NB * 16, W = mask_words<NB>();
  unsigned* bits = reinterpret_cast<unsigned*>(act_base + (BF ? pad_rows(R) * D / 2 : pad_rows(R) * D)) + (off / D) * (4 * W) + g * W;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    unsigned m = 0;
    constexpr int N = (4 * NB < 32) ? 4 * NB : 32;
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {   // highest element first: every step shifts the word left by one and appends a bit
      const int e = 32 * w + k;
      // `v` is a post-ReLU activation (>= +0): it is positive iff its bit pattern is non-zero, i.e. iff 0 - bits has its top
      // bit set (bits <= 0x7fffffff).  v_sub + v_alignbit((m : 0 - bits) >> 31) = two operations per element and no inline asm
      // (round 4 used v_min_u32 in inline asm + shift + or: hipcc puts an s_nop behind every asm statement, 3.3 issue slots)
      m = __builtin_amdgcn_alignbit(m, 0u - __float_as_uint(v[e >> 2][e & 3]), 31);
    }
    bits[w] = m;
  }
}

// bias (or any per-feature vector) in this lane's feature order
template <int NB>
__device__ __forceinline__ void load_features(f32x4 (&v)[NB], const float* vec, int g) {
  load_rows<NB>(v, vec, g);
}

__device__ __forceinline__ float group_sum(float s) {  // sum over the 4 lane groups holding one row
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

template <int NB>
__device__ __forceinline__ float row_sum(const f32x4 (&v)[NB]) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NB; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
  return group_sum(s);
}

// ---------------------------------------------------------------------------- weight streaming
// Workgroup = 4 compute waves + 1 LOADER wave.  The loader streams the weight packs of all stages of every tile of
// this workgroup, chunk by chunk (a chunk = 32 input features x all outputs x 2 fp16 planes + header, 17 KB at
// D = 128), from L2 into a 3-deep LDS ring with LDS-DMA: no registers, two chunks in flight, a counted
// s_waitcnt vmcnt before it publishes a chunk at the workgroup barrier.  One barrier per chunk.
//
// Why a separate wave: vmcnt retires loads and stores through one in-order counter and hipcc waits vmcnt(0)
// whenever both kinds are outstanding, so a compute wave that fetched its own weights would stop at every chunk
// until its activation stores had reached memory -- MFMA phases and HBM phases then add up instead of
// overlapping (measured: kernel time = MFMA time + store time).  Compute waves execute NO vmcnt wait in the
// steady state; their stores drain in the background.
// PL: 16-bit planes per weight (2 = the fp32 path's fp16 pieces h, l; 1 = the bf16 precision, whose packs carry the rounded weight only)
template <int NB, int PL = kPL>
struct Ring {
  static constexpr int D = NB * 16;
  static constexpr int NCH = NB / 2;                          // chunks per stage (one per 32-feature K block)
  static constexpr int CHF = kChunkHdrFloats + NB * 256 * PL; // floats per chunk
  static constexpr int CH4 = CHF / 4;                         // float4 per chunk
  static constexpr int PER = CHF / 256;                       // LDS-DMA instructions (1 KB each) per chunk
  // LDS of a workgroup: [side table: the edge MLP's fiber weights][running magnitude bounds, 16 stages x 8 waves, padded
  // to 1 KB][ring: nr chunks].  The ring depth nr is a LAUNCH parameter (3..6): nr - 1 chunks are in flight, and a lone
  // workgroup per CU (coarse levels, node-level launches) is paced by the LDS-DMA latency of a chunk (~1 us) divided by
  // the chunks in flight, not by any throughput -- deep rings for launches that fit one round of workgroups, 3 slots
  // (two workgroups per CU) for the rest.
  static constexpr int SIDE_FLOATS = 8 * D;
  static constexpr int PRE_FLOATS = SIDE_FLOATS + 256;
  static constexpr int PRE4 = PRE_FLOATS / 4;
  static constexpr size_t lds_bytes(int nr) { return (size_t(PRE_FLOATS) + size_t(nr) * CHF) * sizeof(float); }
  static_assert(CHF % 256 == 0 && PER < 64, "chunk = whole LDS-DMA instructions, countable by vmcnt");
  static_assert(NB % 2 == 0, "K blocks are pairs of 16-feature blocks");
};

// workgroup barrier that waits for this wave's LDS traffic only (never for vmcnt)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_dst, +1 KB) in lane order.  Invisible to
// hipcc's waitcnt bookkeeping (inline asm), which is the point: the loader counts its own vmcnt.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// position in the ring: slot of the next chunk and the ring depth of this launch
struct Slot { int i, nr; };

// `side` (nullable, 8*D floats in HBM): copied once into the side table; the compute waves wait for it at one extra
// barrier before their first tile.
// NL loader waves share a chunk piece by piece (wave LI issues pieces LI, LI + NL, ...): one wave's LDS-DMA rate is its
// own ISSUE rate (~60 cycles per 1 KB piece alone, 100-185 beside a busy compute wave of its SIMD;
// profiles/census/ldsdma_rate.hip: 39 / 61 / 87 GB/s per CU with 1 / 2 / 4 loader waves), and since the fp32 products
// take three MFMAs per fragment pair instead of six the weight stream, not the matrix pipe, paces a stage.
template <int NB, int PL, int NL, int LI>
__device__ __forceinline__ void loader_run(const float4* const* wseq, int nseq, float4* lds, int lane, int ntiles, int nr,
                                           const float* side = nullptr) {
  using R = Ring<NB, PL>;
  constexpr int MINE = (R::PER - LI + NL - 1) / NL;   // pieces of a chunk this wave issues
  __builtin_amdgcn_s_setprio(3);                   // the loader must never be the wave the others wait for
  const unsigned pre0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
  const unsigned lds0 = pre0 + unsigned(R::PRE_FLOATS * sizeof(float));   // the ring
  if (side) {
    if (LI == 0) {
      const unsigned dst = pre0;
#pragma unroll
      for (int i = 0; i < R::SIDE_FLOATS / 256; ++i) glds16(reinterpret_cast<const float4*>(side) + i * 64 + lane, dst + i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const int my_tiles = (ntiles - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
  const int per_tile = nseq * R::NCH, total = my_tiles * per_tile;
  int is = 0, ic = 0, islot = 0;                   // next chunk to issue: sequence entry, chunk in it, ring slot
  auto issue = [&]() {
    const float4* src = wseq[is] + size_t(ic) * R::CH4 + lane;
    const unsigned dst = lds0 + unsigned(islot) * unsigned(R::CHF * sizeof(float));
#pragma unroll
    for (int i = 0; i < MINE; ++i) glds16(src + (LI + i * NL) * 64, dst + (LI + i * NL) * 1024);
    if (++ic == R::NCH) { ic = 0; if (++is == nseq) is = 0; }
    if (++islot == nr) islot = 0;
  };
  const int ahead = nr - 1;                        // chunks in flight
  for (int j = 0; j < ahead && j < total; ++j) issue();
  for (int j = 0; j < total; ++j) {
    // chunk j has landed once at most the younger chunks (issued after it: up to nr - 2) are still outstanding; the
    // launcher guarantees (nr - 2) * MINE <= 63 (the vmcnt field)
    const int younger = min(ahead - 1, total - 1 - j);
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MINE) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MINE <= 63 ? 2 * MINE : 63) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * MINE <= 63 ? 3 * MINE : 63) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * MINE <= 63 ? 4 * MINE : 63) : "memory"); break;
    }
    asm volatile("s_barrier" ::: "memory");       // barrier #j: publishes chunk j; everyone is done with chunk j-1,
    if (j + ahead < total) issue();                // whose slot chunk j + nr - 1 now overwrites
  }
}
// loader wave `li` of `nl` (1..3)
template <int NB, int PL = kPL>
__device__ __forceinline__ void loader_dispatch(int nl, int li, const float4* const* wseq, int nseq, float4* lds, int lane, int ntiles,
                                                int nr, const float* side = nullptr) {
  if (nl == 1) loader_run<NB, PL, 1, 0>(wseq, nseq, lds, lane, ntiles, nr, side);
  else if (nl == 2) { if (li == 0) loader_run<NB, PL, 2, 0>(wseq, nseq, lds, lane, ntiles, nr, side); else loader_run<NB, PL, 2, 1>(wseq, nseq, lds, lane, ntiles, nr, side); }
  else { if (li == 0) loader_run<NB, PL, 3, 0>(wseq, nseq, lds, lane, ntiles, nr, side); else if (li == 1) loader_run<NB, PL, 3, 1>(wseq, nseq, lds, lane, ntiles, nr, side);
         else loader_run<NB, PL, 3, 2>(wseq, nseq, lds, lane, ntiles, nr, side); }
}

// 16-bit pieces of one lane's activations as B operands: plane[kb2] = 8 values = slots i of K block kb2 (chain.h)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// fp16 pieces {h, l} of s * (K block kb2 = features 32 kb2 .. +31 of this lane's row)
template <int NB>
__device__ __forceinline__ void split_block(const f32x4 (&act)[NB], int kb2, float s, u32x4& bh, u32x4& bl) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {  // dword v = slots 2v, 2v+1 = act[2 kb2 + (v >> 1)][2 (v & 1) + {0, 1}]
    unsigned h, l;
    split_h2(act[2 * kb2 + (v >> 1)][2 * (v & 1)], act[2 * kb2 + (v >> 1)][2 * (v & 1) + 1], s, h, l);
    bh[v] = h;
    bl[v] = l;
  }
}

// bf16 precision: the activation IS its bf16 rounding -- one plane, no residuals
template <int NB>
__device__ __forceinline__ void round_block(const f32x4 (&act)[NB], int kb2, u32x4& bh) {
#pragma unroll
  for (int v = 0; v < 4; ++v) bh[v] = pk_bf16(act[2 * kb2 + (v >> 1)][2 * (v & 1)], act[2 * kb2 + (v >> 1)][2 * (v & 1) + 1]);
}

__device__ __forceinline__ f32x4 mma_bf(const float4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const float4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// largest |value| of this lane's row (all four lane groups)
template <int NB>
__device__ __forceinline__ float row_amax(const f32x4 (&v)[NB]) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    m = fmaxf(fmaxf(m, fabsf(v[t][0])), fabsf(v[t][1]));   // v_max3_f32 with |.| modifiers
    m = fmaxf(fmaxf(m, fabsf(v[t][2])), fabsf(v[t][3]));
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  return fmaxf(m, __shfl_xor(m, 32, 64));
}

// Running magnitude bounds (chain.h: kBoundWidth).  Every compute wave keeps, per stage, the largest |value| of the rows
// it has processed in a private row of 16 LDS words (no other wave touches it: no synchronisation); after its last tile
// it writes them to its entry of the bound slots.  `m` is a row maximum (>= 0, the same in the four lanes of a row): max
// over the wave's 16 rows with DPP, then one lane updates the LDS word.  Bit patterns of non-negative floats order like
// integers (inf / nan rows publish inf / nan: the consumer's results are then inf / nan too, as in the reference).
__device__ __forceinline__ bool any_slot(float* const (&slots)[kMaxStages + 1]) {
  bool any = false;
#pragma unroll
  for (int k = 0; k <= kMaxStages; ++k) any |= slots[k] != nullptr;
  return any;
}
template <int NB>
__device__ __forceinline__ unsigned* bound_row(float4* lds, int wave, int lane, bool wanted) {
  if (!wanted) return nullptr;   // uniform
  unsigned* row = reinterpret_cast<unsigned*>(reinterpret_cast<float*>(lds) + Ring<NB>::SIDE_FLOATS) + wave * 16;
  if (lane < 16) row[lane] = 0u;
  return row;
}
__device__ __forceinline__ void note_amax(unsigned* brow, int stage, float m, int lane) {
  if (!brow) return;   // uniform: a launch without bound slots (inference) keeps no running bounds
  int v = __float_as_int(m);
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));   // row_shr:8
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));   // row_shr:4
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));   // row_shr:2
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));   // row_shr:1
  if (lane == 15) brow[stage] = max(brow[stage], unsigned(v));
}
__device__ __forceinline__ void flush_bounds(float* const* slots, int n, const unsigned* brow, int wave, int lane) {
  if (!brow) return;
  const int entry = int(blockIdx.x) * 8 + wave;
  for (int k = 0; k < n; ++k) {  // uniform
    if (!slots[k] || lane != 0) continue;
    if (entry < kBoundWidth - 1) slots[k][entry] = __uint_as_float(brow[k]);
    else atomicMax(reinterpret_cast<unsigned*>(slots[k]) + (kBoundWidth - 1), brow[k]);   // a part with more CUs than the slot was sized for
  }
}

// End of a stage: the accumulators hold sum (s_w W)(s_x x); un-scale by the exact power of two 2^-(k_x + k_w) and add
// the bias (header of the stage's last chunk, still in the ring: its slot is not overwritten before every compute wave
// has passed the next chunk barrier).  `fw` = exponent field of 2^-k_w (header of chunk 0), E = the row's (RowScale).
// The combined factor is a normal float unless a row or a matrix is tiny or huge beyond ~2^+-60: that (wave-uniform)
// case takes v_ldexp_f32, which is exact over the whole range.
template <int NB, bool BIAS>
__device__ __forceinline__ void finish_stage(f32x4 (&acc)[NB], int E, int fw, const float* hdr, int lane) {
  const int f = E + fw - 139;   // exponent field of 2^-(k_x + k_w) = 2^(E - 139) 2^(fw - 127)
  const float* hb = hdr + 4 * (lane >> 4);
  if (__builtin_amdgcn_ballot_w64(unsigned(f - 1) >= 254u) == 0) {
    const float inv = __uint_as_float(unsigned(f) << 23);
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      if (BIAS) {
        const float4 x = *reinterpret_cast<const float4*>(hb + 16 * t);
        acc[t] = f32x4{fmaf(acc[t][0], inv, x.x), fmaf(acc[t][1], inv, x.y), fmaf(acc[t][2], inv, x.z), fmaf(acc[t][3], inv, x.w)};
      } else {
        acc[t] *= inv;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BIAS) x = *reinterpret_cast<const float4*>(hb + 16 * t);
      acc[t] = f32x4{ldexpf(acc[t][0], f - 127) + x.x, ldexpf(acc[t][1], f - 127) + x.y, ldexpf(acc[t][2], f - 127) + x.z,
                     ldexpf(acc[t][3], f - 127) + x.w};
    }
  }
}

// bf16 precision (chain.h): operands are the bf16 roundings of `act` and of the weights (one plane per pack, bias in the
// header of chunk 0), ONE product per fragment pair; `store_base` receives bf16 rows.
template <int NB>
__device__ __forceinline__ void mfma_stage_bf(f32x4 (&acc)[NB], const f32x4 (&act)[NB], float4* lds, Slot& slot, int lane,
                                              bool from_header, float* store_base, int64_t store_off, int64_t mask_rows) {
  using R = Ring<NB, 1>;
  u32x4 bb[NB / 2];
  round_block<NB>(act, 0, bb[0]);
  const bool st = store_base != nullptr && store_off >= 0;
#pragma unroll
  for (int c = 0; c < R::NCH; ++c) {
    lds_barrier();
    const float4* cur = lds + slot.i * R::CH4;
    if (++slot.i == slot.nr) slot.i = 0;
    if (c == 0 && from_header) {
      const float* bl_ = reinterpret_cast<const float*>(cur);
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const float4 x = *reinterpret_cast<const float4*>(bl_ + 16 * t + 4 * (lane >> 4));
        acc[t] = f32x4{x.x, x.y, x.z, x.w};
      }
    }
    if (st) {   // two 16-feature blocks of the saved activation per chunk, spread over the stage like the fp32 pairs
      store_block_bf16<NB>(act, store_base, store_off / (NB * 16), lane >> 4, 2 * c);
      store_block_bf16<NB>(act, store_base, store_off / (NB * 16), lane >> 4, 2 * c + 1);
    }
    const float4* body = cur + kChunkHdrFloats / 4 + lane;
#pragma unroll
    for (int t = 0; t < NB; t += 2) {
      const float4 h0 = body[t * 64], h1 = body[(t + 1) * 64];   // one plane per pack
      acc[t] = mma_bf(h0, bb[c], acc[t]);
      acc[t + 1] = mma_bf(h1, bb[c], acc[t + 1]);
      if (t == 0) {
        if (c + 1 < R::NCH) round_block<NB>(act, c + 1, bb[c + 1]);
        else if (mask_rows) store_mask_bits<NB, true>(act, store_base, mask_rows, store_off, lane >> 4);
      }
    }
  }
}

// One Linear on the compute waves (fp32 path): the three fp16 partial products per fragment pair, chain.h.
//   ZERO: the accumulators start from zero (else: they continue a sum begun by the previous call with the SAME row scale
//         and a pack of the same weight scale -- the two halves of a Linear over concatenated inputs, PackDesc::mate)
//   FIN:  0 leave the raw scaled sums (the next call continues them), 1 un-scale, 2 un-scale and add the pack's bias
// `slot` = ring slot of the stage's first chunk (advanced here).  `rs`: scale of this lane's row (scale_of(row_amax)).
// `store_base` (nullable, uniform) + `row` / `nrows`: HBM tensor that receives `act` as streaming 128-byte pairs, one
// pair per chunk, + its ReLU sign bits when `mask_rows`.  All compute waves of the workgroup must call this together.
template <int NB, bool ZERO, int FIN, bool LONE = false, bool TIMED = false>
__device__ __forceinline__ void mfma_stage(f32x4 (&acc)[NB], const f32x4 (&act)[NB], const RowScale rs, float4* lds, Slot& slot,
                                           int lane, float* store_base = nullptr, int64_t store_off = -1,
                                           int store_mode = 0, int64_t mask_rows = 0,
                                           unsigned long long* waited = nullptr, int64_t row = 0, int64_t nrows = 0) {
  using R = Ring<NB>;
  // The per-element VALU work of a stage (two-way split, sign bits) is spread over the chunks instead of sitting in
  // front of the first MFMA: only K block 0 is split up front, block c + 1 is split in the shadow of chunk c's MFMAs.
  u32x4 bh[NB / 2], bl[NB / 2];
  split_block<NB>(act, 0, rs.s, bh[0], bl[0]);
#ifdef BSMS_EXPERIMENTS
  const bool paired = (store_mode == 1 || store_mode == 2) && nrows > 0;   // saved tensors: 128-byte pieces, one pair per chunk
  const bool streaming = store_mode == 1;
  if (!paired) store_rows<NB, false>(act, store_base, store_off, lane >> 4, store_mode);
#else
  // production: a saved tensor is always written as streaming 128-byte pairs (callers pass nrows > 0 with every
  // store_base); the store-mode switches exist in experiment builds only
  constexpr bool paired = true, streaming = true;
  (void)store_mode;
  if (nrows <= 0) store_base = nullptr;
#endif
  int fw = 0;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if constexpr (LONE && NB == 8) {
    // ---- single-round launches: one wave per SIMD, nothing hides its stalls.  Measured per stage of the node MLP at the
    // coarse levels (profiles/lone_timeline.py): 750 cycles per chunk for 24 MFMAs (408 back to back) = the chunk barrier
    // (~100) + the LDS round trip of the first fragments (~250: reads of a chunk cannot be issued before its barrier) +
    // the MFMAs.  So the chunks are software-pipelined: the wave passes the barrier of chunk c + 1 and requests its
    // fragments BEFORE the last MFMAs of chunk c, which run from registers -- the round trip is exposed once per stage
    // instead of once per chunk.  Barrier count and order are
    // unchanged (one per chunk), and the loader's ring protocol holds: after barrier c + 1 it may overwrite the slot of
    // chunk c, of which this wave has nothing left to read (its fragments are in registers; only the LAST chunk's header
    // -- the bias -- is read after its MFMAs, and the barrier after the last chunk belongs to the next stage).
    // The three products of an accumulator keep their order (h_w l_x, h_w h_x, l_w h_x): bit-identical results.
    // Rolling half-chunk window, 64 fragment registers: feature blocks 0-3 of a chunk ("g0") and 4-7 ("g1") each have
    // one register set; g1 of chunk c is requested before the MFMAs of g0, then the barrier of chunk c + 1 is passed and
    // ITS g0 requested into the registers g0 of chunk c has just freed, under the MFMAs of g1.
    float4 g0[NB], g1[NB];          // [2 * k + plane] for feature block k (g0) / 4 + k (g1)
    const float4* hdr_last = nullptr;
    const float4* body = nullptr;
    auto pass_barrier = [&]() {
      if (TIMED) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        lds_barrier();
        *waited += __builtin_amdgcn_s_memtime() - t0;
      } else {
        lds_barrier();
      }
    };
    auto next_chunk = [&]() {        // the ring's next chunk becomes the current one; its first half is requested
      hdr_last = lds + slot.i * R::CH4;
      if (++slot.i == slot.nr) slot.i = 0;
      body = hdr_last + kChunkHdrFloats / 4 + lane;
#pragma unroll
      for (int k = 0; k < NB; ++k) g0[k] = body[k * 64];
    };
    pass_barrier();
    next_chunk();
    fw = int(__float_as_uint(reinterpret_cast<const float*>(hdr_last)[kScaleSlot]) >> 23);
#pragma unroll
    for (int c = 0; c < R::NCH; ++c) {
#pragma unroll
      for (int k = 0; k < NB; ++k) g1[k] = body[(NB + k) * 64];
      __builtin_amdgcn_sched_barrier(0);
      if (paired) store_pair_stream<NB>(act, store_base, row, nrows, lane, 2 * c, streaming);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = mma(g0[2 * k], bl[c], (ZERO && c == 0) ? zero : acc[k]);
      // VALU work for later, placed among this chunk's MFMAs
      if (c + 1 < R::NCH) split_block<NB>(act, c + 1, rs.s, bh[c + 1], bl[c + 1]);
      else if (mask_rows) store_mask_bits<NB>(act, store_base, mask_rows, store_off, lane >> 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = mma(g0[2 * k], bh[c], acc[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = mma(g0[2 * k + 1], bh[c], acc[k]);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < R::NCH) {
        pass_barrier();               // chunk c + 1 has landed (its lgkmcnt(0): g1 of chunk c is in registers from here on)
        next_chunk();                 // g0 <- first half of chunk c + 1 (the registers the MFMAs above have released)
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[4 + k] = mma(g1[2 * k], bl[c], (ZERO && c == 0) ? zero : acc[4 + k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[4 + k] = mma(g1[2 * k], bh[c], acc[4 + k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[4 + k] = mma(g1[2 * k + 1], bh[c], acc[4 + k]);
    }
    if (FIN != 0) finish_stage<NB, FIN == 2>(acc, rs.E, fw, reinterpret_cast<const float*>(hdr_last), lane);
    return;
  }
#pragma unroll
  for (int c = 0; c < R::NCH; ++c) {
    if (TIMED) {   // experiments: cycles this wave spends waiting at the chunk barriers
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      lds_barrier();
      *waited += __builtin_amdgcn_s_memtime() - t0;
    } else {
      lds_barrier();                                         // chunk has landed (and my reads of the last one are done)
    }
    const float4* cur = lds + slot.i * R::CH4;
    if (++slot.i == slot.nr) slot.i = 0;
    if (c == 0) fw = int(__float_as_uint(reinterpret_cast<const float*>(cur)[kScaleSlot]) >> 23);
    if (paired) store_pair_stream<NB>(act, store_base, row, nrows, lane, 2 * c, streaming);
    const float4* body = cur + kChunkHdrFloats / 4 + lane;
    // Two accumulators interleaved so that back-to-back MFMAs are independent.
    if constexpr (LONE) {
      // (D = 256; D = 128 takes the pipelined path above: the fragments of whole chunks do not fit next to 64 + 64 activation / accumulator registers)
      // pairs are read ONE PAIR AHEAD (f = current, n = next), pinned by sched_barrier.
      float4 f0 = body[0], f1 = body[2 * 64];                                      // (t = 0, plane h)
#pragma unroll
      for (int t = 0; t < NB; t += 2) {
        float4 n0 = body[(t * 2 + 1) * 64], n1 = body[(t * 2 + 3) * 64];          // plane l of (t, t + 1)
        __builtin_amdgcn_sched_barrier(0);
        acc[t] = mma(f0, bl[c], (ZERO && c == 0) ? zero : acc[t]);
        acc[t + 1] = mma(f1, bl[c], (ZERO && c == 0) ? zero : acc[t + 1]);
        acc[t] = mma(f0, bh[c], acc[t]);
        acc[t + 1] = mma(f1, bh[c], acc[t + 1]);
        if (t == 0) {   // VALU work for later, placed among this chunk's MFMAs
          if (c + 1 < R::NCH) split_block<NB>(act, c + 1, rs.s, bh[c + 1], bl[c + 1]);
          else if (mask_rows) store_mask_bits<NB>(act, store_base, mask_rows, store_off, lane >> 4);
        }
        f0 = n0;
        f1 = n1;
        if (t + 2 < NB) {                                                           // plane h of the next block pair
          n0 = body[((t + 2) * 2) * 64];
          n1 = body[((t + 2) * 2 + 2) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[t] = mma(f0, bh[c], acc[t]);
        acc[t + 1] = mma(f1, bh[c], acc[t + 1]);
        f0 = n0;
        f1 = n1;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NB; t += 2) {
        {
          const float4 h0 = body[(t * 2 + 0) * 64], h1 = body[(t * 2 + 2) * 64];
          acc[t] = mma(h0, bl[c], (ZERO && c == 0) ? zero : acc[t]);
          acc[t + 1] = mma(h1, bl[c], (ZERO && c == 0) ? zero : acc[t + 1]);
          acc[t] = mma(h0, bh[c], acc[t]);
          acc[t + 1] = mma(h1, bh[c], acc[t + 1]);
        }
        if (t == 0) {   // VALU work for later, placed among this chunk's MFMAs
          if (c + 1 < R::NCH) split_block<NB>(act, c + 1, rs.s, bh[c + 1], bl[c + 1]);
          else if (mask_rows) store_mask_bits<NB>(act, store_base, mask_rows, store_off, lane >> 4);  // saved activation: + sign bits
        }
        {
          const float4 l0 = body[(t * 2 + 1) * 64], l1 = body[(t * 2 + 3) * 64];
          acc[t] = mma(l0, bh[c], acc[t]);
          acc[t + 1] = mma(l1, bh[c], acc[t + 1]);
        }
      }
    }
    if (FIN != 0 && c == R::NCH - 1) finish_stage<NB, FIN == 2>(acc, rs.E, fw, reinterpret_cast<const float*>(cur), lane);
  }
}

// ReLU as ONE integer operation per element: a float with its sign bit set (negative, -0) is a negative int32, so
// max(bits, 0) is +0 and everything else is left alone -- the bits of fmaxf(x, 0) for every non-NaN input, and a NaN
// stays a NaN as in torch.relu (fmaxf would return 0).  fmaxf compiles to TWO v_max_f32 per element whenever hipcc cannot prove
// its input canonical (the un-scaled accumulators: 64 operations per stage in k_edge_fwd; same-box effect on the fp32 step: none,
// profiles/r05_f32_valu.txt -- kept for the NaN semantics and the halved instruction count).
template <int NB>
__device__ __forceinline__ void relu_into(f32x4 (&dst)[NB], const f32x4 (&srcv)[NB]) {
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[t][r] = __int_as_float(max(__float_as_int(srcv[t][r]), 0));
}

// v += scale * vec (vec in lane feature order)
template <int NB>
__device__ __forceinline__ void axpy_features(f32x4 (&v)[NB], const float* vec, float scale, int g) {
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const float4 w = *reinterpret_cast<const float4*>(vec + 16 * t + 4 * g);
    v[t][0] = fmaf(scale, w.x, v[t][0]);
    v[t][1] = fmaf(scale, w.y, v[t][1]);
    v[t][2] = fmaf(scale, w.z, v[t][2]);
    v[t][3] = fmaf(scale, w.w, v[t][3]);
  }
}

template <int NB>
__device__ __forceinline__ float dot_features(const f32x4 (&v)[NB], const float* vec, int g) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const float4 w = *reinterpret_cast<const float4*>(vec + 16 * t + 4 * g);
    s = fmaf(v[t][0], w.x, s);
    s = fmaf(v[t][1], w.y, s);
    s = fmaf(v[t][2], w.z, s);
    s = fmaf(v[t][3], w.w, s);
  }
  return group_sum(s);
}

// row of the [B, E] edge tensor -> (batch b, edge q) without the 64-bit division hipcc would emit (~100 instructions
// per lane and tile): float reciprocal estimate, corrected by at most one step either way (rows < 2^31, launcher).
struct EdgeRef { int b, q; };
__device__ __forceinline__ EdgeRef edge_ref(unsigned row, unsigned E, float rcpE) {
  int b = int(float(row) * rcpE);
  int q = int(row) - b * int(E);
  if (q < 0) { q += int(E); --b; }
  if (q >= int(E)) { q -= int(E); ++b; }
  return EdgeRef{b, q};
}

}  // namespace
