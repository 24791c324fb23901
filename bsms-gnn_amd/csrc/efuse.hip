// Fused backward of the edge MLP for the bf16 precisions (BSMS_BF16 / BSMS_BF16_NODES), D = 128, hidden = 3:
// forward RECOMPUTE + LayerNorm backward + dgrad chain + the weight gradients of the three D x D Linears in ONE kernel.
// Reference arithmetic: src/ops/basic.py:6-23 (MLP), :90-94 (edge message + scatter), under trainer/trainer.py:146-147.
//
// Why: the unfused backward wrote every layer gradient gE[1..H] and the forward every
// activation a_0..a_{H-1} to HBM only so that a separate split-K kernel could read them back: at airfoil level 0 that is
// 192 MB + 192 MB written and 384 MB re-read per block for tensors whose values the chain kernels had in registers.  Here
// nothing of that touches HBM: the forward saves only the messages y, rstd and the 16-byte fiber of every edge; this
// kernel re-creates a_0 (gather of the two node projections + fiber), a_1, a_2 on the matrix cores (two extra Linears:
// the matrix pipe was 20 % busy), runs the gradient chain, and accumulates dW_l += G_l^T A_{l-1} on chip.
//
// Shape of a workgroup (512 threads, one per CU, persistent over 64-row tiles):
//   waves 0-3  "chain" waves: 16 edge rows each in the chain layout of chain.h (lane <-> row, features in registers);
//              per tile: gather -> a_0 -> a_1 -> a_2 (bit-identical to the forward kernel: same operands, same product
//              order), LayerNorm backward -> g_3, then three times { hand (G_l, A_{l-1}) of its 16 rows to LDS; dgrad
//              through W_l, masked by a_{l-1} > 0 }, finally g_0 -> HBM as bf16 (the scatter kernel's input).
//   waves 4-7  "gradient" waves: hold ALL of dW_1..3 for the workgroup's rows in registers (3 x 128 x 128 fp32 = 192
//              accumulator registers per lane: wave (i, j) owns rows 64 i .., columns 64 j .. of each) and multiply the
//              staged 64-row tiles: the reduction index is the ROW, so operands are column fragments --
//              ds_read_b64_tr_b16 from the row-major staging tiles, as wgrad.hip does from its planes.  They also sum
//              the bias gradients (column sums of G_l) from the staged rows.
//   One chain wave and one gradient wave share a SIMD: the VALU / LDS phases of the one run under the MFMAs of the other.
// Weights: W_1..W_3 as bf16 (rounded once per call by the prepack, like the forward's packs) stay RESIDENT in LDS for the whole launch, row-major
// [n][k] with a 288-byte pitch and an 8-byte-piece XOR swizzle (piece ^= (row >> 2) & 3).  ONE copy serves both
// directions: the forward A-fragments are two ds_read_b64 per lane, the transposed (dgrad) fragments two
// ds_read_b64_tr_b16 -- the K-slot -> feature map of chain.h is exactly what the transposing read delivers.  No loader
// wave, no ring, no chunk barriers.  LDS: 3 x 36 KB weights + 2 x 18 KB staging + 3 KB side tables = 147 KB.
// Determinism: every workgroup writes its partial dW / db once; k_ef_reduce sums them in a fixed order (no atomics).
#include "chain.h"

#pragma clang fp contract(off)

using namespace bsms;

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int D = 128, NB = 8;
constexpr int ROWB = 288;                     // LDS row pitch in bytes: 128 bf16 + 32 bytes (8 banks further per row)
constexpr int W_BYTES = D * ROWB;             // one weight matrix
constexpr int ST_ROWS = 64;                   // rows of a tile = 4 chain waves x 16
constexpr int ST_BYTES = ST_ROWS * ROWB;
constexpr int OFF_G = 3 * W_BYTES, OFF_A = OFF_G + ST_BYTES, OFF_SIDE = OFF_A + ST_BYTES;
constexpr int SIDE_WFT = 0, SIDE_B1 = 4 * D * 4, SIDE_B2 = SIDE_B1 + D * 4;
constexpr int LDS_BYTES = OFF_SIDE + SIDE_B2 + D * 4;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(W_BYTES % 1024 == 0, "whole LDS-DMA pieces");
constexpr int DW_FLOATS = 3 * D * D, DB_FLOATS = 3 * 4 * D;   // partials of one workgroup

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_dst, +1 KB) in lane order (chain.hip: glds16)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// byte offset of the 8-byte piece `piece` (4 consecutive bf16 columns) of row `row` in a swizzled [rows][144] bf16 tile
__device__ __forceinline__ unsigned piece_off(int row, int piece) { return unsigned(row * ROWB + 8 * (piece ^ ((row >> 2) & 3))); }

// forward A-fragment (t, c) of a resident weight matrix: lane (n = lane & 15, g = lane >> 4) needs W[16 t + n][k] for the K
// slots of chain.h, k = 32 c + 4 g + {0..3} and 32 c + 16 + 4 g + {0..3}: two 8-byte pieces, 32 bytes apart
__device__ __forceinline__ u32x4 frag_fwd(const char* W, unsigned fb, int t, int c) {
  const u32x2 lo = *reinterpret_cast<const u32x2*>(W + fb + t * (16 * ROWB) + c * 64);
  const u32x2 hi = *reinterpret_cast<const u32x2*>(W + fb + t * (16 * ROWB) + c * 64 + 32);
  return u32x4{lo[0], lo[1], hi[0], hi[1]};
}
// column fragment of a swizzled row-major tile: lane (col = lane & 15 of the 16-column block `cb`, g = lane >> 4) receives
// rows 32 kb + 4 g + {0..3} (slots 0-3) and 32 kb + 16 + 4 g + {0..3} (slots 4-7) of its column.  ds_read_b64_tr_b16:
// within a 16-lane group lane i' supplies the 8 bytes at its address and lane i receives element (i & 3) of suppliers
// 4 k + (i >> 2), k = 0..3 (wgrad.hip: column_fragment; profiles/census/tr_test.hip).  `tb` = this lane's supplier base
// (4 g + (i' >> 2)) * ROWB + 8 * ((i' & 3) ^ g); the swizzle term of those rows is g for both halves.
__device__ __forceinline__ u32x4 frag_col(const char* T, unsigned tb, int kb, int cb) {
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  const char* p = T + tb + kb * (32 * ROWB) + cb * 32;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * ROWB));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(u32x4, v);
}

__device__ __forceinline__ float group_sum(float s) {  // sum over the 4 lane groups holding one row
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
__device__ __forceinline__ float row_sum(const f32x4 (&v)[NB]) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NB; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
  return group_sum(s);
}
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
// relu + round to bf16: the B operand of the next Linear (chain.hip: round_block) -- and the saved activation a_l itself
__device__ __forceinline__ void relu_pack(u32x4 (&bb)[4], const f32x4 (&acc)[NB]) {
  // round first, then clamp the PAIR with one packed signed-integer max: a bf16 with its sign bit set (a negative value or
  // -0) is a negative int16, so max(., 0) is +0, and a non-negative one is left alone -- the bits of pk_bf16(max(x, 0), max(y, 0))
  // for every non-NaN input (rounding keeps the sign), in two operations per pair instead of three
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const unsigned w = pk_bf16(acc[2 * c + (v >> 1)][2 * (v & 1)], acc[2 * c + (v >> 1)][2 * (v & 1) + 1]);
      asm("v_pk_max_i16 %0, %1, 0" : "=v"(bb[c][v]) : "v"(w));
    }
}
__device__ __forceinline__ void pack(u32x4 (&bb)[4], const f32x4 (&acc)[NB]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) bb[c][v] = pk_bf16(acc[2 * c + (v >> 1)][2 * (v & 1)], acc[2 * c + (v >> 1)][2 * (v & 1) + 1]);
}
// g = acc where the (post-ReLU, bf16) activation is positive, else 0   (ReLU backward; a > 0 <=> its bf16 bits != 0)
__device__ __forceinline__ void mask_by(f32x4 (&g)[NB], const f32x4 (&acc)[NB], const u32x4 (&ap)[4]) {
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned w = ap[t >> 1][2 * (t & 1) + (j >> 1)];
      const bool pos = (j & 1) ? (w > 0xffffu) : ((w & 0xffffu) != 0u);
      g[t][j] = pos ? acc[t][j] : 0.f;
    }
}
// mask_by + pack in one: the packed bf16 gradient pair times a 0 / 1 half-word made from the packed activation pair -- an
// INTEGER multiply of the bit pattern (x * 1 = x, x * 0 = +0): two packed operations per PAIR instead of compare + select per
// element; the bits of pack(mask_by(.)).  (inline asm: left to itself hipcc reasons about the bf16 conversion that made `ap`
// and emits ten compares per pair)
__device__ __forceinline__ void mask_pack(u32x4 (&gb)[4], const f32x4 (&acc)[NB], const u32x4 (&ap)[4]) {
  const unsigned ones = 0x00010001u;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      unsigned m;
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(ap[c][v]), "s"(ones));   // 1 where a > 0 (a >= +0: bits != 0)
      const unsigned w = pk_bf16(acc[2 * c + (v >> 1)][2 * (v & 1)], acc[2 * c + (v >> 1)][2 * (v & 1) + 1]);
      asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(gb[c][v]) : "v"(w), "v"(m));
    }
}
// acc = bias + W x   (forward direction; each accumulator takes its K chunks in the order 0..3 like chain.hip: mfma_stage_bf)
__device__ __forceinline__ void stage_fwd(f32x4 (&acc)[NB], const u32x4 (&bb)[4], const char* W, const float* bias, unsigned fb, int g) {
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const float4 x = *reinterpret_cast<const float4*>(bias + 16 * t + 4 * g);
    acc[t] = f32x4{x.x, x.y, x.z, x.w};
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = mma(frag_fwd(W, fb, t, c), bb[c], acc[t]);
}
// acc = W^T g   (dgrad through the same resident copy)
__device__ __forceinline__ void stage_bwd(f32x4 (&acc)[NB], const u32x4 (&gb)[4], const char* W, unsigned tb) {
#pragma unroll
  for (int t = 0; t < NB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = mma(frag_col(W, tb, c, t), gb[c], acc[t]);
}
// this wave's 16 rows of a bf16 tensor (already packed as B operands) -> staging tile
__device__ __forceinline__ void stage_rows(char* T, unsigned sb, const u32x4 (&bb)[4]) {
#pragma unroll
  for (int t = 0; t < NB; ++t)
    *reinterpret_cast<u32x2*>(T + sb + t * 32) = u32x2{bb[t >> 1][2 * (t & 1)], bb[t >> 1][2 * (t & 1) + 1]};
}

template <int NBX>
__device__ __forceinline__ void load_rows(f32x4 (&v)[NBX], const float* row, int g) {
#pragma unroll
  for (int t = 0; t < NBX; ++t) v[t] = *reinterpret_cast<const f32x4*>(row + 16 * t + 4 * g);
}

// experiments (profiles/ef_timeline.py): phase stamps of chain wave 0 (slots 0-23) / gradient wave 4 (24-39), 40 slots per tile
#ifdef BSMS_EXPERIMENTS
// Stamps of the gradient wave are OFF unless -DEFV_GSTAMP: its stamp stores queue behind the chain waves' gathers and stall it for
// ~2k cycles per tile, for which the chain waves then wait at barrier X2 -- the probe was 10 % of the tile
#ifdef EFV_GSTAMP
#define EF_STAMP_WAVES (wave == 0 || wave == 4)
#else
#define EF_STAMP_WAVES (wave == 0)
#endif
#define EF_STAMP(slot)                                                                                          \
  do {                                                                                                           \
    if (a.timing && lane == 0 && EF_STAMP_WAVES) {                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
      a.timing[size_t(int(blockIdx.x) + it * int(gridDim.x)) * 40 + (wave == 4 ? 24 : 0) + (slot)] = __builtin_amdgcn_s_memtime(); \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
    }                                                                                                            \
  } while (0)
#else
#define EF_STAMP(slot) do {} while (0)
#endif

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_edge_fused_bwd(EdgeFusedBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- prologue: the three PACK_ROWS_BF16 images (bf16, swizzled row-major: written once per call by the block's prepack) go
  // straight from L2 into LDS with LDS-DMA, 108 pieces of 1 KB over the eight waves -- one round trip instead of the six
  // dependent batches of fp32 loads + conversion the first version paid per launch (~4 us of a 29 us coarse-level launch)
  {
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    for (int piece = uwave; piece < 3 * (W_BYTES / 1024); piece += 8) {
      const int l = piece / (W_BYTES / 1024), k = piece - l * (W_BYTES / 1024);
      const char* src = reinterpret_cast<const char*>(a.wr[l]) + k * 1024 + lane * 16;
      glds16(src, __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + unsigned(piece) * 1024u));
    }
  }
  {
    float* side = reinterpret_cast<float*>(lds + OFF_SIDE);
    for (int o = tid; o < 4 * D; o += 512) side[SIDE_WFT / 4 + o] = o < (a.p + 1) * D ? a.wft[o] : 0.f;
    if (tid < D) side[SIDE_B1 / 4 + tid] = a.b[0][tid];
    else if (tid < 2 * D) side[SIDE_B2 / 4 + tid - D] = a.b[1][tid - D];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int my_tiles = (a.ntiles - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
  const char* const W1 = lds, * const W2 = lds + W_BYTES, * const W3 = lds + 2 * W_BYTES;
  char* const GST = lds + OFF_G;
  char* const AST = lds + OFF_A;
  const int r = lane & 15, g = lane >> 4;
  const unsigned tb = unsigned((4 * g + (r >> 2)) * ROWB + 8 * ((r & 3) ^ g));   // supplier base of the transposing reads

  if (wave >= 4) {
    // =================================================================================== gradient waves
    const int gw = wave - 4, gi = gw >> 1, gj = gw & 1;
    f32x4 dw[3][4][4];
    float db[3][2];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      db[l][0] = db[l][1] = 0.f;
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) dw[l][x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // bias gradient: column sums of the staged G rows 16 gw .. 16 gw + 15 (lane <-> columns 2 lane, 2 lane + 1).  The 16 words are
    // READ right after the fragments (they must be out of the staging tile before barrier X); unpacking and adding them happens
    // after that barrier, off the path the chain waves wait on (hand-over to hand-over the chain needs 1.1k cycles, this
    // wave's reads + MFMAs + sums took 1.4k: the chain waited 0.6k at X twice per tile)
    unsigned cs[16];
    auto add_sums = [&](float (&acc2)[2]) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        s0 += bf16_lo(cs[rr]);
        s1 += bf16_hi(cs[rr]);
      }
      acc2[0] += s0;
      acc2[1] += s1;
    };
    for (int it = 0; it < my_tiles; ++it) {
#pragma unroll
      for (int l = 2; l >= 0; --l) {
        EF_STAMP(4 * (2 - l));
        lds_barrier();   // X: the chain waves may overwrite the staging tiles (everybody is done with the previous pair)
        EF_STAMP(4 * (2 - l) + 1);
        if (l == 2) { if (it > 0) add_sums(db[0]); }   // the previous hand-over's rows
        else add_sums(db[l + 1]);
        lds_barrier();   // Y: (G_{l+1}, A_l) of this tile are staged
        EF_STAMP(4 * (2 - l) + 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 gf[4], af[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) gf[x] = frag_col(GST, tb, ks, 4 * gi + x);
#pragma unroll
          for (int y = 0; y < 4; ++y) af[y] = frag_col(AST, tb, ks, 4 * gj + y);
#pragma unroll
          for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) dw[l][x][y] = mma(gf[x], af[y], dw[l][x][y]);
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
          cs[rr] = *reinterpret_cast<const unsigned*>(GST + piece_off(16 * gw + rr, lane >> 1) + 4 * (lane & 1));
        EF_STAMP(4 * (2 - l) + 3);
      }
    }
    if (my_tiles > 0) add_sums(db[0]);
    // partial results of this workgroup: dW[l][n][k] (lane holds rows n = 64 gi + 16 x + 4 g + j, column k = 64 gj + 16 y + r)
    float* part = a.part + size_t(blockIdx.x) * (DW_FLOATS + DB_FLOATS);
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            part[l * D * D + (64 * gi + 16 * x + 4 * g + j) * D + 64 * gj + 16 * y + r] = dw[l][x][y][j];
      float* pdb = part + DW_FLOATS + (l * 4 + gw) * D;
      *reinterpret_cast<float2*>(pdb + 2 * lane) = make_float2(db[l][0], db[l][1]);
    }
    return;
  }

  // ======================================================================================= chain waves
  const int s = (r >> 2) & 3;
  const unsigned fb = unsigned(r * ROWB + 8 * (g ^ s));                    // base of the forward A-fragments
  const unsigned sb = unsigned((16 * wave + r) * ROWB + 8 * (g ^ s));      // base of this lane's staging pieces
  const float* const wft = reinterpret_cast<const float*>(lds + OFF_SIDE + SIDE_WFT);
  const float* const b1 = reinterpret_cast<const float*>(lds + OFF_SIDE + SIDE_B1);
  const float* const b2 = reinterpret_cast<const float*>(lds + OFF_SIDE + SIDE_B2);
  const unsigned uE = unsigned(a.E), uN = unsigned(a.N);
  float gmax = 0.f;

  // what a tile needs from the plan: node rows of the two endpoints and the edge row itself
  struct Where { unsigned row, srow; bool live; unsigned isrc, idst; };
  const float rcpE = 1.f / float(a.E);
  auto locate = [&](int tile) {
    Where wq;
    const int64_t row64 = int64_t(tile) * ST_ROWS + wave * 16 + r;
    wq.live = row64 < a.R;
    wq.srow = unsigned(row64);                 // where this lane STORES: rows past R are padding of g0 (chain.h: pad_rows), never read
    wq.row = wq.live ? unsigned(row64) : 0u;   // lanes past the end read row 0; nothing of theirs is summed
    // row -> (batch, edge) without an integer division: reciprocal estimate, corrected by at most one step (rows < 2^31, launcher)
    int b = int(float(wq.row) * rcpE);
    int q = int(wq.row) - b * int(uE);
    if (q < 0) { q += int(uE); --b; }
    if (q >= int(uE)) { q -= int(uE); ++b; }
    if (unsigned(q) >= uE) { b = int(wq.row / uE); q = int(wq.row - unsigned(b) * uE); }   // estimate off by more than one (tiny E, huge B): exact
    wq.isrc = unsigned(b) * uN + unsigned(a.src[q]);
    wq.idst = unsigned(b) * uN + unsigned(a.dst[q]);
    return wq;
  };
  f32x4 ps[NB], pd[NB];
  float4 fib;
  Where cur = locate(int(blockIdx.x));
  load_rows<NB>(ps, a.Ps + size_t(cur.isrc) * D, g);
  load_rows<NB>(pd, a.Pd + size_t(cur.idst) * D, g);
  fib = *reinterpret_cast<const float4*>(a.fiber + size_t(cur.row) * 4);
  // g_0 of the PREVIOUS tile, packed: its eight stores are issued two at a time between the phases of the next tile.  Issued
  // right after dgrad 1 they (a) fill the CU's store path in one burst and (b) sit, in the in-order vmcnt queue, between the
  // endpoint rows requested a tile ahead and their first use, which then waits for the stores to be acknowledged
  // (profiles/r05_ef_timeline_before.txt: 3.2k + 3.0k of 22k cycles per tile).
  u32x4 g0p[4] = {};
  unsigned g0row = 0;
  auto store_g0 = [&](int t0) {   // feature blocks t0, t0 + 1 of the previous tile's rows
    u32x2* op = reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(a.g0) + size_t(g0row) * D + 4 * g);
    op[4 * t0] = u32x2{g0p[t0 >> 1][0], g0p[t0 >> 1][1]};
    op[4 * t0 + 4] = u32x2{g0p[t0 >> 1][2], g0p[t0 >> 1][3]};
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int it = 0; it < my_tiles; ++it) {
    const int tile = int(blockIdx.x) + it * int(gridDim.x);
    EF_STAMP(0);
    // ---- requests of this tile's gradient inputs (consumed after the two forward Linears) and of the next tile's endpoints
    f32x4 dy[NB];
    u32x2 yb[NB];
    load_rows<NB>(dy, a.dy + size_t(cur.idst) * D, g);
    {
      const u32x2* yp = reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(a.y) + size_t(cur.row) * D + 4 * g);
#pragma unroll
      for (int t = 0; t < NB; ++t) yb[t] = yp[4 * t];
    }
    const float rstd = a.rstd[cur.row];
    const bool more = it + 1 < my_tiles;
    const Where nxt = locate(more ? tile + int(gridDim.x) : tile);
    // ---- a_0 = relu(Ps[src] + Pd[dst] + Wf . fiber)   (chain.hip: k_chain_fwd IN_EDGE, same operations in the same order)
    f32x4 act[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) act[t] = ps[t] + pd[t];
    {
      const float fv[4] = {fib.x, fib.y, fib.z, fib.w};
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (c < a.p) axpy_features(act, wft + c * D, fv[c], g);
      const float nrm = a.p == 1 ? fib.y : (a.p == 2 ? fib.z : fib.w);
      axpy_features(act, wft + a.p * D, nrm, g);
    }
    u32x4 a0p[4], a1p[4], a2p[4];
    relu_pack(a0p, act);
    EF_STAMP(1);
    if (it > 0) store_g0(0);
    f32x4 acc[NB];
    stage_fwd(acc, a0p, W1, b1, fb, g);
    relu_pack(a1p, acc);
    EF_STAMP(2);
    if (it > 0) store_g0(2);
    stage_fwd(acc, a1p, W2, b2, fb, g);
    relu_pack(a2p, acc);
    EF_STAMP(3);
    if (it > 0) store_g0(4);
    // ---- LayerNorm backward (no affine): g_3 = rstd * (dy - mean(dy) - y * mean(dy * y))   (chain.hip: k_chain_bwd), written as
    // two fused multiply-adds per element, g_3 = fma(-c1, y, fma(c0, dy, c2)) with c0 = rstd, c1 = rstd * mean(dy y),
    // c2 = -rstd * mean(dy): a quarter of the VALU work of the literal form (the chain waves are issue-bound: every VALU
    // operation delays the MFMAs behind it; the packed ReLU alone was worth 11 % of this kernel)
    f32x4 gr[NB];
    {
      f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        act[t] = f32x4{bf16_lo(yb[t][0]), bf16_hi(yb[t][0]), bf16_lo(yb[t][1]), bf16_hi(yb[t][1])};
        const f32x2 d0 = {dy[t][0], dy[t][1]}, d1 = {dy[t][2], dy[t][3]}, y0 = {act[t][0], act[t][1]}, y1 = {act[t][2], act[t][3]};
        sa += d0 + d1;
        sb = __builtin_elementwise_fma(d0, y0, sb);
        sb = __builtin_elementwise_fma(d1, y1, sb);
      }
      const float m1 = group_sum(sa[0] + sa[1]) * (1.f / D);
      const float m2 = group_sum(sb[0] + sb[1]) * (1.f / D);
      const float c0 = cur.live ? rstd : 0.f;   // rows past the end contribute nothing to dW / db
      const float c1 = c0 * m2, c2 = -(c0 * m1);
      const f32x4 v0 = {c0, c0, c0, c0}, v1 = {-c1, -c1, -c1, -c1}, v2 = {c2, c2, c2, c2};
#pragma unroll
      for (int t = 0; t < NB; ++t) gr[t] = __builtin_elementwise_fma(v1, act[t], __builtin_elementwise_fma(v0, dy[t], v2));
    }
    if (it > 0) store_g0(6);
    // ---- the next tile's endpoint rows are requested now: they land under the three gradient stages
    load_rows<NB>(ps, a.Ps + size_t(nxt.isrc) * D, g);
    load_rows<NB>(pd, a.Pd + size_t(nxt.idst) * D, g);
    fib = *reinterpret_cast<const float4*>(a.fiber + size_t(nxt.row) * 4);
    u32x4 gb[4];
    pack(gb, gr);
    EF_STAMP(4);
    // ---- one gradient stage: acc = W_l^T g_l, and in the middle of it (G_l, A_{l-1}) of this wave's 16 rows go to the gradient
    // waves.  The hand-over sits INSIDE the stage: barrier X (everybody has finished reading the previous pair) is reached two
    // K chunks after the previous stage's barrier Y, when the gradient waves are (nearly) done, and the staging stores land under
    // the third chunk's MFMAs instead of in front of an idle barrier (before: 3 x 1.25k cycles at Y, up to 2.4k at X per tile).
    auto chunk_bwd = [&](const char* W, int c) {
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] = mma(frag_col(W, tb, c, t), gb[c], acc[t]);
    };
#define EF_DGRAD(W, AP, K)                                                         \
    do {                                                                           \
      _Pragma("unroll") for (int t = 0; t < NB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; \
      chunk_bwd(W, 0);                                                             \
      chunk_bwd(W, 1);                                                             \
      EF_STAMP(5 + 4 * (K));                                                       \
      lds_barrier(); /* X */                                                       \
      EF_STAMP(6 + 4 * (K));                                                       \
      stage_rows(GST, sb, gb);                                                     \
      stage_rows(AST, sb, AP);                                                     \
      chunk_bwd(W, 2);                                                             \
      EF_STAMP(7 + 4 * (K));                                                       \
      lds_barrier(); /* Y */                                                       \
      EF_STAMP(8 + 4 * (K));                                                       \
      chunk_bwd(W, 3);                                                             \
    } while (0)
    EF_DGRAD(W3, a2p, 0);        // Linear 3: g_2 = (W_3^T g_3) . [a_2 > 0]
    mask_pack(gb, acc, a2p);
    EF_DGRAD(W2, a1p, 1);        // Linear 2
    mask_pack(gb, acc, a1p);
    EF_DGRAD(W1, a0p, 2);        // Linear 1
#undef EF_DGRAD
    EF_STAMP(17);
    // ---- g_0 (bf16: input of the scatter / fiber-gradient kernel) is kept packed for the deferred stores
    if (a.gmax) {   // uniform: somebody wants its magnitude bound (A/B builds with fp16 x 2 projections' weight gradients, gmp.hip)
      mask_by(gr, acc, a0p);
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        m = fmaxf(fmaxf(m, fabsf(gr[t][0])), fabsf(gr[t][1]));
        m = fmaxf(fmaxf(m, fabsf(gr[t][2])), fabsf(gr[t][3]));
      }
      gmax = fmaxf(gmax, m);
      pack(g0p, gr);
    } else {
      mask_pack(g0p, acc, a0p);
    }
    g0row = cur.srow;
    EF_STAMP(18);
    cur = nxt;
  }
  if (my_tiles > 0) {   // the last tile's g_0
    store_g0(0); store_g0(2); store_g0(4); store_g0(6);
  }
  if (a.gmax) {   // bound slot of gE[0] (chain.h: kBoundWidth): this wave's entry
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
    if (lane == 0) a.gmax[int(blockIdx.x) * 8 + wave] = gmax;
  }
}

// dW_l = sum over workgroups of their partials, in workgroup order per split and split order at the end (fixed: run-to-run
// reproducible).  Block = 32 float4 outputs x 8 splits of the workgroup range; the last blocks sum the bias gradients.
__global__ __launch_bounds__(256) void k_ef_reduce(const float* part, int nwg, float* dW0, float* dW1, float* dW2, float* db0, float* db1,
                                                   float* db2) {
  __shared__ float4 red[8][32];
  const int tid = threadIdx.x;
  constexpr int NBLK = DW_FLOATS / 4 / 32;   // 384
  if (int(blockIdx.x) < NBLK) {
    const int e = tid & 31, sp = tid >> 5;
    const int o4 = int(blockIdx.x) * 32 + e;     // float4 index into [3][D*D]
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = DW_FLOATS + DB_FLOATS;
    int w = sp;
    for (; w + 24 < nwg; w += 32) {   // four independent loads in flight
      const float4 v0 = *reinterpret_cast<const float4*>(part + size_t(w) * stride + 4 * o4);
      const float4 v1 = *reinterpret_cast<const float4*>(part + size_t(w + 8) * stride + 4 * o4);
      const float4 v2 = *reinterpret_cast<const float4*>(part + size_t(w + 16) * stride + 4 * o4);
      const float4 v3 = *reinterpret_cast<const float4*>(part + size_t(w + 24) * stride + 4 * o4);
      s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
      s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
      s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
      s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; w < nwg; w += 8) {
      const float4 v = *reinterpret_cast<const float4*>(part + size_t(w) * stride + 4 * o4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red[sp][e] = s;
    __syncthreads();
    if (sp == 0) {
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        const float4 v = red[k][e];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int l = o4 / (D * D / 4), o = o4 - l * (D * D / 4);
      float* dst = l == 0 ? dW0 : (l == 1 ? dW1 : dW2);
      reinterpret_cast<float4*>(dst)[o] = s;
    }
    return;
  }
  // bias gradients: 3 x 128 outputs, each the sum of nwg x 4 partials (per split: workgroup order; then split order)
  __shared__ float redb[8][32];
  const int e = tid & 31, sp = tid >> 5;
  const int o = (int(blockIdx.x) - NBLK) * 32 + e;     // 12 blocks x 32 outputs
  const int l = o / D, n = o - l * D;
  float sum = 0.f;
  for (int w = sp; w < nwg; w += 8) {
    const float* p = part + size_t(w) * (DW_FLOATS + DB_FLOATS) + DW_FLOATS + l * 4 * D + n;
    sum += (p[0] + p[D]) + (p[2 * D] + p[3 * D]);
  }
  redb[sp][e] = sum;
  __syncthreads();
  if (sp == 0) {
#pragma unroll
    for (int k = 1; k < 8; ++k) sum += redb[k][e];
    (l == 0 ? db0 : (l == 1 ? db1 : db2))[n] = sum;
  }
}

int device_cus_ef() { return device_cu_count(); }   // common.h: cached per device

}  // namespace

namespace bsms {

bool edge_fused_supported(int64_t D_, int H, int64_t p, int precision) {
  return precision != BSMS_F32 && D_ == 128 && H == 3 && p >= 1 && p <= 3;
}
size_t edge_fused_part_floats(int64_t rows) { return size_t(std::min<int64_t>(ceil_div(rows, ST_ROWS), kEdgeFusedMaxWg)) * (DW_FLOATS + DB_FLOATS); }

int launch_edge_fused_bwd(EdgeFusedBwdArgs a, int* nwg_out, hipStream_t s) {
  BSMS_REQUIRE(a.R < (int64_t(1) << 31) && a.p >= 1 && a.p <= 3, BSMS_E_UNSUPPORTED, "edge_fused_bwd: R = %lld, p = %d", (long long)a.R, a.p);
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_edge_fused_bwd), LDS_BYTES);
  BSMS_REQUIRE(attr == hipSuccess, BSMS_E_HIP, "edge_fused_bwd: cannot reserve %d bytes of LDS", LDS_BYTES);
  a.ntiles = int(ceil_div(a.R, ST_ROWS));
  const int nwg = int(std::min<int64_t>(a.ntiles, std::min(device_cus_ef(), kEdgeFusedMaxWg)));
  *nwg_out = nwg;
  if (nwg > 0) {
    hipLaunchKernelGGL(k_edge_fused_bwd, dim3(nwg), dim3(512), LDS_BYTES, s, a);
    BSMS_LAUNCH_CHECK();
  }
  return BSMS_OK;
}

// the fixed-order sum of the workgroups' partials into the three weight / bias gradients (stream-ordered after the kernel above)
int launch_edge_fused_reduce(const float* part, int nwg, float* const dW[3], float* const db[3], hipStream_t s) {
  hipLaunchKernelGGL(k_ef_reduce, dim3(DW_FLOATS / 4 / 32 + 3 * D / 32), dim3(256), 0, s, part, nwg, dW[0], dW[1], dW[2], db[0], db[1], db[2]);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

}  // namespace bsms
