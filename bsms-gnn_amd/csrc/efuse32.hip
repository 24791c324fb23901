// Fused backward of the edge MLP in fp32 (BSMS_F32), D = 128, hidden = 3: forward RECOMPUTE + LayerNorm backward + dgrad chain +
// the weight / bias gradients of the three D x D Linears in ONE kernel -- the fp32 port of efuse.hip ("The prize
// of the fp32 port").  Reference arithmetic: src/ops/basic.py:6-23 (MLP), :90-94 (edge message + scatter), under
// trainer/trainer.py:146-147.
//
// Why: the unfused fp32 backward writes gE[1..3] and the forward a_0..a_2 to HBM only so that a split-K kernel on a side lane
// can read them back (8.5 GB of the 23.5 GB a training step moves); measured by ablation that traffic costs the step 1.10 ms
// of 5.20 (profiles/r05_fusion_bound.txt), most of it through HBM contention with the node-level kernels.  Here the forward
// saves only the messages y, rstd, the fiber rows and the two node projections; a_0..a_2 are re-created on the matrix cores
// with the SAME stage as the forward (chain_dev.h: mfma_stage -- same operands, same product order: bit-identical), the
// gradient chain runs through the transposed packs, and dW_l += G_l^T A_{l-1} accumulates in registers.
//
// What differs from the bf16 kernel, and why:
//  * fp32 values are two fp16 pieces each (chain.h): W_1..W_3 are 192 KB and do not stay in LDS.  The weights STREAM through
//    the LDS ring of the chain kernels in the order [W_1, W_2, W_3^T, W_2^T, W_1^T] per 64-row tile (the FRAG / FRAG_T packs
//    the block's prepack writes anyway), 20 chunks of 17 KB, one workgroup barrier per chunk.
//  * waves 0-3 "chain" waves (16 rows each, lane <-> row, 256 registers: a_0..a_2 stay in fp32 registers for the masks and the
//    hand-over), waves 4-7 "gradient" waves: 192 dW accumulators per lane AND the ring's loaders (they issue no other vector
//    memory instruction, so their vmcnt counts LDS-DMA pieces only; chain_dev.h: loader_run explains why compute waves must
//    not load weights themselves).
//  * no barrier besides the ring's: a pair (G_l, A_{l-1}) is staged by the chain waves AFTER dgrad stage l (the stage's input
//    g_l is still in registers), between the stage's last chunk barrier and the next stage's first; the gradient waves read
//    it during the following three chunk periods and are done before the next pair is staged four barriers later.
//  * the reduction index of dW is the ROW: both operands need ONE power-of-two scale across all rows a workgroup ever
//    multiplies, and the magnitude of G_l is not known before the launch.  Running block exponent: every chain wave publishes
//    the largest |g|, |a| of its 16 rows before the stage's first barrier; after it everybody (chain and gradient waves alike)
//    takes the maximum of the four and RAISES the workgroup's exponent of that operand if needed -- the gradient waves then
//    multiply their accumulators by the exact power of two.  Elements 2^18 below the largest row seen so far lose their low
//    pieces: the window the per-tensor scale of k_wgrad has.  Partials are un-scaled by the final exponents and summed in fixed
//    order by k_ef_reduce (efuse.hip): run-to-run reproducible.
#include "chain.h"

#pragma clang fp contract(off)

using namespace bsms;

#include "chain_dev.h"

#include <type_traits>

namespace {

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int D = 128, NB = 8;
using R8 = Ring<NB>;
constexpr int NR = 4;                         // ring depth: three chunks in flight
constexpr int ROWB = 288;                     // staging row pitch in bytes: 128 halves + 32 bytes
constexpr int ST_ROWS = 64;                   // rows of a tile = 4 chain waves x 16
constexpr int ST_BYTES = ST_ROWS * ROWB;
constexpr int OFF_RING = R8::PRE_FLOATS * 4;  // [side table: fiber weights][exchange words] come first (chain_dev.h: Ring)
constexpr int OFF_GH = OFF_RING + NR * R8::CHF * 4, OFF_GL = OFF_GH + ST_BYTES, OFF_AH = OFF_GL + ST_BYTES, OFF_AL = OFF_AH + ST_BYTES;
constexpr int LDS_BYTES = OFF_AL + ST_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
constexpr int OFF_XCH = R8::SIDE_FLOATS * 4;  // 256 words after the side table: maxima of (pair, operand, chain wave)
constexpr int NSEQ = 5, CHUNKS = NSEQ * R8::NCH;   // packs per tile, ring chunks per tile (20)
constexpr int DW_FLOATS = 3 * D * D, DB_FLOATS = 3 * 4 * D;   // partials of one workgroup: the layout of efuse.hip (k_ef_reduce)

__device__ __forceinline__ f32x4 mma16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// byte offset of the 8-byte piece `piece` (4 consecutive columns) of row `row` in a swizzled [rows][144] 16-bit tile (efuse.hip)
__device__ __forceinline__ unsigned piece_off(int row, int piece) { return unsigned(row * ROWB + 8 * (piece ^ ((row >> 2) & 3))); }
// column fragment of a swizzled row-major 16-bit tile (efuse.hip: frag_col; census/tr_test.hip for the lane semantics)
__device__ __forceinline__ u32x4 frag_col(const char* T, unsigned tb, int kb, int cb) {
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  const char* p = T + tb + kb * (32 * ROWB) + cb * 32;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * ROWB));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(u32x4, v);
}
__device__ __forceinline__ float wave_rows_max(float m) {   // m: a row maximum (equal in the row's four lanes) -> max over the 16 rows
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m;
}
__device__ __forceinline__ int exp_of(float amax) {   // biased exponent E of chain_dev.h: scale_of (s = 2^(139 - E): amax s in [2^12, 2^13))
  const int E = int(__float_as_uint(amax) >> 23);
  return E < 12 ? 12 : E;
}
__device__ __forceinline__ float scale_from(int E) { return __uint_as_float(unsigned(266 - E) << 23); }
__device__ __forceinline__ float pow2i(int d) {   // 2^d for the exact rescaling of accumulators (d <= 0; below 2^-126: 0, the sums are negligible then)
  return d < -126 ? 0.f : __uint_as_float(unsigned(127 + d) << 23);
}

// 16 rows of one operand -> two staging tiles (h, l planes of s x), K block by K block (split_block's dword order = the
// B-operand order efuse.hip stages: pieces t = 2 kb2, 2 kb2 + 1)
__device__ __forceinline__ void stage_split(char* TH, char* TL, unsigned sb, const f32x4 (&v)[NB], float s) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    u32x4 h, l;
    split_block<NB>(v, c, s, h, l);
    *reinterpret_cast<u32x2*>(TH + sb + (2 * c) * 32) = u32x2{h[0], h[1]};
    *reinterpret_cast<u32x2*>(TH + sb + (2 * c + 1) * 32) = u32x2{h[2], h[3]};
    *reinterpret_cast<u32x2*>(TL + sb + (2 * c) * 32) = u32x2{l[0], l[1]};
    *reinterpret_cast<u32x2*>(TL + sb + (2 * c + 1) * 32) = u32x2{l[2], l[3]};
  }
}

// ------------------------------------------------------------------------------------------------ gradient + loader waves
// LI = 0..3: which pieces of a chunk this wave loads (chain_dev.h: loader_run with NL = 4) and which quarter of every dW it owns
template <int LI>
__device__ __forceinline__ void gradient_wave(const EdgeFused32Args& a, char* lds, int lane, int my_tiles) {
  constexpr int MINE = (R8::PER - LI + 3) / 4;
  const int gi = LI >> 1, gj = LI & 1;
  const int r = lane & 15, g = lane >> 4;
#ifdef EFV_PRIO
  __builtin_amdgcn_s_setprio(EFV_PRIO);   // experiments: the loaders / gradient waves ahead of the chain wave of their SIMD
#endif
  const unsigned tb = unsigned((4 * g + (r >> 2)) * ROWB + 8 * ((r & 3) ^ g));   // supplier base of the transposing reads
  const char* const GH = lds + OFF_GH; const char* const GL = lds + OFF_GL;
  const char* const AH = lds + OFF_AH; const char* const AL = lds + OFF_AL;
  const unsigned* const xch = reinterpret_cast<const unsigned*>(lds + OFF_XCH);
  f32x4 dw[3][4][4];
  float db[3][2];   // bias gradients: column sums of G_l over the staged rows 16 LI .. 16 LI + 15, columns 2 lane, 2 lane + 1
  int Eg[3], Ea[3];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    db[l][0] = db[l][1] = 0.f;
    Eg[l] = Ea[l] = 12;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) dw[l][x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- loader state (chain_dev.h: loader_run)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds) + unsigned(OFF_RING);
  const int total = my_tiles * CHUNKS;
  int is = 0, ic = 0, islot = 0;
  auto issue = [&]() {
    const float4* src = a.wseq[is] + size_t(ic) * R8::CH4 + lane;
    const unsigned dst = lds0 + unsigned(islot) * unsigned(R8::CHF * sizeof(float));
#pragma unroll
    for (int i = 0; i < MINE; ++i) glds16(src + (LI + i * 4) * 64, dst + (LI + i * 4) * 1024);
    if (++ic == R8::NCH) { ic = 0; if (++is == NSEQ) is = 0; }
    if (++islot == NR) islot = 0;
  };
  // exponents of pair l (0: Linear 1 .. 2: Linear 3) after this tile's maxima; the accumulators follow a raised exponent
  auto update_exps = [&](auto L) {
    constexpr int l = decltype(L)::value;   // compile-time: a run-time index would put the accumulator array into scratch memory
    unsigned mg = 0u, ma = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mg = max(mg, xch[(l * 2 + 0) * 4 + w]); ma = max(ma, xch[(l * 2 + 1) * 4 + w]); }
    const int eg = __builtin_amdgcn_readfirstlane(exp_of(__uint_as_float(mg))), ea = __builtin_amdgcn_readfirstlane(exp_of(__uint_as_float(ma)));   // uniform: scalar registers
    const int dg = max(eg - Eg[l], 0), da = max(ea - Ea[l], 0);
    if (dg + da > 0) {   // uniform
      const float f = pow2i(-(dg + da)), fb = pow2i(-dg);
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) dw[l][x][y] *= f;
      db[l][0] *= fb; db[l][1] *= fb;
      Eg[l] += dg; Ea[l] += da;
    }
  };
  // A pair's 96 products in 16 units of 6: unit u = ((ks * 2 + yp) * 4 + x) multiplies the G column block x with the two A column
  // blocks 2 yp, 2 yp + 1 over the rows 32 ks .. 32 ks + 31 of the staged tile.  The units of a pair are spread over the chunk
  // periods it may use (three for pairs 3 and 2): a SIMD's matrix pipe also serves a chain wave's 24 products per period, and with a
  // whole half pair (48) in one period the chain waves waited 3.7-5k cycles per stage at the chunk barriers (first timeline).
  // Two A column blocks at a time: with all four (32 fragment registers beside 192 accumulators) the wave spilled its accumulators.
  auto dw_units = [&](auto L, auto U0, auto U1) {
    constexpr int l = decltype(L)::value, u0 = decltype(U0)::value, u1 = decltype(U1)::value;
    u32x4 afh[2], afl[2];
#pragma unroll
    for (int u = u0; u < u1; ++u) {
      const int ks = u >> 3, yp = (u >> 2) & 1, x = u & 3;
      if (u == u0 || x == 0) {
#pragma unroll
        for (int y = 0; y < 2; ++y) { afh[y] = frag_col(AH, tb, ks, 4 * gj + 2 * yp + y); afl[y] = frag_col(AL, tb, ks, 4 * gj + 2 * yp + y); }
      }
      const u32x4 gh = frag_col(GH, tb, ks, 4 * gi + x), gl = frag_col(GL, tb, ks, 4 * gi + x);
#pragma unroll
      for (int y = 0; y < 2; ++y) dw[l][x][2 * yp + y] = mma16(gh, afl[y], dw[l][x][2 * yp + y]);
#pragma unroll
      for (int y = 0; y < 2; ++y) dw[l][x][2 * yp + y] = mma16(gh, afh[y], dw[l][x][2 * yp + y]);
#pragma unroll
      for (int y = 0; y < 2; ++y) dw[l][x][2 * yp + y] = mma16(gl, afh[y], dw[l][x][2 * yp + y]);
      __builtin_amdgcn_sched_barrier(0);   // keeps the fragments of later units from being requested ahead: 192 accumulators leave room for 24
    }
  };
  // bias gradient: column sums of the staged G rows 16 LI .. 16 LI + 15 (lane <-> columns 2 lane, 2 lane + 1), h + l pieces, four rows
  // at a time.  (Summing the column fragments dw_half holds anyway looked cheaper -- no LDS reads -- and came out 1-2 % WRONG: hipcc
  // 7.2 extracts the dwords of a ds_read_b64_tr_b16 pair incorrectly when they feed conversions, profiles/r05_efuse32_notes.txt.)
  auto bias_sums = [&](auto L) {
    constexpr int l = decltype(L)::value;
    using h2 = __attribute__((ext_vector_type(2))) _Float16;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) {
      // piece_off(row, lane >> 1) with row = 16 LI + 4 b4 + rr: the swizzle term is b4 for the four rows -- one lane-dependent base per
      // batch and immediate row offsets (sixteen per-row addresses kept across the tile loop cost sixteen registers)
      unsigned ln = unsigned(lane);
      asm volatile("" : "+v"(ln));   // opaque: keeps the four bases from being hoisted out of the tile loop (four registers the accumulators need)
      const unsigned base = 8u * ((ln >> 1) ^ unsigned(b4)) + 4u * (ln & 1u);
      unsigned hw[4], lw[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        hw[rr] = *reinterpret_cast<const unsigned*>(GH + base + (16 * LI + 4 * b4 + rr) * ROWB);
        lw[rr] = *reinterpret_cast<const unsigned*>(GL + base + (16 * LI + 4 * b4 + rr) * ROWB);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const h2 hv = __builtin_bit_cast(h2, hw[rr]), lv = __builtin_bit_cast(h2, lw[rr]);
        s0 += float(hv[0]) + float(lv[0]);
        s1 += float(hv[1]) + float(lv[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    db[l][0] += s0;
    db[l][1] += s1;
  };
  using L0 = std::integral_constant<int, 0>; using L1 = std::integral_constant<int, 1>; using L2 = std::integral_constant<int, 2>;
  const int ahead = NR - 1;
  for (int j = 0; j < ahead && j < total; ++j) issue();
  int j = 0;
  // one ring chunk: wait until chunk j has landed, publish it at the workgroup barrier, request the chunk that takes the freed slot
  auto chunk_step = [&]() {
    const int younger = min(ahead - 1, total - 1 - j);   // chunks issued after chunk j that may still be in flight
    if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MINE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MINE) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier q of the tile: chunk j is published; my LDS reads are done
    if (j + ahead < total) issue();
    ++j;
  };
  // The 20 chunk periods of a tile as STRAIGHT-LINE code: with the work selected by a switch on the chunk index inside one loop
  // every accumulator went through a phi at the merge and the allocator spilled 170 of them (first version: 688 bytes of scratch
  // per lane, the whole workgroup waiting for this wave at every barrier).  In the first tile the periods 0-2 multiply the
  // zero-initialised staging tiles (the kernel's prologue clears them): no condition on the tile index either.
  using U0 = std::integral_constant<int, 0>; using U4 = std::integral_constant<int, 4>; using U6 = std::integral_constant<int, 6>;
  using U8 = std::integral_constant<int, 8>; using U12 = std::integral_constant<int, 12>; using U16 = std::integral_constant<int, 16>;
  for (int it = 0; it < my_tiles; ++it) {
    chunk_step(); chunk_step(); chunk_step(); chunk_step();   // 0-3: loading only (the first forward stage runs at the ring's pace)
    chunk_step(); dw_units(L0{}, U0{}, U4{});            // 4: pair 1 of the PREVIOUS tile (staged after its last stage) in the periods that
    chunk_step(); dw_units(L0{}, U4{}, U8{});            // 5  have no pair of their own; the staging tile is rewritten behind barrier 11
    chunk_step(); dw_units(L0{}, U8{}, U12{});           // 6
    chunk_step(); dw_units(L0{}, U12{}, U16{});          // 7
    chunk_step(); update_exps(L2{}); bias_sums(L0{});    // 8: maxima of (g_3, a_2), published before this barrier
    chunk_step(); chunk_step(); chunk_step();            // 9-11
    chunk_step(); update_exps(L1{}); dw_units(L2{}, U0{}, U6{});    // 12: pair 3 was staged between barriers 11 and 12
    chunk_step(); dw_units(L2{}, U6{}, U12{});           // 13
    chunk_step(); dw_units(L2{}, U12{}, U16{}); bias_sums(L2{});    // 14
    chunk_step();                                        // 15
    chunk_step(); update_exps(L0{}); dw_units(L1{}, U0{}, U6{});    // 16: pair 2: between 15 and 16
    chunk_step(); dw_units(L1{}, U6{}, U12{});           // 17
    chunk_step(); dw_units(L1{}, U12{}, U16{}); bias_sums(L1{});    // 18
    chunk_step();                                        // 19
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // pair 1 of the last tile is staged
  dw_units(L0{}, U0{}, U16{});
  bias_sums(L0{});
  // ---- partial results of this workgroup, un-scaled by its final exponents: dW[l][n][k] (lane holds rows n = 64 gi + 16 x + 4 g + j,
  // column k = 64 gj + 16 y + r); layout of efuse.hip
  float* part = a.part + size_t(blockIdx.x) * (DW_FLOATS + DB_FLOATS);
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int e = Eg[l] + Ea[l] - 278, eb = Eg[l] - 139;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          part[l * D * D + (64 * gi + 16 * x + 4 * g + jj) * D + 64 * gj + 16 * y + r] = ldexpf(dw[l][x][y][jj], e);
    float* pdb = part + DW_FLOATS + (l * 4 + LI) * D;   // k_ef_reduce adds the four waves' rows
    *reinterpret_cast<float2*>(pdb + 2 * lane) = make_float2(ldexpf(db[l][0], eb), ldexpf(db[l][1], eb));
  }
}

// experiments (profiles/ef32_timeline.py): phase stamps of chain wave 0, 16 slots per tile
#ifdef BSMS_EXPERIMENTS
#define EF32_STAMP(slot)                                                                                        \
  do {                                                                                                           \
    if (a.timing && lane == 0 && wave == 0) {                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
      a.timing[size_t(int(blockIdx.x) + it * int(gridDim.x)) * 16 + (slot)] = __builtin_amdgcn_s_memtime();     \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
    }                                                                                                            \
  } while (0)
#else
#define EF32_STAMP(slot) do {} while (0)
#endif

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_edge_fused32_bwd(EdgeFused32Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int my_tiles = (a.ntiles - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
  {   // side table: fiber weights^T, rows past p + 1 zero; the exchange words start at zero
    float* side = reinterpret_cast<float*>(lds);
    for (int o = tid; o < 4 * D; o += 512) side[o] = o < (a.p + 1) * D ? a.wft[o] : 0.f;
    if (tid < 64) reinterpret_cast<unsigned*>(lds + OFF_XCH)[tid] = 0u;
    for (int o = tid; o < 4 * ST_BYTES / 16; o += 512) reinterpret_cast<float4*>(lds + OFF_GH)[o] = make_float4(0.f, 0.f, 0.f, 0.f);   // the staging tiles (see gradient_wave)
  }
  __syncthreads();
  if (wave >= 4) {   // uniform
    switch (wave - 4) {
      case 0: gradient_wave<0>(a, lds, lane, my_tiles); break;
      case 1: gradient_wave<1>(a, lds, lane, my_tiles); break;
      case 2: gradient_wave<2>(a, lds, lane, my_tiles); break;
      default: gradient_wave<3>(a, lds, lane, my_tiles); break;
    }
    return;
  }
  // ======================================================================================= chain waves
  const int r = lane & 15, g = lane >> 4;
  const unsigned sb = unsigned((16 * wave + r) * ROWB + 8 * (g ^ ((r >> 2) & 3)));   // base of this lane's staging pieces (efuse.hip)
  const float* const wft = reinterpret_cast<const float*>(lds);
  unsigned* const xch = reinterpret_cast<unsigned*>(lds + OFF_XCH);
  float4* const ring = reinterpret_cast<float4*>(lds + OFF_RING);
  char* const GH = lds + OFF_GH; char* const GL = lds + OFF_GL; char* const AH = lds + OFF_AH; char* const AL = lds + OFF_AL;
  Slot slot{0, NR};
  const float rcpE = 1.f / float(a.E);
  int Eg[3] = {12, 12, 12}, Ea[3] = {12, 12, 12};   // the workgroup's running exponents, kept in step with the gradient waves'
  auto publish = [&](int l, float gmax_rows, float amax_rows) {   // before the first barrier of dgrad stage l + 1
    const float mg = wave_rows_max(gmax_rows), ma = wave_rows_max(amax_rows);
    if (lane == 0) { xch[(l * 2 + 0) * 4 + wave] = __float_as_uint(mg); xch[(l * 2 + 1) * 4 + wave] = __float_as_uint(ma); }
  };
  auto hand_over = [&](int l, const f32x4 (&gv)[NB], const f32x4 (&av)[NB]) {   // after dgrad stage l + 1: (G_{l+1}, A_l) -> staging
    unsigned mg = 0u, ma = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mg = max(mg, xch[(l * 2 + 0) * 4 + w]); ma = max(ma, xch[(l * 2 + 1) * 4 + w]); }
    Eg[l] = max(Eg[l], exp_of(__uint_as_float(mg)));
    Ea[l] = max(Ea[l], exp_of(__uint_as_float(ma)));
    stage_split(GH, GL, sb, gv, scale_from(Eg[l]));
    stage_split(AH, AL, sb, av, scale_from(Ea[l]));
  };
  // what a tile needs from the plan: the edge row itself and the node rows of its two endpoints
  struct Where { unsigned row; int64_t row64; bool live; int64_t isrc, idst; };
  auto locate = [&](int tile) {
    Where wq;
    wq.row64 = int64_t(tile) * ST_ROWS + wave * 16 + r;
    wq.live = wq.row64 < a.R;
    wq.row = wq.live ? unsigned(wq.row64) : 0u;   // lanes past the end read row 0; their gradient is zeroed below
    EdgeRef e = edge_ref(wq.row, unsigned(a.E), rcpE);
    if (unsigned(e.q) >= unsigned(a.E)) { e.b = int(wq.row / unsigned(a.E)); e.q = int(wq.row - unsigned(e.b) * unsigned(a.E)); }   // estimate off by more than one (tiny E, huge B): exact
    wq.isrc = int64_t(e.b) * a.N + a.src[e.q];
    wq.idst = int64_t(e.b) * a.N + a.dst[e.q];
    return wq;
  };
  // The endpoint rows of a tile are requested a tile AHEAD (before the last gradient stage, when a_1 / a_2 have left their
  // registers), dy / y before the second forward Linear: in the first version both gathers sat in front of their first use,
  // 8.8k + 5.4k of 57.8k cycles per tile (profiles/r05_ef32_timeline.txt).
  f32x4 ps[NB], pd[NB];
  float4 fib;
  Where cur = locate(int(blockIdx.x));
  load_rows<NB>(ps, a.Ps + cur.isrc * D, g);
  load_rows<NB>(pd, a.Pd + cur.idst * D, g);
  fib = *reinterpret_cast<const float4*>(a.fiber + size_t(cur.row) * 4);
  // g_0 of the PREVIOUS tile: its eight stores go out in three batches between the first phases of the next tile -- issued in one
  // burst behind the last stage the four chain waves fill the CU's store path (~10 B/clk) and sit on it for 2.5k cycles (timeline)
  f32x4 g0v[NB];
  int64_t g0off = -1;
  auto store_g0 = [&](int t0, int t1) {
    if (g0off < 0) return;   // per lane: rows past the end (and the first tile) store nothing
    float* rowp = a.g0 + g0off;
#pragma unroll
    for (int t = 0; t < NB; ++t)
      if (t >= t0 && t < t1) *reinterpret_cast<f32x4*>(rowp + 16 * t + 4 * g) = g0v[t];
  };
  for (int it = 0; it < my_tiles; ++it) {
    const int tile = int(blockIdx.x) + it * int(gridDim.x);
    EF32_STAMP(0);
    store_g0(0, 3);
    const int64_t row64 = cur.row64;
    const bool live = cur.live;
    const unsigned row = cur.row;
    const int64_t idst = cur.idst;
    const Where nxt = locate(it + 1 < my_tiles ? tile + int(gridDim.x) : tile);
    // ---- a_0 = relu(Ps[src] + Pd[dst] + Wf . fiber)   (chain.hip: k_edge_fwd / k_chain_fwd IN_EDGE, same operations in the same order)
    f32x4 a0[NB], a1[NB], a2[NB], acc[NB];
    {
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] = ps[t] + pd[t];
      const float fv[4] = {fib.x, fib.y, fib.z, fib.w};
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (c < a.p) axpy_features<NB>(acc, wft + c * D, fv[c], g);
      const float nrm = a.p == 1 ? fib.y : (a.p == 2 ? fib.z : fib.w);
      axpy_features<NB>(acc, wft + a.p * D, nrm, g);
      relu_into<NB>(a0, acc);
    }
    const float m0 = row_amax<NB>(a0);
    store_g0(3, 6);
    EF32_STAMP(1);
    mfma_stage<NB, true, 2>(acc, a0, scale_of(m0), ring, slot, lane);
    relu_into<NB>(a1, acc);
    store_g0(6, 8);
    EF32_STAMP(2);
    const float m1a = row_amax<NB>(a1);
    f32x4 gr[NB], yr[NB];   // requested now, used after the second Linear
    load_rows<NB>(gr, a.dy + idst * D, g);
    load_rows<NB>(yr, a.y + size_t(row) * D, g);
    const float rstd_row = a.rstd[row];
    mfma_stage<NB, true, 2>(acc, a1, scale_of(m1a), ring, slot, lane);
    relu_into<NB>(a2, acc);
    const float m2a = row_amax<NB>(a2);
    EF32_STAMP(3);
    // ---- LayerNorm backward (no affine): g_3 = rstd (dy - mean(dy) - y mean(dy y))   (chain.hip: k_edge_bwd, same order)
    {
      const float rs = live ? rstd_row : 0.f;   // rows past the end contribute nothing to dW / db
      const float mu1 = row_sum<NB>(gr) * (1.f / D);
      float s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) s2 = fmaf(gr[t][k], yr[t][k], s2);
      s2 = group_sum(s2);
      const float mu2 = s2 * (1.f / D);
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) gr[t][k] = rs * (gr[t][k] - mu1 - yr[t][k] * mu2);
    }
    // ---- three gradient stages: acc = W_l^T g_l (transposed pack), masked by a_{l-1} > 0; the pair (G_l, A_{l-1}) goes to the
    // gradient waves after the stage, the next pair's maxima before the next stage's first barrier
    f32x4 gn[NB];
    float mg = row_amax<NB>(gr);
    EF32_STAMP(4);
    publish(2, mg, m2a);
    EF32_STAMP(5);
    mfma_stage<NB, true, 1>(acc, gr, scale_of(mg), ring, slot, lane);               // Linear 3
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) gn[t][k] = a2[t][k] > 0.f ? acc[t][k] : 0.f;
    EF32_STAMP(6);
    hand_over(2, gr, a2);
    mg = row_amax<NB>(gn);
    publish(1, mg, m1a);
    EF32_STAMP(7);
    mfma_stage<NB, true, 1>(acc, gn, scale_of(mg), ring, slot, lane);               // Linear 2
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) gr[t][k] = a1[t][k] > 0.f ? acc[t][k] : 0.f;
    EF32_STAMP(8);
    hand_over(1, gn, a1);
    mg = row_amax<NB>(gr);
    load_rows<NB>(ps, a.Ps + nxt.isrc * D, g);   // the next tile's endpoint rows land under the last gradient stage
    load_rows<NB>(pd, a.Pd + nxt.idst * D, g);
    fib = *reinterpret_cast<const float4*>(a.fiber + size_t(nxt.row) * 4);
    publish(0, mg, m0);
    EF32_STAMP(9);
    mfma_stage<NB, true, 1>(acc, gr, scale_of(mg), ring, slot, lane);               // Linear 1
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) gn[t][k] = a0[t][k] > 0.f ? acc[t][k] : 0.f;
    EF32_STAMP(10);
    hand_over(0, gr, a0);
    EF32_STAMP(11);
    // ---- g_0: input of the scatter / fiber-gradient kernel (plain stores: read next, they stay in L2 / the memory-side cache)
#pragma unroll
    for (int t = 0; t < NB; ++t) g0v[t] = gn[t];
    g0off = live ? int64_t(row64) * D : -1;
    EF32_STAMP(12);
    cur = nxt;
  }
  store_g0(0, 8);   // the last tile's
  lds_barrier();   // the last pair is staged: the gradient waves finish behind this barrier
}

int device_cus_ef32() { return device_cu_count(); }   // common.h: cached per device

}  // namespace

namespace bsms {

bool edge_fused32_supported(int64_t D_, int H, int64_t p, int precision) {
  return precision == BSMS_F32 && D_ == 128 && H == 3 && p >= 1 && p <= 3;
}

int launch_edge_fused32_bwd(EdgeFused32Args a, int* nwg_out, hipStream_t s) {
  BSMS_REQUIRE(a.R < (int64_t(1) << 31) && a.p >= 1 && a.p <= 3, BSMS_E_UNSUPPORTED, "edge_fused32_bwd: R = %lld, p = %d", (long long)a.R, a.p);
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_edge_fused32_bwd), LDS_BYTES);
  BSMS_REQUIRE(attr == hipSuccess, BSMS_E_HIP, "edge_fused32_bwd: cannot reserve %d bytes of LDS", LDS_BYTES);
  a.ntiles = int(ceil_div(a.R, ST_ROWS));
  const int nwg = int(std::min<int64_t>(a.ntiles, std::min(device_cus_ef32(), kEdgeFusedMaxWg)));
  *nwg_out = nwg;
  if (nwg > 0) {
    hipLaunchKernelGGL(k_edge_fused32_bwd, dim3(nwg), dim3(512), LDS_BYTES, s, a);
    BSMS_LAUNCH_CHECK();
  }
  return BSMS_OK;
}

}  // namespace bsms
