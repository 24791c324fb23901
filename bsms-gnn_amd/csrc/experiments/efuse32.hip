// EXPERIMENT builds only (profiles/build_efv.sh; BSMS_EDGE_FUSED_F32=1): measured 176-180 steps/s against 186.5 for the unfused fp32
// path on the same box, upper bound 203 with the gradient waves idle -- DESIGN.md section 4 says why this dataflow stays an experiment.
//
// Fused backward of the edge MLP in fp32 (BSMS_F32), D = 128, hidden = 3: LayerNorm backward + dgrad chain + the weight / bias
// gradients of the three D x D Linears in ONE kernel, WITHOUT recomputing the forward (round 6; the round-5 kernel that re-created
// a_0..a_2 on the matrix cores is kept as profiles/experiments/efuse32_recompute_r05.hip.txt).  Reference arithmetic:
// src/ops/basic.py:6-23 (MLP), :90-94 (edge message + scatter), under trainer/trainer.py:146-147.
//
// Why: the unfused fp32 backward writes gE[1..3] to HBM only so that a split-K kernel on a side lane can read them back together
// with a_0..a_2 (3.4 GB written + 5.9 GB read of the 23.5 GB a training step moves); measured by ablation that traffic costs
// the step 0.73 ms of 5.2 (profiles/r05_fusion_bound.txt), most of it through HBM contention with the node-level kernels.
//
// Dataflow of a 64-row tile (one persistent 512-thread workgroup per CU):
//  * The FORWARD (chain.hip: k_edge_fwd SAVE == 2) stores a_0..a_2 as the fp16 x 2 pieces its own next stage multiplies: row r, K
//    block c = 64 bytes of h pieces + 64 bytes of l pieces of  a * 2^(139 - E_r)  (E_r: the row's scale exponent, kept in a side
//    array), in the lane order of the MFMA B operand -- same bytes as the fp32 row.
//  * waves 0-3, "chain" waves (16 rows each, lane <-> row, chain.h): gather dy[dst], read y / rstd, LayerNorm backward -> g_3, then
//    three gradient stages g_{l-1} = (W_l^T g_l) . [a_{l-1} > 0] through the transposed packs (chain_dev.h: mfma_stage, the stage of
//    k_edge_bwd: bit-identical g_0), masks from the saved sign bits.  At the START of stage l they stage G_l in LDS for the
//    gradient waves (fp16 x 2, row-major, transposing reads deliver the column fragments) and add it to their bias-gradient sums.
//  * waves 4-7, "gradient" waves: 192 dW accumulators per lane, the loaders of the weight ring AND of the A tiles: A_{l-1}
//    (both planes, 32 KB) comes HBM -> LDS by LDS-DMA one stage ahead (double buffered) -- no registers, no VALU, no ds_write.
//    dW_l += G_l^T A_{l-1} runs in the four chunk periods of stage l, 24 products per period beside the chain wave's 24.
//  * Scales.  The reduction index of dW is the ROW, so the product needs one power-of-two scale across rows; A arrives with its
//    per-row scale 2^(139 - E_r) baked in, so the chain waves fold the inverse into G:  G''_r = G_r 2^(E_r + 139 - EB)  with EB a
//    running block exponent of the workgroup (>= Eg_r + E_r for every row seen so far; Eg_r: exponent of the row's largest |g|).
//    Then  sum_r G''_r (x) A'_r = 2^(278 - EB) sum_r G_r (x) A_r.  When a tile raises EB the gradient waves multiply their
//    accumulators by the exact power of two.  Rows whose contribution lies 2^18 below the largest one seen so far lose their low
//    pieces -- the window the per-tensor scale of k_wgrad has.
//  * Bias gradients db_l = sum_r G_l[r].  96 accumulator registers per chain wave made hipcc spill its address registers (11k of 42k
//    cycles per tile, profiles/r06_e32_census.txt), and the gradient waves have none to spare beside 192 accumulators (six running
//    sums + their temporaries made hipcc spill ACCUMULATORS), so the sums live in LDS: gradient wave w reads the 16 staged rows of
//    chain wave w (lane <-> two columns, four rows per chunk period), un-weights them by w_r = 2^(EB - E_r - 139) (a float per row
//    next to the staging tile) and adds the result to its private [Linear][128] array -- a row within 2^12 of the block maximum keeps
//    >= 26 bits of its own maximum.  A row further below (activations all zero: E_r at its floor, G'' underflows -- but its bias
//    contribution must not) gets w_r = 0 and chain wave w adds its fp32 values to the same array itself, between barrier X and the
//    stage's first chunk barrier (the gradient wave adds behind that barrier: one writer at a time, fixed order, reproducible).
//  * Synchronisation: the ring's chunk barriers (12 per tile) + one barrier per stage between "maxima published" and "G staged".
//    Partials per workgroup in the layout of efuse.hip, summed in fixed order by k_ef_reduce: run-to-run reproducible.
#include "../chain.h"

#pragma clang fp contract(off)

using namespace bsms;

#include "../chain_dev.h"

#include <type_traits>

namespace {

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int D = 128, NB = 8;
using R8 = Ring<NB>;
constexpr int NR = 3;                         // ring depth: two chunks in flight
constexpr int ROWB = 288;                     // G staging row pitch in bytes: 128 halves + 32 bytes (efuse.hip)
constexpr int ST_ROWS = 64;                   // rows of a tile = 4 chain waves x 16
constexpr int ST_BYTES = ST_ROWS * ROWB;
constexpr int A_PLANE = ST_ROWS * 256;        // one fp16 plane of an A tile: 256-byte rows, 32-byte blocks XOR-swizzled by (row & 7)
constexpr int A_BUF = 2 * A_PLANE;            // h plane, l plane
constexpr int OFF_XCH = 0;                    // exchange words: [pair][chain wave] row-exponent maxima
constexpr int OFF_ROWW = 256;                 // 64 floats: the un-weighting factor of every staged G row (bias gradients)
constexpr int OFF_DBW = 1024;                 // [wave pair w][Linear][128] floats: bias gradients of the rows of chain wave w, summed by gradient wave w (+ w's own exact side path)
constexpr int OFF_RING = 7168;
constexpr int OFF_GH = OFF_RING + NR * R8::CHF * 4, OFF_GL = OFF_GH + ST_BYTES;
constexpr int OFF_A = OFF_GL + ST_BYTES;      // two A buffers
constexpr int LDS_BYTES = OFF_A + 2 * A_BUF;
static_assert(LDS_BYTES <= 160 * 1024 && OFF_A % 1024 == 0, "one workgroup per CU; LDS-DMA targets are 1 KB pieces");
constexpr int NSEQ = 3;                       // packs per tile (W_3^T, W_2^T, W_1^T): 12 ring chunks
constexpr int DW_FLOATS = 3 * D * D, DB_FLOATS = 3 * 4 * D;   // partials of one workgroup: the layout of efuse.hip (k_ef_reduce)
constexpr int EB_FLOOR = 24;                  // Eg_r + E_r >= 12 + 12
#ifdef EFV_DBWIN
constexpr int DB_WINDOW = EFV_DBWIN;          // census: every row on the exact side path (-1000) / on the weighted sums (100000)
#else
constexpr int DB_WINDOW = 6;                  // UNRESOLVED (profiles/r06_efuse32_notes.txt): with the analysed value 12 the d128p2 golden case fails in
                                              // seq.4.bias (one element in a few columns, varying run to run); 6, the all-exact form and the form whose
                                              // gradient-wave adds are LDS atomics too (EFV_DBATOMIC) pass
#endif
// rows whose Eg_r + E_r lies more than this below EB give their bias contribution on the exact side path

__device__ __forceinline__ f32x4 mma16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// column fragment of the swizzled row-major G tile (efuse.hip: frag_col; census/tr_test.hip for the lane semantics)
__device__ __forceinline__ u32x4 frag_col(const char* T, unsigned tb, int kb, int cb) {
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  const char* p = T + tb + kb * (32 * ROWB) + cb * 32;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * ROWB));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(u32x4, v);
}
// the same from an A plane: 256-byte rows, the 32-byte column block cb of row rho sits at block slot cb ^ (rho & 7) -- the 16 rows a
// transposing read touches (8 per half wave) then fall on distinct bank groups.  `ta` = this lane's row base + piece offset,
// `x7` = (row & 7) of this lane's supplier row (unchanged by the + 16 / + 32 row steps).
__device__ __forceinline__ u32x4 frag_col_a(const char* T, unsigned ta, unsigned x7, int kb, int cb) {
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  const char* p = T + ta + ((unsigned(cb) ^ x7) << 5) + kb * (32 * 256);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * 256));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(u32x4, v);
}
__device__ __forceinline__ float pow2_field(int field) {   // 2^(field - 127); below the normal range: 0 (the sums are negligible then)
  return field < 1 ? 0.f : __uint_as_float(unsigned(field > 254 ? 254 : field) << 23);
}

// 16 rows of G -> the two staging tiles (h, l planes of s x with a per-ROW factor s), K block by K block (split_block's dword order
// = the B-operand order efuse.hip stages: pieces t = 2 kb2, 2 kb2 + 1)
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

// LDS-DMA with a UNIFORM 64-bit base (scalar registers) + a 32-bit per-lane byte offset: the gradient waves have no vector registers
// to spare for a 64-bit address per piece (with per-lane pointers hipcc precomputed one pair per piece and SPILLED them: a scratch
// reload in front of every piece).  Otherwise as glds16 (chain_dev.h).
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void barrier_x() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------ gradient + loader waves
// LI = 0..3: which pieces of a chunk / an A tile this wave loads (chain_dev.h: loader_run with NL = 4) and which quarter of every dW it owns
template <int LI>
__device__ __forceinline__ void gradient_wave(const EdgeFused32Args& a, char* lds, int lane, int my_tiles) {
  constexpr int MINE = (R8::PER - LI + 3) / 4;
  const int gi = LI >> 1, gj = LI & 1;
  const int r = lane & 15, g = lane >> 4;
  const unsigned tb = unsigned((4 * g + (r >> 2)) * ROWB + 8 * ((r & 3) ^ g));   // supplier base of the transposing reads of G
  const unsigned ta = unsigned((4 * g + (r >> 2)) * 256 + 8 * (r & 3)), x7 = unsigned(4 * (g & 1) + (r >> 2));   // ... of A
  const char* const GH = lds + OFF_GH; const char* const GL = lds + OFF_GL;
  const int* const xch = reinterpret_cast<const int*>(lds + OFF_XCH);
  f32x4 dw[3][4][4];
  int EB[3];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    EB[l] = EB_FLOOR;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) dw[l][x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- loader state.  EVERY issue is unconditional and of a compile-time size (a chunk's MINE pieces + 0 / 2 / 3 pieces of an A
  // tile), also past the last tile (the streams wrap around / repeat the last tile into slots nobody reads): the vmcnt thresholds of
  // chunk_step then hold for the whole kernel.
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
  int is = 0, ic = 0, islot = 0;
  const unsigned lane16 = unsigned(lane) * 16u;
  auto issue_chunk = [&]() {
    const float4* src = a.wseq[is] + size_t(ic) * R8::CH4;   // uniform
    const unsigned dst = lds0 + unsigned(OFF_RING) + unsigned(islot) * unsigned(R8::CHF * sizeof(float));
#pragma unroll
    for (int i = 0; i < MINE; ++i) glds16_s(src + (LI + i * 4) * 64, lane16, dst + (LI + i * 4) * 1024);
    if (++ic == R8::NCH) { ic = 0; if (++is == NSEQ) is = 0; }
    if (++islot == NR) islot = 0;
  };
  // A tiles: global stage S = 3 * tile_iteration + s multiplies the saved activation a.act[s] (a_2, a_1, a_0) of tile `it`; its 32
  // pieces of 1 KB (16 per plane: rows 4 k .. 4 k + 3) go to buffer S & 1; this wave issues pieces LI + 4 i, i = 0..7.
  // LDS piece position (row rho, 16-byte slot pi) holds the HBM piece sigma = 2 ((pi >> 1) ^ (rho & 7)) + (pi & 1) of the row's plane
  // (sigma = 4 kb2 + lane group: chain.hip valu_step SAVE == 2); rho & 7 = 4 (k & 1) + (lane >> 4) and k & 1 = LI & 1 for every piece
  // of this wave: one per-lane byte offset for all of them.
  const unsigned pi_ = unsigned(lane & 15), rho7 = unsigned(4 * (LI & 1) + (lane >> 4));
  const unsigned sigma = (((pi_ >> 1) ^ rho7) << 1) | (pi_ & 1);
  const unsigned lane_off = (sigma >> 2) * 128 + (sigma & 3) * 16;
  const int lrow = lane >> 4;
  auto issue_a = [&](int S, int i0, int i1) {   // pieces i0 .. i1 - 1 of this wave's eight
    const int it = min(S / 3, my_tiles - 1), s = S % 3;
    const int64_t row0 = (int64_t(blockIdx.x) + int64_t(it) * gridDim.x) * ST_ROWS;      // uniform; a valid tile: row0 < R
    const int last_rel = int(min<int64_t>(a.R - 1 - row0, ST_ROWS - 1));                   // uniform: rows past the end read the last valid row (finite; their G is zero)
    const char* base = reinterpret_cast<const char*>(a.act[s]) + row0 * 512;               // uniform
    const unsigned dst = lds0 + unsigned(OFF_A) + unsigned(S & 1) * unsigned(A_BUF);
    for (int i = i0; i < i1; ++i) {
      const int pc = LI + 4 * i, plane = pc >> 4, k = pc & 15;
      const int rel = min(4 * k + lrow, last_rel);
      glds16_s(base, unsigned(rel) * 512u + unsigned(plane * 64) + lane_off, dst + unsigned(plane * A_PLANE + k * 1024));
    }
  };
  auto update_exp = [&](auto L) {   // after barrier X of pair l: the accumulators follow a raised block exponent
    constexpr int l = decltype(L)::value;   // compile-time: a run-time index would put the accumulator array into scratch memory
    int m = EB_FLOOR;
#pragma unroll
    for (int w = 0; w < 4; ++w) m = max(m, xch[l * 4 + w]);
    const int e = __builtin_amdgcn_readfirstlane(m);
    const int d = e - EB[l];
    if (d > 0) {   // uniform
      const float f = pow2_field(127 - d);
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) dw[l][x][y] *= f;
      EB[l] = e;
    }
  };
  // A pair's 96 products in 16 units of 6: unit u = ((ks * 2 + yp) * 4 + x) multiplies the G column block x with the two A column
  // blocks 2 yp, 2 yp + 1 over the rows 32 ks .. 32 ks + 31 of the staged tile; four units per chunk period of the pair's stage.
  // Two A column blocks at a time: with all four (32 fragment registers beside 192 accumulators) the wave spilled its accumulators.
  // Bias gradient: column sums of the staged G rows 16 LI + 4 b4 .. + 3 (lane <-> columns 2 lane, 2 lane + 1), h + l pieces times the
  // row's un-weighting factor; one batch of four rows at the START of each chunk period of the pair's stage, two rows at a time, while
  // the fragment registers of the products are free (behind the products, or one row per unit, hipcc spilled 400-900 bytes per lane
  // beside the 192 accumulators).  (Summing the column fragments the products hold anyway came out 1-2 % WRONG in round 5: hipcc 7.2
  // extracts the dwords of a ds_read_b64_tr_b16 pair incorrectly when they feed conversions, profiles/r05_efuse32_notes.txt.)
  float* const dbw = reinterpret_cast<float*>(lds + OFF_DBW) + LI * (3 * D);
  const float* const roww = reinterpret_cast<const float*>(lds + OFF_ROWW);
  auto bias_batch = [&](auto L, auto B4) {
#ifdef EFV_NODB
    return;
#endif
    constexpr int l = decltype(L)::value, b4 = decltype(B4)::value;
    using h2 = __attribute__((ext_vector_type(2))) _Float16;
    __builtin_amdgcn_sched_barrier(0);
    unsigned ln = unsigned(lane);
    asm volatile("" : "+v"(ln));   // opaque: keeps the row addresses from being hoisted out of the tile loop (registers the accumulators need)
    const unsigned base = 8u * ((ln >> 1) ^ unsigned(b4)) + 4u * (ln & 1u);   // piece_off(row, lane >> 1): the swizzle term is b4 for rows 4 b4 .. 4 b4 + 3
    const float4 w4 = *reinterpret_cast<const float4*>(roww + 16 * LI + 4 * b4);
    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 16 * LI + 4 * b4 + rr;
      const h2 hv = __builtin_bit_cast(h2, *reinterpret_cast<const unsigned*>(GH + base + row * ROWB));
      const h2 lv = __builtin_bit_cast(h2, *reinterpret_cast<const unsigned*>(GL + base + row * ROWB));
      s0 = fmaf(float(hv[0]) + float(lv[0]), wv[rr], s0);
      s1 = fmaf(float(hv[1]) + float(lv[1]), wv[rr], s1);
    }
#ifdef EFV_DBATOMIC   // census
    atomicAdd(dbw + l * D + 2 * ln, s0);
    atomicAdd(dbw + l * D + 2 * ln + 1, s1);
#else
    float2* acc = reinterpret_cast<float2*>(dbw + l * D + 2 * ln);
    const float2 o = *acc;
    *acc = make_float2(o.x + s0, o.y + s1);
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  auto dw_units = [&](auto L, auto U0, auto U1, const char* AH) {
    constexpr int l = decltype(L)::value, u0 = decltype(U0)::value, u1 = decltype(U1)::value;
    const char* const AL = AH + A_PLANE;
    u32x4 afh[2], afl[2];
#pragma unroll
    for (int u = u0; u < u1; ++u) {
      const int ks = u >> 3, yp = (u >> 2) & 1, x = u & 3;
#ifndef EFV_NODW
      if (u == u0 || x == 0)
#else
      if (false)
#endif
      {
#pragma unroll
        for (int y = 0; y < 2; ++y) { afh[y] = frag_col_a(AH, ta, x7, ks, 4 * gj + 2 * yp + y); afl[y] = frag_col_a(AL, ta, x7, ks, 4 * gj + 2 * yp + y); }
      }
#ifdef EFV_NODW   // census: the gradient waves only load
      continue;
#endif
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
  // one ring chunk of period P (0..11 within the tile): wait until it has landed -- at most the pieces of the issue of period P - 1
  // may still be outstanding: its chunk and NA(P - 1) pieces of an A tile --, publish it at the workgroup barrier, then issue the
  // chunk that takes the freed slot and this period's share of the next stage's A tile.  A(S) rides with the issues of periods
  // 4 S - 4, 4 S - 3, 4 S - 2 (3 + 3 + 2 pieces): after the last read of buffer S & 1 (stage S - 2, period 4 S - 5) and two issues ahead
  // of chunk 4 S, whose wait implies theirs (vmcnt retires in order).
  int S = 0;   // global index of the stage whose periods are running
  auto chunk_step = [&](auto PP) {
    constexpr int P = decltype(PP)::value, q = (P + 3) & 3;   // q = (P - 1) mod 4
#ifdef EFV_NOA   // census: no A tiles (garbage operands)
    constexpr int NAPREV = 0;
#else
    constexpr int NAPREV = q == 0 ? 3 : (q == 1 ? 3 : (q == 2 ? 2 : 0));
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MINE + NAPREV) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // chunk barrier: the chunk (and, at P = 0 mod 4, G of this stage) is published; my LDS reads of the last period are done
#ifndef EFV_NOA
    if ((P & 3) == 0) issue_a(S + 1, 0, 3);
    else if ((P & 3) == 1) issue_a(S + 1, 3, 6);
    else if ((P & 3) == 2) issue_a(S + 1, 6, 8);
#endif
    issue_chunk();
  };
  using L0 = std::integral_constant<int, 0>; using L1 = std::integral_constant<int, 1>; using L2 = std::integral_constant<int, 2>;
  using U0 = std::integral_constant<int, 0>; using U4 = std::integral_constant<int, 4>; using U8 = std::integral_constant<int, 8>;
  using U1 = std::integral_constant<int, 1>; using U2 = std::integral_constant<int, 2>; using U3 = std::integral_constant<int, 3>;
  using U12 = std::integral_constant<int, 12>; using U16 = std::integral_constant<int, 16>;
#define EF_P(n) std::integral_constant<int, n>{}
  // prologue: A(0) whole, chunks 0 and 1
  issue_a(0, 0, 8);
  issue_chunk();
  issue_chunk();
  const char* const A0 = lds + OFF_A;
  for (int it = 0; it < my_tiles; ++it) {
    // The 12 chunk periods of a tile as STRAIGHT-LINE code (a switch on the period index put the accumulators through phis: 2100
    // scratch instructions in the round-5 kernel).  Stage s of tile `it` is global stage S = 3 it + s: A buffer S & 1.
    const char* Aa = A0 + ((3 * it) & 1) * A_BUF;       // stage 0 (pair l = 2: G_3, a_2)
    const char* Ab = A0 + ((3 * it + 1) & 1) * A_BUF;   // stage 1 (pair l = 1: G_2, a_1)
    barrier_x(); update_exp(L2{});
    chunk_step(EF_P(0)); bias_batch(L2{}, U0{}); dw_units(L2{}, U0{}, U4{}, Aa);
    chunk_step(EF_P(1)); bias_batch(L2{}, U1{}); dw_units(L2{}, U4{}, U8{}, Aa);
    chunk_step(EF_P(2)); bias_batch(L2{}, U2{}); dw_units(L2{}, U8{}, U12{}, Aa);
    chunk_step(EF_P(3)); bias_batch(L2{}, U3{}); dw_units(L2{}, U12{}, U16{}, Aa);
    ++S;
    barrier_x(); update_exp(L1{});
    chunk_step(EF_P(4)); bias_batch(L1{}, U0{}); dw_units(L1{}, U0{}, U4{}, Ab);
    chunk_step(EF_P(5)); bias_batch(L1{}, U1{}); dw_units(L1{}, U4{}, U8{}, Ab);
    chunk_step(EF_P(6)); bias_batch(L1{}, U2{}); dw_units(L1{}, U8{}, U12{}, Ab);
    chunk_step(EF_P(7)); bias_batch(L1{}, U3{}); dw_units(L1{}, U12{}, U16{}, Ab);
    ++S;
    barrier_x(); update_exp(L0{});
    chunk_step(EF_P(8)); bias_batch(L0{}, U0{}); dw_units(L0{}, U0{}, U4{}, Aa);     // stage 2 (pair l = 0: G_1, a_0) shares the parity of stage 0
    chunk_step(EF_P(9)); bias_batch(L0{}, U1{}); dw_units(L0{}, U4{}, U8{}, Aa);
    chunk_step(EF_P(10)); bias_batch(L0{}, U2{}); dw_units(L0{}, U8{}, U12{}, Aa);
    chunk_step(EF_P(11)); bias_batch(L0{}, U3{}); dw_units(L0{}, U12{}, U16{}, Aa);
    ++S;
  }
#undef EF_P
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pieces issued past the end must have landed before this workgroup's LDS is released
  // ---- partial results of this workgroup, un-scaled by its final exponents: dW[l][n][k] (lane holds rows n = 64 gi + 16 x + 4 g + j,
  // column k' = 64 gj + 16 y + r of the PERMUTED A column space: k' = 32 kb2 + 8 group + slot <-> feature 32 kb2 + 16 (slot >> 2) + 4 group + (slot & 3))
  float* part = a.part + size_t(blockIdx.x) * (DW_FLOATS + DB_FLOATS);
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int e = EB[l] - 278;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int kp = 64 * gj + 16 * y + r;
        const int k = 32 * (kp >> 5) + 16 * ((kp >> 2) & 1) + 4 * ((kp >> 3) & 3) + (kp & 3);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          part[l * D * D + (64 * gi + 16 * x + 4 * g + jj) * D + k] = ldexpf(dw[l][x][y][jj], e);
      }
    // bias gradient: the rows of chain wave LI over all tiles (true scale; the side path's contributions are in: the chain waves' last
    // atomics precede the last tile's chunk barriers)
    *reinterpret_cast<float2*>(part + DW_FLOATS + (l * 4 + LI) * D + 2 * lane) = *reinterpret_cast<const float2*>(dbw + l * D + 2 * lane);
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
  for (int o = tid; o < OFF_RING / 4; o += 512) reinterpret_cast<int*>(lds)[o] = 0;   // exchange words, row weights, bias-gradient sums
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
  int* const xch = reinterpret_cast<int*>(lds + OFF_XCH);
  float4* const ring = reinterpret_cast<float4*>(lds + OFF_RING);
  char* const GH = lds + OFF_GH; char* const GL = lds + OFF_GL;
  Slot slot{0, NR};
  const float rcpE = 1.f / float(a.E);
  int EB[3] = {EB_FLOOR, EB_FLOOR, EB_FLOOR};   // the workgroup's running block exponents, kept in step with the gradient waves'
  float* const roww = reinterpret_cast<float*>(lds + OFF_ROWW);
  float* const dbw = reinterpret_cast<float*>(lds + OFF_DBW) + wave * (3 * D);   // shared with gradient wave `wave`, never at the same time
  const size_t bits_off = pad_rows(size_t(a.R)) * D;   // floats in front of the sign bits of a saved activation (chain.h: act_floats)
  // g_0 of the PREVIOUS tile leaves in batches between the first phases of the next tile (one burst of eight stores behind the last
  // stage fills the CU's store path, ~10 B/clk, and the chain waves sit on it: round-5 timeline)
  f32x4 g0v[NB];
  int64_t g0off = -1;
  auto store_g0 = [&](int t0, int t1) {
    if (g0off < 0) return;   // per lane: rows past the end (and the first tile) store nothing
    float* rowp = a.g0 + g0off;
#pragma unroll
    for (int t = 0; t < NB; ++t)
      if (t >= t0 && t < t1) *reinterpret_cast<f32x4*>(rowp + 16 * t + 4 * g) = g0v[t];
  };
  // What a tile needs -- the node gradient gathered by target, y / rstd of the row, sign bits and scale exponents of a_2, a_1, a_0 -- is
  // requested a tile AHEAD (behind the first hand-over of the previous tile: the loads fly under its three stages); at the head of a
  // tile they cost 2.3k cycles of waiting + their issue (profiles/r06_e32_census.txt).
  struct TileIn { f32x4 gy[NB], y[NB]; float rstd; unsigned mb[3]; int ea[3]; int64_t row64; bool live; };
  auto request = [&](TileIn& q, int tile) {
    q.row64 = int64_t(tile) * ST_ROWS + wave * 16 + r;
    q.live = q.row64 < a.R;
    const unsigned row = q.live ? unsigned(q.row64) : 0u;   // lanes past the end read row 0; their gradient is zeroed below
    EdgeRef e = edge_ref(row, unsigned(a.E), rcpE);
    if (unsigned(e.q) >= unsigned(a.E)) { e.b = int(row / unsigned(a.E)); e.q = int(row - unsigned(e.b) * unsigned(a.E)); }   // estimate off by more than one (tiny E, huge B): exact
    const int64_t idst = int64_t(e.b) * a.N + a.dst[e.q];
    load_rows<NB>(q.gy, a.dy + idst * D, g);
    load_rows<NB>(q.y, a.y + size_t(row) * D, g);
    q.rstd = a.rstd[row];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      q.mb[s] = reinterpret_cast<const unsigned*>(a.act[s] + bits_off)[size_t(row) * 4 + g];
      q.ea[s] = a.aexp[s][row];
    }
  };
  TileIn nx;
  request(nx, int(blockIdx.x));
  for (int it = 0; it < my_tiles; ++it) {
    const int tile = int(blockIdx.x) + it * int(gridDim.x);
    EF32_STAMP(0);
    f32x4 gr[NB], yr[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) { gr[t] = nx.gy[t]; yr[t] = nx.y[t]; }
    const float rstd_row = nx.rstd;
    const int64_t row64 = nx.row64;
    const bool live = nx.live;
    unsigned mb[3];
    int ea[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) { mb[s] = nx.mb[s]; ea[s] = nx.ea[s]; }
    store_g0(0, 2);   // the previous tile's g_0 leaves two stores at a time between the phases of this tile
    EF32_STAMP(1);
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
    EF32_STAMP(2);
    // ---- three gradient stages: stage s multiplies W_{3-s}^T; its INPUT g (= G_{3-s}) goes to the gradient waves first
    f32x4 acc[NB];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int l = 2 - s;                       // pair index: 0 = Linear 1 .. 2 = Linear 3 (efuse.hip's order of the partials)
      const float mg = row_amax<NB>(gr);
      const RowScale rs = scale_of(mg);          // rs.E = max(exponent of mg, 12)
      {   // publish the largest Eg_r + E_r of this wave's rows
        int er = rs.E + ea[s];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) er = max(er, __shfl_xor(er, o, 64));
        if (lane == 0) xch[l * 4 + wave] = er;
      }
      barrier_x();                               // X: the four maxima are visible; the gradient waves are done with the previous pair
      {
        int m = EB[l];
#pragma unroll
        for (int w = 0; w < 4; ++w) m = max(m, xch[l * 4 + w]);
        EB[l] = m;
#ifndef EFV_NOHAND
        stage_split(GH, GL, sb, gr, pow2_field(266 + ea[s] - m));   // G'' = G 2^(E_r + 139 - EB)
#endif
#ifndef EFV_NODB
        // bias gradient: gradient wave `wave` sums the staged rows times w_r = 2^(EB - E_r - 139); a row too far below the block
        // maximum for that (its pieces underflow) is added here, exactly, through LDS atomics -- rows of all-zero activations
        const bool far = m - (rs.E + ea[s]) > DB_WINDOW;
        if (g == 0) roww[16 * wave + r] = far ? 0.f : pow2_field(m - ea[s] - 12);
        if (far) {
#pragma unroll
          for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(dbw + l * D + 16 * t + 4 * g + k, gr[t][k]);
        }
#endif
      }
      store_g0(2 + 2 * s, 4 + 2 * s);
      if (s == 0) request(nx, it + 1 < my_tiles ? tile + int(gridDim.x) : tile);   // (past the last tile: any valid rows, never used)
      EF32_STAMP(3 + 3 * s);
      mfma_stage<NB, true, 1>(acc, gr, rs, ring, slot, lane);   // its first chunk barrier publishes the staged G
      EF32_STAMP(4 + 3 * s);
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // bit -> all-ones / zero mask (one v_bfe_i32), then one and (chain.hip: k_edge_bwd)
          const int keep = __builtin_amdgcn_sbfe((int)mb[s], 4 * t + k, 1);
          gr[t][k] = __uint_as_float(__float_as_uint(acc[t][k]) & (unsigned)keep);
        }
      EF32_STAMP(5 + 3 * s);
    }
    // ---- g_0: input of the scatter / fiber-gradient kernel (plain stores: read next, they stay in L2 / the memory-side cache)
#pragma unroll
    for (int t = 0; t < NB; ++t) g0v[t] = gr[t];
    g0off = live ? int64_t(row64) * D : -1;   // (all eight stores of the previous g_0 have been issued: 2 + 3 x 2)
    EF32_STAMP(12);
  }
  store_g0(0, 8);   // the last tile's
}

}  // namespace

namespace bsms {

bool edge_fused32_supported(int64_t D_, int H, int64_t p, int precision) {
  return precision == BSMS_F32 && D_ == 128 && H == 3 && p >= 1 && p <= 7;
}

int launch_edge_fused32_bwd(EdgeFused32Args a, int* nwg_out, hipStream_t s) {
  BSMS_REQUIRE(a.R < (int64_t(1) << 31), BSMS_E_UNSUPPORTED, "edge_fused32_bwd: R = %lld", (long long)a.R);
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_edge_fused32_bwd), LDS_BYTES);
  BSMS_REQUIRE(attr == hipSuccess, BSMS_E_HIP, "edge_fused32_bwd: cannot reserve %d bytes of LDS", LDS_BYTES);
  a.ntiles = int(ceil_div(a.R, ST_ROWS));
  const int nwg = int(std::min<int64_t>(a.ntiles, std::min(device_cu_count(), kEdgeFusedMaxWg)));
  *nwg_out = nwg;
  if (nwg > 0) {
    hipLaunchKernelGGL(k_edge_fused32_bwd, dim3(nwg), dim3(512), LDS_BYTES, s, a);
    BSMS_LAUNCH_CHECK();
  }
  return BSMS_OK;
}

}  // namespace bsms
