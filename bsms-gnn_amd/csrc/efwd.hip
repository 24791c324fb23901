// Edge MLP FORWARD of the bf16 precisions (BSMS_BF16 / BSMS_BF16_NODES), D = 128, hidden = 3, with RESIDENT weights:
//     relu(Ps[src] + Pd[dst] + Wf . fiber) -> (Linear, ReLU) x 2 -> Linear -> LayerNorm -> y (bf16), rstd, fiber
// Reference arithmetic: src/ops/basic.py:6-23 (MLP), :83-92 (fiber, concatenation split by linearity, gmp.hip).
//
// Why a kernel of its own: the generic chain kernel streams every weight chunk through an LDS ring, one
// workgroup barrier per 32-feature chunk, 4-7 compute waves + a loader per CU; with ONE bf16 product per fragment pair the
// MFMAs of a chunk are over long before the barrier round trip is (r04_bf16_edge_pmc.txt: waves parked 59 % of their
// resident cycles, issuing 12-15 %).  At D = 128 the three D x D weights of the edge MLP are 96 KB as bf16: they FIT.  So
// here every workgroup copies the bf16 packs (already in MFMA fragment order) once into LDS, where they
// stay; there is no loader wave, no ring and NO barrier after the prologue.  A workgroup is 16 independent
// waves (4 per SIMD, <= 128 VGPRs) that each take 16-row tiles from a static stride; the hardware interleaves four waves
// per SIMD, which is what hides the gathers, the LDS round trips and the LayerNorm of one wave behind the MFMAs of the
// others.  A-fragments are lane-linear ds_read_b128 (conflict-free, 256 B/clk).
// Arithmetic and its order are those of k_chain_fwd<8, IN_EDGE, OUT_LN, BF> (chain.hip): same operand roundings, the K
// chunks of every accumulator in the order 0..3 starting from the bias, the same LayerNorm expression -- the results are
// bit-identical (profiles/r05_efwd_ab.txt).
#include "chain.h"

using namespace bsms;

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int D = 128, NB = 8;
constexpr int W_BYTES = D * D * 2;            // one weight matrix as bf16 fragments: [c][t][lane] 16 bytes (= the pack bodies)
constexpr int OFF_SIDE = 3 * W_BYTES;         // floats: fiber weights^T [4][D], then the three biases [3][D]
constexpr int SIDE_B = 4 * D;
constexpr int LDS_BYTES = OFF_SIDE + (4 * D + 3 * D) * 4;
#ifndef EFWD_WAVES
#define EFWD_WAVES 16   // (variants for A/B builds: 8 = two waves per SIMD / 256 VGPRs, 12 = three / 168)
#endif
constexpr int WAVES = EFWD_WAVES;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_dst, +1 KB) in lane order (chain.hip: glds16)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ float group_sum(float s) {  // sum over the 4 lane groups holding one row
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
__device__ __forceinline__ void load_rows(f32x4 (&v)[NB], const float* row, int g) {
#pragma unroll
  for (int t = 0; t < NB; ++t) v[t] = *reinterpret_cast<const f32x4*>(row + 16 * t + 4 * g);
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
// relu, then the bf16 rounding that makes the B operand of the next Linear (chain.hip: relu_into + round_block)
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
// acc = bias + W x: every accumulator takes its K chunks in the order 0..3 (chain.hip: mfma_stage_bf)
__device__ __forceinline__ void stage(f32x4 (&acc)[NB], const u32x4 (&bb)[4], const char* W, const float* bias, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const float4 x = *reinterpret_cast<const float4*>(bias + 16 * t + 4 * g);
    acc[t] = f32x4{x.x, x.y, x.z, x.w};
  }
  const u32x4* frag = reinterpret_cast<const u32x4*>(W) + lane;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = mma(frag[(c * NB + t) * 64], bb[c], acc[t]);
}

// row of the [B, E] edge tensor -> (batch, edge) without a 64-bit division (chain.hip: edge_ref; rows < 2^31, launcher)
struct EdgeRef { int b, q; };
__device__ __forceinline__ EdgeRef edge_ref(unsigned row, unsigned E, float rcpE) {
  int b = int(float(row) * rcpE);
  int q = int(row) - b * int(E);
  if (q < 0) { q += int(E); --b; }
  if (q >= int(E)) { q -= int(E); ++b; }
  if (unsigned(q) >= E) { b = int(row / E); q = int(row - unsigned(b) * E); }   // estimate off by more than one (tiny E, huge B): exact
  return EdgeRef{b, q};
}

__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void k_edge_fwd_res(EdgeFwdResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  // ---- prologue: the bodies of the three bf16 weight packs (chain.h: one plane, [t][lane] 16 bytes per 32-feature K chunk --
  // exactly the A-fragment order, written once per call by the block's prepack) go straight from L2 into LDS with LDS-DMA:
  // 96 pieces of 1 KB, six per wave, no registers, no conversion.  The biases ride in the header of each pack's chunk 0.
  {
    constexpr int CHF = kChunkHdrFloats + NB * 256;   // floats per chunk of a one-plane pack (chain.hip: Ring<NB, 1>::CHF)
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    for (int piece = uwave; piece < 96; piece += WAVES) {   // 0..95: pack l = piece / 32, chunk c = (piece / 8) & 3, 1 KB slice k = piece & 7
      const int l = piece >> 5, c = (piece >> 3) & 3, k = piece & 7;
      const float* src = reinterpret_cast<const float*>(a.wp[l]) + size_t(c) * CHF + kChunkHdrFloats + k * 256 + lane * 4;
      glds16(src, __builtin_amdgcn_readfirstlane(lds0 + unsigned(piece) * 1024u));
    }
    float* side = reinterpret_cast<float*>(lds + OFF_SIDE);
    for (int o = tid; o < 4 * D; o += WAVES * 64) side[o] = o < (a.p + 1) * D ? a.wft[o] : 0.f;
    if (tid < 3 * D) side[SIDE_B + tid] = reinterpret_cast<const float*>(a.wp[tid >> 7])[tid & (D - 1)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const float* const wft = reinterpret_cast<const float*>(lds + OFF_SIDE);
  const float* const bias = wft + SIDE_B;
  const float rcpE = 1.f / float(a.E);
  const int stride = int(gridDim.x) * WAVES;
  // plan-order endpoints of this lane's row, fetched one tile ahead: a tile starts with its row gathers, not with a
  // dependent index round trip
  struct Where { int b, i, j; };
  auto locate = [&](int tile) {
    const int64_t row64 = int64_t(tile) * 16 + (lane & 15);
    const EdgeRef e = edge_ref(unsigned(row64 < a.R ? row64 : 0), unsigned(a.E), rcpE);   // a lane past the end reads row 0
    return Where{e.b, a.src[e.q], a.dst[e.q]};
  };
  int tile = int(blockIdx.x) * WAVES + wave;
  if (tile >= a.ntiles) return;
  Where nxt = locate(tile);
  for (; tile < a.ntiles; tile += stride) {
    const Where cur = nxt;
    const int64_t row = int64_t(tile) * 16 + (lane & 15);
    const bool live = row < a.R;
    f32x4 act[NB], acc[NB];
    load_rows(act, a.Ps + (int64_t(cur.b) * a.N + cur.i) * D, g);
    load_rows(acc, a.Pd + (int64_t(cur.b) * a.N + cur.j) * D, g);
    const float* pb = a.pos + cur.b * a.pos_bstride;
    float pi[3], pj[3];
    if (a.p == 2) {   // uniform; the common width loads whole points
      const float2 xi = *reinterpret_cast<const float2*>(pb + int64_t(cur.i) * 2), xj = *reinterpret_cast<const float2*>(pb + int64_t(cur.j) * 2);
      pi[0] = xi.x; pi[1] = xi.y; pj[0] = xj.x; pj[1] = xj.y;
      pi[2] = pj[2] = 0.f;
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int cc = c < a.p ? c : 0;   // uniform clamp: the loads stay unconditional
        pi[c] = pb[int64_t(cur.i) * a.p + cc];
        pj[c] = pb[int64_t(cur.j) * a.p + cc];
      }
    }
    if (tile + stride < a.ntiles) nxt = locate(tile + stride);   // uniform; lands under the stages
    __builtin_amdgcn_sched_barrier(0);   // all gathers of the tile are in flight before the first use
    // ---- input stage (chain.hip: k_chain_fwd IN_EDGE -- the same operations in the same order)
#pragma unroll
    for (int t = 0; t < NB; ++t) act[t] += acc[t];
    float n2 = 0.f, rel[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < a.p) {
        rel[c] = pi[c] - pj[c];
        n2 = fmaf(rel[c], rel[c], n2);
        axpy_features(act, wft + c * D, rel[c], g);
      } else {
        rel[c] = 0.f;
      }
    const float nrm = sqrtf(n2);
    axpy_features(act, wft + a.p * D, nrm, g);
    if (a.fiber_out && live && g == 0) {   // one lane per row keeps the fiber for the backward (fiber_ld(p) = 4 for p <= 3)
      const float f[4] = {a.p > 0 ? rel[0] : nrm, a.p > 1 ? rel[1] : (a.p == 1 ? nrm : 0.f), a.p > 2 ? rel[2] : (a.p == 2 ? nrm : 0.f),
                          a.p == 3 ? nrm : 0.f};
      *reinterpret_cast<float4*>(a.fiber_out + row * 4) = make_float4(f[0], f[1], f[2], f[3]);
    }
    // ---- the three Linears, activations in registers between them
    u32x4 bb[4];
    relu_pack(bb, act);
    stage(acc, bb, lds, bias, lane);
    relu_pack(bb, acc);
    stage(acc, bb, lds + W_BYTES, bias + D, lane);
    relu_pack(bb, acc);
    stage(acc, bb, lds + 2 * W_BYTES, bias + 2 * D, lane);
    // ---- LayerNorm(elementwise_affine=False), eps 1e-5  (ops/basic.py:18; chain.hip: k_chain_fwd OUT_LN)
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t) s += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    const float mean = group_sum(s) * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[t][r] -= mean;
        ss = fmaf(acc[t][r], acc[t][r], ss);
      }
    ss = group_sum(ss);
    const float rstd = 1.f / sqrtf(ss * (1.f / D) + 1e-5f);
    if (!live) continue;
    u32x2* yp = reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(a.y) + row * D + 4 * g);
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      f32x4 v = acc[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rstd;
      yp[4 * t] = u32x2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])};
    }
    if (a.rstd && g == 0) a.rstd[row] = rstd;
  }
}

int device_cus() { return device_cu_count(); }   // common.h: cached per device

}  // namespace

namespace bsms {

bool edge_fwd_res_supported(int64_t D_, int H, int64_t p, int precision) {
  return precision != BSMS_F32 && D_ == 128 && H == 3 && p >= 1 && p <= 3;
}

int launch_edge_fwd_res(EdgeFwdResArgs a, hipStream_t s) {
  BSMS_REQUIRE(a.R < (int64_t(1) << 31) && a.p >= 1 && a.p <= 3, BSMS_E_UNSUPPORTED, "edge_fwd_res: R = %lld, p = %d", (long long)a.R, a.p);
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_edge_fwd_res), LDS_BYTES);
  BSMS_REQUIRE(attr == hipSuccess, BSMS_E_HIP, "edge_fwd_res: cannot reserve %d bytes of LDS", LDS_BYTES);
  a.ntiles = int(ceil_div(a.R, 16));
  const int nwg = int(std::min<int64_t>(ceil_div(a.ntiles, WAVES), device_cus()));
  if (nwg > 0) {
    hipLaunchKernelGGL(k_edge_fwd_res, dim3(nwg), dim3(WAVES * 64), LDS_BYTES, s, a);
    BSMS_LAUNCH_CHECK();
  }
  return BSMS_OK;
}

}  // namespace bsms
