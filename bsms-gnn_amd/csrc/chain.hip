// Chain kernels (see chain.h for the register-layout idea) + weight prepack.
#include "chain.h"

using namespace bsms;

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Register layout ("chain layout", see chain.h): a wave owns 16 rows; lane l <-> row (l & 15), group g = l >> 4.
// For every 16-feature block t the lane holds features 16 t + 4 g + {0,1,2,3} as one f32x4.

// ---------------------------------------------------------------------------------- prepack ----
// FRAG  : dst[((kb*NB + t)*64 + lane)*4 + s] = M[16 t + (lane & 15)][16 kb + 4 (lane >> 4) + s]
// with M[n][k] = W[row0+n][col0+k] (FRAG) or W[row0+k][col0+n] (FRAG_T); NB = N/16, kb < K/16.
__global__ __launch_bounds__(256) void k_prepack(PackTable tab) {
  const PackDesc d = tab.d[blockIdx.y];
  const int total = d.N * d.K;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    int n, k;
    if (d.kind == PACK_TRANSPOSE) {
      k = o / d.N;
      n = o % d.N;
    } else {
      const int sidx = o & 3, lane = (o >> 2) & 63;
      const int rest = o >> 8, nb = d.N >> 4;
      const int t = rest % nb, kb = rest / nb;
      n = 16 * t + (lane & 15);
      k = 16 * kb + 4 * (lane >> 4) + sidx;
    }
    const float v = (d.kind == PACK_FRAG_T) ? d.W[int64_t(d.row0 + k) * d.ld + d.col0 + n]
                                            : d.W[int64_t(d.row0 + n) * d.ld + d.col0 + k];
    d.dst[o] = v;
  }
}

// ------------------------------------------------------------------------ register-tile helpers
template <int NB>
__device__ __forceinline__ void zero_tile(f32x4 (&v)[NB]) {
#pragma unroll
  for (int t = 0; t < NB; ++t) v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// lane's row pointer (nullptr = row out of range -> zeros); 4 lanes of a row read 64 contiguous bytes
template <int NB>
__device__ __forceinline__ void load_rows(f32x4 (&v)[NB], const float* row, int g) {
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const float4 x = row ? *reinterpret_cast<const float4*>(row + 16 * t + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[t] = f32x4{x.x, x.y, x.z, x.w};
  }
}

template <int NB, bool ACCUM>
__device__ __forceinline__ void store_rows(const f32x4 (&v)[NB], float* row, int g, int mode = 0) {
  if (!row) return;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    float4* p = reinterpret_cast<float4*>(row + 16 * t + 4 * g);
    float4 x = make_float4(v[t][0], v[t][1], v[t][2], v[t][3]);
    if (ACCUM) {
      const float4 o = *p;
      x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
    }
    if (mode == 1) {
      __builtin_nontemporal_store(v[t], reinterpret_cast<f32x4*>(p));
    } else if (mode == 2) {
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v[t]) : "memory");
    } else {
      *p = x;
    }
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

// One Linear: acc[t] += sum_kb W(kb,t) * act[kb] on v_mfma_f32_16x16x4_f32.  The A operand of step s is the
// weight fragment W[16t + (l&15)][16kb + 4g + s], the B operand this lane's own act[kb][s].  Weight chunks
// (32 input features x all outputs) stream L2 -> registers -> LDS ring; one __syncthreads per chunk.
// Two accumulators are interleaved so back-to-back MFMAs are independent (40-cycle dependent latency vs
// 32-cycle issue).  All waves of the workgroup must call this together.
// ---------------------------------------------------------------------------- weight streaming
// Workgroup = 4 compute waves + 1 LOADER wave.  The loader streams the weight packs of all stages of the
// kernel, chunk by chunk (a chunk = 32 input features x all outputs, 16 KB at D = 128), from L2 into a 2-deep
// LDS ring, always one chunk ahead of the compute waves and straight across stage boundaries; with a stage's
// first chunk it also drops the stage's bias into LDS.  One workgroup barrier per chunk.
//
// Why a separate wave: vmcnt retires loads and stores through one in-order counter and hipcc waits vmcnt(0)
// whenever both kinds are outstanding, so a compute wave that fetched its own weights would stop at every chunk
// until its activation stores had reached memory -- MFMA phases and HBM phases then add up instead of
// overlapping (measured: kernel time = MFMA time + store time).  Compute waves execute NO vmcnt wait in the
// steady state; their stores drain in the background.
template <int NB>
struct Ring {
  static constexpr int KB_PER = NB >= 2 ? 2 : 1;  // 16-feature k blocks per chunk
  static constexpr int NCH = NB / KB_PER;         // chunks per stage
  static constexpr int CH = KB_PER * NB * 64;     // float4 per chunk
  static constexpr int D = NB * 16;
  static constexpr size_t lds_bytes = size_t(2) * CH * sizeof(float4) + size_t(2) * D * sizeof(float);
  static_assert(CH % 64 == 0, "chunk must be a whole number of float4 per loader lane");
  static_assert(NB % 2 == 0, "accumulators are processed in pairs");
};

// workgroup barrier that waits for LDS traffic only (never for vmcnt)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NB>
__device__ __forceinline__ float* ring_bias(float4* lds, int stage) {
  return reinterpret_cast<float*>(lds + 2 * Ring<NB>::CH) + (stage & 1) * Ring<NB>::D;
}

// the loader wave's whole life
template <int NB>
__device__ __forceinline__ void loader_run(const float4* const* wseq, const float* const* bseq, int nseq, float4* lds,
                                           int lane) {
  using R = Ring<NB>;
  __builtin_amdgcn_s_setprio(3);                   // the loader must never be the wave the others wait for
  constexpr int ROUND = 16;                        // float4 per lane per round (64 VGPRs)
  constexpr int PER = R::CH / 64;                  // float4 per lane per chunk
  int j = 0;
  for (int s = 0; s < nseq; ++s) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(wseq[s]);
    for (int c = 0; c < R::NCH; ++c, ++j) {
      f32x4* dst = reinterpret_cast<f32x4*>(lds) + (j & 1) * R::CH;
      const f32x4* src = wp + c * R::CH;
#pragma unroll
      for (int r0 = 0; r0 < PER; r0 += ROUND) {
        f32x4 v[ROUND];
#pragma unroll
        for (int i = 0; i < ROUND; ++i)
          if (r0 + i < PER) v[i] = src[(r0 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < ROUND; ++i)
          if (r0 + i < PER) dst[(r0 + i) * 64 + lane] = v[i];
      }
      if (c == 0 && bseq && bseq[s]) {
        float* bl = ring_bias<NB>(lds, s);
        for (int f = lane; f < R::D; f += 64) bl[f] = bseq[s][f];
      }
      lds_barrier();                               // publishes chunk j (compute waves arrive after chunk j-1)
    }
  }
  lds_barrier();                                   // pairs with the compute waves' barrier after the last chunk
}

// One Linear on the compute waves: acc[t] += sum_kb W(kb,t) * act[kb] with v_mfma_f32_16x16x4_f32.  The A operand
// of step s is the weight fragment W[16t + (l&15)][16kb + 4g + s] (lane-linear ds_read_b128 from the ring), the B
// operand this lane's own act[kb][s].  `j` = running chunk counter (ring slot = j & 1).  Two accumulators are
// interleaved so back-to-back MFMAs are independent (40-cycle dependent latency vs 32-cycle issue).
// `store_row` (nullable): this lane's row of an HBM tensor that receives `act`; issued with the first chunk so the
// store has the whole stage to drain.
template <int NB>
__device__ __forceinline__ void mfma_stage(f32x4 (&acc)[NB], const f32x4 (&act)[NB], float4* lds, int& j, int lane,
                                           float* store_row = nullptr, int store_mode = 0) {
  using R = Ring<NB>;
  store_rows<NB, false>(act, store_row, lane >> 4, store_mode);
#pragma unroll
  for (int c = 0; c < R::NCH; ++c) {
    const float4* cur = lds + ((j + c) & 1) * R::CH;
    // fragment reads are written one (t, t+1) pair ahead of the 8 MFMAs that consume them; hipcc sinks them back
    // next to their use (pinning the order with sched_barrier measured no gain: other waves hide the LDS latency)
    constexpr int HP = NB / 2, NP = R::KB_PER * HP;      // pairs per chunk
    float4 w0 = cur[lane], w1 = cur[64 + lane];
#pragma unroll
    for (int pidx = 0; pidx < NP; ++pidx) {
      const int kk = pidx / HP, t = 2 * (pidx % HP), kb = c * R::KB_PER + kk;
      float4 n0 = w0, n1 = w1;
      if (pidx + 1 < NP) {
        const int kk2 = (pidx + 1) / HP, t2 = 2 * ((pidx + 1) % HP);
        n0 = cur[(kk2 * NB + t2) * 64 + lane];
        n1 = cur[(kk2 * NB + t2 + 1) * 64 + lane];
      }
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, act[kb][0], acc[t], 0, 0, 0);
      acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, act[kb][0], acc[t + 1], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, act[kb][1], acc[t], 0, 0, 0);
      acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, act[kb][1], acc[t + 1], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, act[kb][2], acc[t], 0, 0, 0);
      acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, act[kb][2], acc[t + 1], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, act[kb][3], acc[t], 0, 0, 0);
      acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, act[kb][3], acc[t + 1], 0, 0, 0);
      w0 = n0;
      w1 = n1;
    }
    lds_barrier();
  }
  j += R::NCH;
}

// acc = bias of sequence entry `stage` (from the ring) or 0
template <int NB>
__device__ __forceinline__ void init_acc(f32x4 (&acc)[NB], float4* lds, int stage, bool has_bias, int lg) {
  if (has_bias) {
    const float* bl = ring_bias<NB>(lds, stage);
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const float4 x = *reinterpret_cast<const float4*>(bl + 16 * t + 4 * lg);
      acc[t] = f32x4{x.x, x.y, x.z, x.w};
    }
  } else {
    zero_tile<NB>(acc);
  }
}

template <int NB>
__device__ __forceinline__ void relu_into(f32x4 (&dst)[NB], const f32x4 (&srcv)[NB]) {
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[t][r] = fmaxf(srcv[t][r], 0.f);
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

// -------------------------------------------------------------------------------- forward chain
template <int NB, int IN, int OUT>
__global__ __launch_bounds__(kChainThreads) void k_chain_fwd(ChainFwdArgs a) {
  constexpr int D = NB * 16;
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4;
  if (wave == kComputeWaves) {  // loader wave (uniform branch)
    loader_run<NB>(a.wseq, a.bseq, a.nseq, lds, lane);
    return;
  }
  const int64_t row = int64_t(blockIdx.x) * kTileRows + wave * 16 + (lane & 15);
  const bool live = row < a.R;

  f32x4 act[NB], acc[NB];

  // ---- input stage
  if (IN == IN_ROWS || IN == IN_ROWS2) {
    load_rows<NB>(act, live ? a.x + row * D : nullptr, lg);
  } else if (IN == IN_SMALL) {
    if (live) {
      load_features<NB>(act, a.bias_in, lg);
      for (int k = 0; k < a.K0; ++k) axpy_features<NB>(act, a.w0t + k * D, a.x[row * a.K0 + k], lg);
      relu_into<NB>(act, act);
    } else {
      zero_tile<NB>(act);
    }
  } else {  // IN_EDGE: relu(Ps[src] + Pd[dst] + Wf . [pos_i - pos_j, |pos_i - pos_j|])   (ops/basic.py:70-92)
    if (live) {
      const int b = int(row / a.E), q = int(row - int64_t(b) * a.E);
      const int i = a.src[q], j = a.dst[q];
      load_rows<NB>(act, a.Ps + (int64_t(b) * a.N + i) * D, lg);
      load_rows<NB>(acc, a.Pd + (int64_t(b) * a.N + j) * D, lg);
#pragma unroll
      for (int t = 0; t < NB; ++t) act[t] += acc[t];
      const float* pb = a.pos + b * a.pos_bstride;
      float n2 = 0.f;
      for (int c = 0; c < a.p; ++c) {
        const float rel = pb[int64_t(i) * a.p + c] - pb[int64_t(j) * a.p + c];
        n2 = fmaf(rel, rel, n2);
        axpy_features<NB>(act, a.w0t + c * D, rel, lg);
      }
      axpy_features<NB>(act, a.w0t + a.p * D, sqrtf(n2), lg);
      relu_into<NB>(act, act);
    } else {
      zero_tile<NB>(act);
    }
  }

  // ---- MFMA stages.  The activation entering a stage is stored to HBM from inside that stage (mfma_stage).
  int j = 0, sq = 0;  // ring chunk counter, index into the loader's weight sequence
  lds_barrier();      // chunk 0 (+ bias of the first stage) is in the ring
  float* pending = ((IN == IN_SMALL || IN == IN_EDGE) && live && a.store_in) ? a.store_in + row * D : nullptr;
  if (a.nstage == 0 && pending) store_rows<NB, false>(act, pending, lg);
  for (int l = 0; l < a.nstage; ++l) {
    init_acc<NB>(acc, lds, sq, a.bias[l] != nullptr, lg);
    mfma_stage<NB>(acc, act, lds, j, lane, pending, a.store_mode);
    ++sq;
    pending = nullptr;
    if (IN == IN_ROWS2 && l == 0) {
      load_rows<NB>(act, live ? a.x2 + row * D : nullptr, lg);
      mfma_stage<NB>(acc, act, lds, j, lane);
      ++sq;
    }
    const bool last = (l == a.nstage - 1);
    if (!last || OUT == OUT_SMALL) {
      relu_into<NB>(act, acc);
      if (!last) pending = (live && a.store[l]) ? a.store[l] + row * D : nullptr;
      else if (live && a.store[l]) store_rows<NB, false>(act, a.store[l] + row * D, lg);
    }
  }
  if (!live) return;

  // ---- output
  if (OUT == OUT_LN) {  // LayerNorm(elementwise_affine=False), eps 1e-5  (ops/basic.py:18)
    const float mean = row_sum<NB>(acc) * (1.f / D);
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
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] *= rstd;
    if (a.yln) store_rows<NB, false>(acc, a.yln + row * D, lg);
    if (a.rstd && lg == 0) a.rstd[row] = rstd;
    if (a.resid) {
      load_rows<NB>(act, a.resid + row * D, lg);
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] += act[t];
    }
    store_rows<NB, false>(acc, a.y + row * D, lg, a.out_mode);
  } else if (OUT == OUT_PLAIN) {
    if (a.accumulate) store_rows<NB, true>(acc, a.y + row * D, lg);
    else store_rows<NB, false>(acc, a.y + row * D, lg);
  } else {  // OUT_SMALL: the narrow last Linear (decoder, models/model.py:22) on the VALU
    for (int c = 0; c < a.C; ++c) {
      const float v = dot_features<NB>(act, a.wout + c * D, lg);
      if (lg == 0) a.y[row * a.C + c] = v + a.bout[c];
    }
  }
}

// ------------------------------------------------------------------------------- backward chain
template <int NB>
__device__ __forceinline__ void mask_by(f32x4 (&gr)[NB], const float* act_row, int lg) {
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const float4 m = *reinterpret_cast<const float4*>(act_row + 16 * t + 4 * lg);
    gr[t][0] = m.x > 0.f ? gr[t][0] : 0.f;
    gr[t][1] = m.y > 0.f ? gr[t][1] : 0.f;
    gr[t][2] = m.z > 0.f ? gr[t][2] : 0.f;
    gr[t][3] = m.w > 0.f ? gr[t][3] : 0.f;
  }
}

template <int NB, int GIN, int FIRST>
__global__ __launch_bounds__(kChainThreads) void k_chain_bwd(ChainBwdArgs a) {
  constexpr int D = NB * 16;
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4;
  if (wave == kComputeWaves) {  // loader wave (uniform branch)
    loader_run<NB>(a.wseq, nullptr, a.nseq, lds, lane);
    return;
  }
  const int64_t row = int64_t(blockIdx.x) * kTileRows + wave * 16 + (lane & 15);
  const bool live = row < a.R;

  f32x4 g[NB], acc[NB];
  zero_tile<NB>(g);
  if (live) {
    if (GIN == G_SMALL) {  // g = (dy . W_out) masked by the last hidden activation
      for (int c = 0; c < a.C; ++c) axpy_features<NB>(g, a.wout + c * D, a.dy[row * a.C + c], lg);
      mask_by<NB>(g, a.mask_in + row * D, lg);
    } else {
      const float* dyrow;
      if (GIN == G_EDGE_LN) {  // autograd of scatter_sum: gather the node gradient by target
        const int b = int(row / a.E), q = int(row - int64_t(b) * a.E);
        dyrow = a.dy + (int64_t(b) * a.N + a.dst[q]) * D;
      } else {
        dyrow = a.dy + row * D;
      }
      load_rows<NB>(g, dyrow, lg);
      load_rows<NB>(acc, a.yln + row * D, lg);  // acc = normalised output y
      // LayerNorm backward (no affine): dz = rstd * (dy - mean(dy) - y * mean(dy * y))
      const float m1 = row_sum<NB>(g) * (1.f / D);
      float s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s2 = fmaf(g[t][r], acc[t][r], s2);
      s2 = group_sum(s2);
      const float m2 = s2 * (1.f / D), rs = a.rstd[row];
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[t][r] = rs * (g[t][r] - m1 - acc[t][r] * m2);
    }
  }
  // The gradient entering a stage is stored to HBM from inside that stage (mfma_stage), so the store has a whole
  // stage to drain before the next vmcnt wait (the ReLU-mask rows at the end of the stage).
  int j = 0;
  lds_barrier();      // chunk 0 is in the ring
  float* pending = (live && a.gstore[0]) ? a.gstore[0] + row * D : nullptr;

  for (int k = 0; k < a.nstage; ++k) {
    zero_tile<NB>(acc);
    mfma_stage<NB>(acc, g, lds, j, lane, pending, a.store_mode);
    const bool masked = live && a.mask[k];
    if (masked) load_rows<NB>(g, a.mask[k] + row * D, lg);            // g is consumed: reuse it for the mask rows
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) g[t][r] = (!masked || g[t][r] > 0.f) ? acc[t][r] : 0.f;
    pending = (live && a.gstore[k + 1]) ? a.gstore[k + 1] + row * D : nullptr;
  }

  if (FIRST != F_NONE) {
    zero_tile<NB>(acc);
    mfma_stage<NB>(acc, g, lds, j, lane, pending);
    pending = nullptr;
    if (live && a.dres) {
      f32x4 r[NB];
      load_rows<NB>(r, a.dres + row * D, lg);
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] += r[t];
    }
    if (FIRST == F_HEADS2) {
      f32x4 acc2[NB];
      zero_tile<NB>(acc2);
      mfma_stage<NB>(acc2, g, lds, j, lane);
      if (live) {
        store_rows<NB, false>(acc, a.dx + row * D, lg);
        store_rows<NB, false>(acc2, a.dx2 + row * D, lg);
      }
    } else if (live) {
      store_rows<NB, false>(acc, a.dx + row * D, lg);
    }
  }
  if (pending) store_rows<NB, false>(g, pending, lg);
}

template <int NB, int IN, int OUT>
int launch_fwd_t(const ChainFwdArgs& a0, hipStream_t s) {
  ChainFwdArgs a = a0;
  a.nseq = 0;
  for (int l = 0; l < a.nstage; ++l) {  // the loader follows exactly the compute waves' stage order
    a.wseq[a.nseq] = a.wp[l]; a.bseq[a.nseq++] = a.bias[l];
    if (IN == IN_ROWS2 && l == 0) { a.wseq[a.nseq] = a.wp0b; a.bseq[a.nseq++] = nullptr; }
  }
  const size_t lds = Ring<NB>::lds_bytes;
  hipLaunchKernelGGL((k_chain_fwd<NB, IN, OUT>), dim3((unsigned)ceil_div(a.R, kTileRows)), dim3(kChainThreads), lds, s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
// Only the combinations the path uses are instantiated (each is a large unrolled kernel).
template <int NB>
int launch_fwd_n(int in_mode, int out_mode, const ChainFwdArgs& a, hipStream_t s) {
#define BSMS_FWD(I, O) \
  if (in_mode == I && out_mode == O) return launch_fwd_t<NB, I, O>(a, s)
  BSMS_FWD(IN_ROWS, OUT_PLAIN);   // node pre-projection x W^T
  BSMS_FWD(IN_ROWS2, OUT_PLAIN);  // its input gradient
  BSMS_FWD(IN_EDGE, OUT_LN);      // edge MLP
  BSMS_FWD(IN_ROWS2, OUT_LN);     // node MLP on [x, aggr]
  BSMS_FWD(IN_SMALL, OUT_LN);     // encoder
  BSMS_FWD(IN_ROWS, OUT_SMALL);   // decoder
  BSMS_FWD(IN_ROWS, OUT_LN);      // generic D -> D MLP
#undef BSMS_FWD
  BSMS_FAIL(BSMS_E_UNSUPPORTED, "chain_fwd: in/out mode (%d,%d) not built", in_mode, out_mode);
}

template <int NB, int GIN, int FIRST>
int launch_bwd_t(const ChainBwdArgs& a0, hipStream_t s) {
  ChainBwdArgs a = a0;
  a.nseq = 0;
  for (int k = 0; k < a.nstage; ++k) a.wseq[a.nseq++] = a.wpt[k];
  if (FIRST != F_NONE) a.wseq[a.nseq++] = a.wh0;
  if (FIRST == F_HEADS2) a.wseq[a.nseq++] = a.wh1;
  const size_t lds = Ring<NB>::lds_bytes;
  hipLaunchKernelGGL((k_chain_bwd<NB, GIN, FIRST>), dim3((unsigned)ceil_div(a.R, kTileRows)), dim3(kChainThreads), lds, s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
template <int NB>
int launch_bwd_n(int gin, int first, const ChainBwdArgs& a, hipStream_t s) {
#define BSMS_BWD(G, F) \
  if (gin == G && first == F) return launch_bwd_t<NB, G, F>(a, s)
  BSMS_BWD(G_ROWS_LN, F_HEADS2);  // node MLP
  BSMS_BWD(G_EDGE_LN, F_NONE);    // edge MLP
  BSMS_BWD(G_ROWS_LN, F_NONE);    // encoder
  BSMS_BWD(G_SMALL, F_HEADS1);    // decoder
  BSMS_BWD(G_ROWS_LN, F_HEADS1);  // generic D -> D MLP
#undef BSMS_BWD
  BSMS_FAIL(BSMS_E_UNSUPPORTED, "chain_bwd: grad/first mode (%d,%d) not built", gin, first);
}

}  // namespace

namespace bsms {

int launch_prepack(const PackTable& t, hipStream_t s) {
  if (t.n == 0) return BSMS_OK;
  int biggest = 0;
  for (int i = 0; i < t.n; ++i) biggest = biggest > t.d[i].N * t.d[i].K ? biggest : t.d[i].N * t.d[i].K;
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(biggest, 256), 64);
  hipLaunchKernelGGL(k_prepack, dim3(gx, t.n), dim3(256), 0, s, t);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

int launch_chain_fwd(int D, int in_mode, int out_mode, const ChainFwdArgs& a, hipStream_t s) {
  if (a.R == 0) return BSMS_OK;
  switch (D) {
    case 32: return launch_fwd_n<2>(in_mode, out_mode, a, s);
    case 64: return launch_fwd_n<4>(in_mode, out_mode, a, s);
    case 128: return launch_fwd_n<8>(in_mode, out_mode, a, s);
    case 256: return launch_fwd_n<16>(in_mode, out_mode, a, s);
  }
  BSMS_FAIL(BSMS_E_UNSUPPORTED, "latent width D=%d not supported (32, 64, 128, 256)", D);
}

int launch_chain_bwd(int D, int gin, int first, const ChainBwdArgs& a, hipStream_t s) {
  if (a.R == 0) return BSMS_OK;
  switch (D) {
    case 32: return launch_bwd_n<2>(gin, first, a, s);
    case 64: return launch_bwd_n<4>(gin, first, a, s);
    case 128: return launch_bwd_n<8>(gin, first, a, s);
    case 256: return launch_bwd_n<16>(gin, first, a, s);
  }
  BSMS_FAIL(BSMS_E_UNSUPPORTED, "latent width D=%d not supported (32, 64, 128, 256)", D);
}

}  // namespace bsms
