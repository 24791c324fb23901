// Chain kernels (see chain.h for the register-layout idea) + weight prepack.
#include "chain.h"

using namespace bsms;

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ int feat_of(int blk, int s, int hh) { return 32 * blk + (s & 3) + 8 * (s >> 2) + 4 * hh; }

// ---------------------------------------------------------------------------------- prepack ----
__global__ __launch_bounds__(256) void k_prepack(PackTable tab) {
  const PackDesc d = tab.d[blockIdx.y];
  const int total = d.N * d.K;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    int n, k;
    if (d.kind == PACK_TRANSPOSE) {
      k = o / d.N;
      n = o % d.N;
    } else {
      // o = (((kb*NT + t)*4 + s4)*64 + lane)*4 + c
      const int c = o & 3, lane = (o >> 2) & 63, s4 = (o >> 8) & 3;
      const int rest = o >> 10, nt = d.N >> 5;
      const int t = rest % nt, kb = rest / nt;
      n = 32 * t + (lane & 31);
      k = feat_of(kb, 4 * s4 + c, lane >> 5);
    }
    const float v = (d.kind == PACK_FRAG_T) ? d.W[int64_t(d.row0 + k) * d.ld + d.col0 + n]
                                            : d.W[int64_t(d.row0 + n) * d.ld + d.col0 + k];
    d.dst[o] = v;
  }
}

// ------------------------------------------------------------------------ register-tile helpers
template <int NT>
__device__ __forceinline__ void zero_tile(f32x16 (&v)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[t][r] = 0.f;
}

// lane's row pointer (nullptr = row out of range -> zeros)
template <int NT>
__device__ __forceinline__ void load_rows(f32x16 (&v)[NT], const float* row, int hh) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 x = row ? *reinterpret_cast<const float4*>(row + 32 * t + 8 * q + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[t][4 * q + 0] = x.x; v[t][4 * q + 1] = x.y; v[t][4 * q + 2] = x.z; v[t][4 * q + 3] = x.w;
    }
}

template <int NT, bool ACCUM>
__device__ __forceinline__ void store_rows(const f32x16 (&v)[NT], float* row, int hh) {
  if (!row) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4* p = reinterpret_cast<float4*>(row + 32 * t + 8 * q + 4 * hh);
      float4 x = make_float4(v[t][4 * q + 0], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]);
      if (ACCUM) {
        float4 o = *p;
        x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
      }
      *p = x;
    }
}

// bias (or any per-feature vector) in this lane's feature order
template <int NT>
__device__ __forceinline__ void load_features(f32x16 (&v)[NT], const float* vec, int hh) {
  load_rows<NT>(v, vec, hh);
}

template <int NT>
__device__ __forceinline__ float row_sum(const f32x16 (&v)[NT]) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[t][r];
  return s + __shfl_xor(s, 32, 64);
}

// One Linear: acc[t] += sum_kb W(kb) * act[kb].  Weight chunks stream L2 -> registers -> LDS ring;
// one __syncthreads per 32-feature K block.  All waves of the workgroup must call this together.
template <int NT>
__device__ __forceinline__ void gemm_stage(f32x16 (&acc)[NT], const f32x16 (&act)[NT], const float4* __restrict__ wp,
                                           float4* lds, int tid, int lane) {
  constexpr int CH = NT * 256;  // float4 per chunk
  float4 st[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) st[i] = wp[i * 256 + tid];
#pragma unroll
  for (int i = 0; i < NT; ++i) lds[i * 256 + tid] = st[i];
  __syncthreads();
#pragma unroll
  for (int kb = 0; kb < NT; ++kb) {
    const float4* cur = lds + (kb & 1) * CH;
    float4* nxt = lds + ((kb + 1) & 1) * CH;
    if (kb + 1 < NT) {
#pragma unroll
      for (int i = 0; i < NT; ++i) st[i] = wp[(kb + 1) * CH + i * 256 + tid];
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 w = cur[(t * 4 + s4) * 64 + lane];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, act[kb][4 * s4 + 0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, act[kb][4 * s4 + 1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, act[kb][4 * s4 + 2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, act[kb][4 * s4 + 3], acc[t], 0, 0, 0);
      }
    }
    if (kb + 1 < NT) {
#pragma unroll
      for (int i = 0; i < NT; ++i) nxt[i * 256 + tid] = st[i];
    }
    __syncthreads();
  }
}

template <int NT>
__device__ __forceinline__ void relu_into(f32x16 (&dst)[NT], const f32x16 (&srcv)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[t][r] = fmaxf(srcv[t][r], 0.f);
}

// v += scale * vec (vec in lane feature order)
template <int NT>
__device__ __forceinline__ void axpy_features(f32x16 (&v)[NT], const float* vec, float scale, int hh) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w = *reinterpret_cast<const float4*>(vec + 32 * t + 8 * q + 4 * hh);
      v[t][4 * q + 0] = fmaf(scale, w.x, v[t][4 * q + 0]);
      v[t][4 * q + 1] = fmaf(scale, w.y, v[t][4 * q + 1]);
      v[t][4 * q + 2] = fmaf(scale, w.z, v[t][4 * q + 2]);
      v[t][4 * q + 3] = fmaf(scale, w.w, v[t][4 * q + 3]);
    }
}

template <int NT>
__device__ __forceinline__ float dot_features(const f32x16 (&v)[NT], const float* vec, int hh) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w = *reinterpret_cast<const float4*>(vec + 32 * t + 8 * q + 4 * hh);
      s = fmaf(v[t][4 * q + 0], w.x, s);
      s = fmaf(v[t][4 * q + 1], w.y, s);
      s = fmaf(v[t][4 * q + 2], w.z, s);
      s = fmaf(v[t][4 * q + 3], w.w, s);
    }
  return s + __shfl_xor(s, 32, 64);
}

// -------------------------------------------------------------------------------- forward chain
template <int NT, int IN, int OUT>
__global__ __launch_bounds__(256) void k_chain_fwd(ChainFwdArgs a) {
  constexpr int D = NT * 32;
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int64_t row = int64_t(blockIdx.x) * kTileRows + wave * 32 + (lane & 31);
  const bool live = row < a.R;

  f32x16 act[NT], acc[NT];

  // ---- input stage
  if (IN == IN_ROWS || IN == IN_ROWS2) {
    load_rows<NT>(act, live ? a.x + row * D : nullptr, hh);
  } else if (IN == IN_SMALL) {
    if (live) {
      load_features<NT>(act, a.bias_in, hh);
      for (int k = 0; k < a.K0; ++k) axpy_features<NT>(act, a.w0t + k * D, a.x[row * a.K0 + k], hh);
      relu_into<NT>(act, act);
      store_rows<NT, false>(act, a.store_in ? a.store_in + row * D : nullptr, hh);
    } else {
      zero_tile<NT>(act);
    }
  } else {  // IN_EDGE: relu(Ps[src] + Pd[dst] + Wf . [pos_i - pos_j, |pos_i - pos_j|])   (ops/basic.py:70-92)
    if (live) {
      const int b = int(row / a.E), q = int(row - int64_t(b) * a.E);
      const int i = a.src[q], j = a.dst[q];
      load_rows<NT>(act, a.Ps + (int64_t(b) * a.N + i) * D, hh);
      load_rows<NT>(acc, a.Pd + (int64_t(b) * a.N + j) * D, hh);
#pragma unroll
      for (int t = 0; t < NT; ++t) act[t] += acc[t];
      const float* pb = a.pos + b * a.pos_bstride;
      float n2 = 0.f;
      for (int c = 0; c < a.p; ++c) {
        const float rel = pb[int64_t(i) * a.p + c] - pb[int64_t(j) * a.p + c];
        n2 = fmaf(rel, rel, n2);
        axpy_features<NT>(act, a.w0t + c * D, rel, hh);
      }
      axpy_features<NT>(act, a.w0t + a.p * D, sqrtf(n2), hh);
      relu_into<NT>(act, act);
      store_rows<NT, false>(act, a.store_in ? a.store_in + row * D : nullptr, hh);
    } else {
      zero_tile<NT>(act);
    }
  }

  // ---- MFMA stages
  for (int l = 0; l < a.nstage; ++l) {
    if (a.bias[l]) load_features<NT>(acc, a.bias[l], hh);
    else zero_tile<NT>(acc);
    gemm_stage<NT>(acc, act, a.wp[l], lds, tid, lane);
    if (IN == IN_ROWS2 && l == 0) {
      load_rows<NT>(act, live ? a.x2 + row * D : nullptr, hh);
      gemm_stage<NT>(acc, act, a.wp0b, lds, tid, lane);
    }
    const bool last = (l == a.nstage - 1);
    if (!last || OUT == OUT_SMALL) {
      relu_into<NT>(act, acc);
      if (live && a.store[l]) store_rows<NT, false>(act, a.store[l] + row * D, hh);
    }
  }
  if (!live) return;

  // ---- output
  if (OUT == OUT_LN) {  // LayerNorm(elementwise_affine=False), eps 1e-5  (ops/basic.py:18)
    const float mean = row_sum<NT>(acc) * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[t][r] -= mean;
        ss = fmaf(acc[t][r], acc[t][r], ss);
      }
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = 1.f / sqrtf(ss * (1.f / D) + 1e-5f);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] *= rstd;
    if (a.yln) store_rows<NT, false>(acc, a.yln + row * D, hh);
    if (a.rstd && hh == 0) a.rstd[row] = rstd;
    if (a.resid) {
      load_rows<NT>(act, a.resid + row * D, hh);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] += act[t];
    }
    store_rows<NT, false>(acc, a.y + row * D, hh);
  } else if (OUT == OUT_PLAIN) {
    if (a.accumulate) store_rows<NT, true>(acc, a.y + row * D, hh);
    else store_rows<NT, false>(acc, a.y + row * D, hh);
  } else {  // OUT_SMALL: the narrow last Linear (decoder, models/model.py:22) on the VALU
    for (int c = 0; c < a.C; ++c) {
      const float v = dot_features<NT>(act, a.wout + c * D, hh);
      if (hh == 0) a.y[row * a.C + c] = v + a.bout[c];
    }
  }
}

// ------------------------------------------------------------------------------- backward chain
template <int NT>
__device__ __forceinline__ void mask_by(f32x16 (&g)[NT], const float* act_row, int hh) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 m = *reinterpret_cast<const float4*>(act_row + 32 * t + 8 * q + 4 * hh);
      g[t][4 * q + 0] = m.x > 0.f ? g[t][4 * q + 0] : 0.f;
      g[t][4 * q + 1] = m.y > 0.f ? g[t][4 * q + 1] : 0.f;
      g[t][4 * q + 2] = m.z > 0.f ? g[t][4 * q + 2] : 0.f;
      g[t][4 * q + 3] = m.w > 0.f ? g[t][4 * q + 3] : 0.f;
    }
}

template <int NT, int GIN, int FIRST>
__global__ __launch_bounds__(256) void k_chain_bwd(ChainBwdArgs a) {
  constexpr int D = NT * 32;
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int64_t row = int64_t(blockIdx.x) * kTileRows + wave * 32 + (lane & 31);
  const bool live = row < a.R;

  f32x16 g[NT], acc[NT];
  zero_tile<NT>(g);
  if (live) {
    if (GIN == G_SMALL) {  // g = (dy . W_out) masked by the last hidden activation
      for (int c = 0; c < a.C; ++c) axpy_features<NT>(g, a.wout + c * D, a.dy[row * a.C + c], hh);
      mask_by<NT>(g, a.mask_in + row * D, hh);
    } else {
      const float* dyrow;
      if (GIN == G_EDGE_LN) {  // autograd of scatter_sum: gather the node gradient by target
        const int b = int(row / a.E), q = int(row - int64_t(b) * a.E);
        dyrow = a.dy + (int64_t(b) * a.N + a.dst[q]) * D;
      } else {
        dyrow = a.dy + row * D;
      }
      load_rows<NT>(g, dyrow, hh);
      load_rows<NT>(acc, a.yln + row * D, hh);  // acc = normalised output y
      // LayerNorm backward (no affine): dz = rstd * (dy - mean(dy) - y * mean(dy * y))
      const float m1 = row_sum<NT>(g) * (1.f / D);
      float s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s2 = fmaf(g[t][r], acc[t][r], s2);
      s2 += __shfl_xor(s2, 32, 64);
      const float m2 = s2 * (1.f / D), rs = a.rstd[row];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = rs * (g[t][r] - m1 - acc[t][r] * m2);
    }
    if (a.gstore[0]) store_rows<NT, false>(g, a.gstore[0] + row * D, hh);
  }

  for (int k = 0; k < a.nstage; ++k) {
    zero_tile<NT>(acc);
    gemm_stage<NT>(acc, g, a.wpt[k], lds, tid, lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) g[t] = acc[t];
    if (live) {
      if (a.mask[k]) mask_by<NT>(g, a.mask[k] + row * D, hh);
      if (a.gstore[k + 1]) store_rows<NT, false>(g, a.gstore[k + 1] + row * D, hh);
    }
  }

  if (FIRST != F_NONE) {
    zero_tile<NT>(acc);
    gemm_stage<NT>(acc, g, a.wh0, lds, tid, lane);
    if (live) {
      if (a.dres) {
        f32x16 r[NT];
        load_rows<NT>(r, a.dres + row * D, hh);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] += r[t];
      }
      store_rows<NT, false>(acc, a.dx + row * D, hh);
    }
    if (FIRST == F_HEADS2) {
      zero_tile<NT>(acc);
      gemm_stage<NT>(acc, g, a.wh1, lds, tid, lane);
      if (live) store_rows<NT, false>(acc, a.dx2 + row * D, hh);
    }
  }
}

template <int NT, int IN, int OUT>
int launch_fwd_t(const ChainFwdArgs& a, hipStream_t s) {
  const size_t lds = size_t(2) * NT * 256 * sizeof(float4);
  hipLaunchKernelGGL((k_chain_fwd<NT, IN, OUT>), dim3((unsigned)ceil_div(a.R, kTileRows)), dim3(256), lds, s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
// Only the combinations the path uses are instantiated (each is a large unrolled kernel).
template <int NT>
int launch_fwd_n(int in_mode, int out_mode, const ChainFwdArgs& a, hipStream_t s) {
#define BSMS_FWD(I, O) \
  if (in_mode == I && out_mode == O) return launch_fwd_t<NT, I, O>(a, s)
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

template <int NT, int GIN, int FIRST>
int launch_bwd_t(const ChainBwdArgs& a, hipStream_t s) {
  const size_t lds = size_t(2) * NT * 256 * sizeof(float4);
  hipLaunchKernelGGL((k_chain_bwd<NT, GIN, FIRST>), dim3((unsigned)ceil_div(a.R, kTileRows)), dim3(256), lds, s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
template <int NT>
int launch_bwd_n(int gin, int first, const ChainBwdArgs& a, hipStream_t s) {
#define BSMS_BWD(G, F) \
  if (gin == G && first == F) return launch_bwd_t<NT, G, F>(a, s)
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
    case 32: return launch_fwd_n<1>(in_mode, out_mode, a, s);
    case 64: return launch_fwd_n<2>(in_mode, out_mode, a, s);
    case 128: return launch_fwd_n<4>(in_mode, out_mode, a, s);
    case 256: return launch_fwd_n<8>(in_mode, out_mode, a, s);
  }
  BSMS_FAIL(BSMS_E_UNSUPPORTED, "latent width D=%d not supported (32, 64, 128, 256)", D);
}

int launch_chain_bwd(int D, int gin, int first, const ChainBwdArgs& a, hipStream_t s) {
  if (a.R == 0) return BSMS_OK;
  switch (D) {
    case 32: return launch_bwd_n<1>(gin, first, a, s);
    case 64: return launch_bwd_n<2>(gin, first, a, s);
    case 128: return launch_bwd_n<4>(gin, first, a, s);
    case 256: return launch_bwd_n<8>(gin, first, a, s);
  }
  BSMS_FAIL(BSMS_E_UNSUPPORTED, "latent width D=%d not supported (32, 64, 128, 256)", D);
}

}  // namespace bsms
