// Chain kernels (see chain.h for the register-layout idea) + weight prepack.
#include "chain.h"

// No implicit contraction in this file: hipcc defaults to -ffp-contract=fast and decides PER INSTANTIATION whether a
// multiply feeding an add becomes one fma -- the LayerNorm backward `g - m1 - y * m2` came out fused in some kernel
// variants and not in others, so the input gradient of a row depended (in the last bit) on the launch shape that
// happened to process it (found by tests/test_hip_parity.py::test_feature_split_kernels_equal_the_ring_kernels).
// Every fused multiply-add of the arithmetic is written as fmaf() explicitly; with this pragma nothing else is fused and
// all variants of a kernel (ring / single-round / pipelined edge / feature-split) agree bit for bit.
#pragma clang fp contract(off)

using namespace bsms;

#include "chain_dev.h"

namespace {

// pass 1 (one 1024-thread block per pack): 2^-k_w from the largest |M| of the matrix and of its mate -> header float
// kScaleSlot of chunk 0, where pass 2 and the chain kernels read it
__global__ __launch_bounds__(1024) void k_pack_scale(PackTable tab) {
  if (tab.zero)   // clear the block's bound slots (chain.h: kBoundWidth), spread over the launch
    for (int o = blockIdx.x * 1024 + threadIdx.x; o < kBoundSlots * kBoundWidth / 4; o += gridDim.x * 1024)
      reinterpret_cast<float4*>(tab.zero)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  const PackDesc d = tab.d[blockIdx.x];
  if (d.kind == PACK_TRANSPOSE || d.kind == PACK_ROWS_BF16 || d.bf16) return;
  __shared__ float red[16];
  float m = 0.f;
  auto scan = [&](const PackDesc& e) {   // coalesced along the rows of W whatever the logical orientation
    const int rows = (e.kind == PACK_FRAG_T) ? e.K : e.N, cols = (e.kind == PACK_FRAG_T) ? e.N : e.K;
    const int tr = threadIdx.x / cols, tc = threadIdx.x % cols, step = 1024 / cols;   // cols divides 1024 (32 .. 256)
    for (int r = tr; r < rows; r += step) m = fmaxf(m, fabsf(e.W[int64_t(e.row0 + r) * e.ld + e.col0 + tc]));
  };
  scan(d);
  if (d.mate) scan(tab.d[d.mate - 1]);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    int Ew = int(__float_as_uint(m) >> 23);
    Ew = Ew < 13 ? 13 : (Ew > 254 ? 254 : Ew);            // 2^(139 - Ew) and its inverse are normal floats
    d.dst[kScaleSlot] = __uint_as_float(unsigned(Ew - 12) << 23);    // 2^(Ew - 139) = 2^-k_w
  }
}

__global__ __launch_bounds__(256) void k_prepack(PackTable tab) {
  const PackDesc d = tab.d[blockIdx.y];
  if (d.kind == PACK_ROWS_BF16) { pack_rows_bf16(d, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256); return; }
  if (d.kind == PACK_TRANSPOSE) {
    const int total = d.N * d.K;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
      const int k = o / d.N, n = o % d.N;
      d.dst[o] = d.W[int64_t(d.row0 + n) * d.ld + d.col0 + k];
    }
    return;
  }
  float sw = 1.f;
  if (!d.bf16) sw = __uint_as_float(unsigned(254 - int(__float_as_uint(d.dst[kScaleSlot]) >> 23)) << 23);   // 2^k_w = 1 / header value
  const int nb = d.N >> 4, planes = d.bf16 ? 1 : kPL, chf = kChunkHdrFloats + nb * 256 * planes, nch = d.K >> 5;
  const int total = nch * chf;
  unsigned* dst = reinterpret_cast<unsigned*>(d.dst);
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    const int c = o / chf, w = o % chf;
    if (w < kChunkHdrFloats) {
      float v = 0.f;
      if (d.bf16) {
        if (c == 0 && d.bias && w < d.N) v = d.bias[w];
      } else {
        if (c == nch - 1 && d.bias && w < d.N) v = d.bias[w];
        if (c == 0 && w == kScaleSlot) continue;          // written by k_pack_scale (nch == 1 means N = 32: no clash with the bias)
      }
      d.dst[o] = v;
      continue;
    }
    const int q = w - kChunkHdrFloats;
    const int v = q & 3, lane = (q >> 2) & 63, tp = q >> 8, plane = tp % planes, t = tp / planes;
    const int n = 16 * t + (lane & 15), k = 16 * (2 * c + ((2 * v) >> 2)) + 4 * (lane >> 4) + ((2 * v) & 3);   // slots 2v, 2v + 1
    const float x0 = pack_elem(d, n, k), x1 = pack_elem(d, n, k + 1);
    if (d.bf16) {   // bf16 precision: the weight IS its bf16 rounding (nearest even)
      dst[o] = pk_bf16(x0, x1);
    } else {
      unsigned h, l;
      split_h2(x0, x1, sw, h, l);
      dst[o] = plane == 0 ? h : l;
    }
  }
}

// k_pack_scale + k_prepack in ONE launch (round 4: a training step packs 13 weight sets, three of them in front of kernels on
// the caller's stream): every workgroup of a pack repeats the scan for the matrix maximum (64-128 KB from L2) and then writes
// its share of the pack; same scale, same pieces, same bytes as the two-pass form.
__global__ __launch_bounds__(1024) void k_prepack_fused(PackTable tab) {
  const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (tab.zero)   // clear the block's bound slots (chain.h: kBoundWidth), spread over the launch
    for (int o = wg * 1024 + threadIdx.x; o < kBoundSlots * kBoundWidth / 4; o += nwg * 1024)
      reinterpret_cast<float4*>(tab.zero)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  const PackDesc d = tab.d[blockIdx.y];
  if (d.kind == PACK_ROWS_BF16) { pack_rows_bf16(d, blockIdx.x * 1024 + threadIdx.x, gridDim.x * 1024); return; }
  if (d.kind == PACK_TRANSPOSE) {
    const int total = d.N * d.K;
    for (int o = blockIdx.x * 1024 + threadIdx.x; o < total; o += gridDim.x * 1024) {
      const int k = o / d.N, n = o % d.N;
      d.dst[o] = d.W[int64_t(d.row0 + n) * d.ld + d.col0 + k];
    }
    return;
  }
  float sw = 1.f;
  unsigned hdr_scale = 0;
  if (!d.bf16) {
    __shared__ float red[16];
    float m = 0.f;
    auto scan = [&](const PackDesc& e) {   // coalesced along the rows of W whatever the logical orientation
      const int rows = (e.kind == PACK_FRAG_T) ? e.K : e.N, cols = (e.kind == PACK_FRAG_T) ? e.N : e.K;
      const int tr = threadIdx.x / cols, tc = threadIdx.x % cols, step = 1024 / cols;   // cols divides 1024 (32 .. 256)
      const float* base = e.W + int64_t(e.row0) * e.ld + e.col0 + tc;
      int r = tr;
      for (; r + 7 * step < rows; r += 8 * step) {   // eight independent loads in flight (one at a time: 16-64 dependent L2 round trips)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[int64_t(r + u * step) * e.ld];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(v[u]));
      }
      for (; r < rows; r += step) m = fmaxf(m, fabsf(base[int64_t(r) * e.ld]));
    };
    scan(d);
    if (d.mate) scan(tab.d[d.mate - 1]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    int Ew = int(__float_as_uint(m) >> 23);
    Ew = Ew < 13 ? 13 : (Ew > 254 ? 254 : Ew);            // 2^(139 - Ew) and its inverse are normal floats
    hdr_scale = unsigned(Ew - 12) << 23;                   // 2^(Ew - 139) = 2^-k_w  (header float kScaleSlot of chunk 0)
    sw = __uint_as_float(unsigned(254 - (Ew - 12)) << 23); // 2^k_w
  }
  const int nb = d.N >> 4, planes = d.bf16 ? 1 : kPL, chf = kChunkHdrFloats + nb * 256 * planes, nch = d.K >> 5;
  const int total = nch * chf;
  unsigned* dst = reinterpret_cast<unsigned*>(d.dst);
  // four pack dwords per round: their eight weight loads are in flight together
  const int stride = gridDim.x * 1024;
  for (int o0 = blockIdx.x * 1024 + threadIdx.x; o0 < total; o0 += 4 * stride) {
    float x0[4], x1[4];
    bool body_[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + u * stride, w = o % chf;
      body_[u] = o < total && w >= kChunkHdrFloats;
      x0[u] = x1[u] = 0.f;
      if (body_[u]) {
        const int c = o / chf, q = w - kChunkHdrFloats;
        const int v = q & 3, lane = (q >> 2) & 63, tp = q >> 8, t = tp / planes;
        const int n = 16 * t + (lane & 15), k = 16 * (2 * c + ((2 * v) >> 2)) + 4 * (lane >> 4) + ((2 * v) & 3);   // slots 2v, 2v + 1
        x0[u] = pack_elem(d, n, k);
        x1[u] = pack_elem(d, n, k + 1);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + u * stride;
      if (o >= total) break;
      const int c = o / chf, w = o % chf;
      if (!body_[u]) {   // chunk header
        float v = 0.f;
        if (d.bf16) {
          if (c == 0 && d.bias && w < d.N) v = d.bias[w];
        } else {
          if (c == nch - 1 && d.bias && w < d.N) v = d.bias[w];
          if (c == 0 && w == kScaleSlot) { dst[o] = hdr_scale; continue; }   // (nch == 1 means N = 32: no clash with the bias)
        }
        d.dst[o] = v;
        continue;
      }
      const int plane = ((w - kChunkHdrFloats) >> 8) % planes;
      if (d.bf16) {
        dst[o] = pk_bf16(x0[u], x1[u]);
      } else {
        unsigned h, l;
        split_h2(x0[u], x1[u], sw, h, l);
        dst[o] = plane == 0 ? h : l;
      }
    }
  }
}

// -------------------------------------------------------------------------------- forward chain
// TIMING (experiments, profiles/tile_timeline.py): phase stamps of wave 0; a separate instantiation so that the
// production kernel carries none of it.
template <int NB, int IN, int OUT, bool TIMING = false, bool BF = false, bool LONE = false>
__global__ __launch_bounds__(kChainMaxThreads) __attribute__((amdgpu_waves_per_eu(NB <= 8 ? (LONE ? 2 : BSMS_CHAIN_WPE) : 2))) void k_chain_fwd(ChainFwdArgs a) {
  constexpr int D = NB * 16;
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4;
  const int cw = int(blockDim.x >> 6) - a.nload;   // compute waves of this launch (4..7, chosen by the launcher); the last wave(s) load
  if (wave >= cw) {  // loader wave (uniform branch)
    loader_dispatch<NB, BF ? 1 : kPL>(a.nload, wave - cw, a.wseq, a.nseq, lds, lane, a.ntiles, a.nring, IN == IN_EDGE ? a.w0t : nullptr);
    return;
  }
  // IN_EDGE: the fiber weights are read from the LDS side table (read from HBM/L2 they cost one dependent round
  // trip per 16 bytes: 24 of them per tile, the largest part of the input stage)
  const float* w0t = a.w0t;
  if (IN == IN_EDGE) {
    lds_barrier();
    w0t = reinterpret_cast<const float*>(lds);
  }
  Slot slot{0, a.nring};  // ring slot of the next chunk; runs on across this workgroup's tiles exactly like the loader's
  float4* const ring = lds + Ring<NB>::PRE4;
  unsigned* brow = bound_row<NB>(lds, wave, lane, any_slot(a.amax));   // this wave's running magnitude bounds
  // Persistent workgroups: the grid is sized to what the chip holds at once and strides over the tiles, so a CU
  // never waits for the dispatcher to refill a slot (measured: 20-35 % of slot time was empty with one
  // workgroup per tile) and the loader is already fetching the next tile's first chunk during this epilogue.
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
  const int64_t row = int64_t(tile) * (16 * cw) + wave * 16 + (lane & 15);
  const bool live = row < a.R;
  const int64_t rowc = live ? row : 0;   // what a lane past the end reads (its results are never stored)
  const int64_t roff = live ? row * D : -1;  // row offset for stores; negative = no store
  int stamp_i = 0;
  unsigned long long waited = 0;
  auto stamp = [&]() {  // experiments: wave 0 / lane 0 records the shader clock at phase boundaries
    if (TIMING && a.timing && tid == 0 && stamp_i < 16) a.timing[int64_t(tile) * 16 + stamp_i++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  if (TIMING && a.timing && tid == 0) {
    a.timing[int64_t(tile) * 16 + 14] = __builtin_amdgcn_s_memrealtime();
    a.timing[int64_t(tile) * 16 + 13] = (uint64_t(__builtin_amdgcn_s_getreg(63508)) << 32) |  // XCC_ID
                                        uint32_t(__builtin_amdgcn_s_getreg(63492));             // HW_ID
  }

  f32x4 act[NB], acc[NB];
  // single-round launches (256-register budget, nothing to overlap a memory round trip with): the second source of a Linear
  // over [x, x2] is requested together with the first and stays in registers (the multi-round variants re-read it twice)
  constexpr bool KEEP2 = LONE && IN == IN_ROWS2 && NB == 8;
  f32x4 x2t[KEEP2 ? NB : 1];

  // ---- input stage
  if (IN == IN_ROWS || IN == IN_ROWS2) {
    load_rows<NB>(act, a.x + rowc * D, lg);
    if constexpr (KEEP2) load_rows<NB>(x2t, a.x2 + rowc * D, lg);
  } else if (IN == IN_SMALL) {
    load_features<NB>(act, a.bias_in, lg);
    for (int k = 0; k < a.K0; ++k) axpy_features<NB>(act, w0t + k * D, a.x[rowc * a.K0 + k], lg);
    relu_into<NB>(act, act);
  } else {  // IN_EDGE: relu(Ps[src] + Pd[dst] + Wf . [pos_i - pos_j, |pos_i - pos_j|])   (ops/basic.py:70-92)
    {
      const int b = int(rowc / a.E), q = int(rowc - int64_t(b) * a.E);
      const int i = a.src[q], j = a.dst[q];
      load_rows<NB>(act, a.Ps + (int64_t(b) * a.N + i) * D, lg);
      load_rows<NB>(acc, a.Pd + (int64_t(b) * a.N + j) * D, lg);
      const float* pb = a.pos + b * a.pos_bstride;
      float pi[7], pj[7];  // check_gmp: p <= 7
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        const int cc = c < a.p ? c : 0;   // uniform clamp: the loads stay unconditional
        pi[c] = pb[int64_t(i) * a.p + cc];
        pj[c] = pb[int64_t(j) * a.p + cc];
      }
      // all gathers of the tile are in flight before the first use
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NB; ++t) act[t] += acc[t];
      float n2 = 0.f;
#pragma unroll
      for (int c = 0; c < 7; ++c)
        if (c < a.p) {
          const float rel = pi[c] - pj[c];
          n2 = fmaf(rel, rel, n2);
          axpy_features<NB>(act, w0t + c * D, rel, lg);
        }
      const float nrm = sqrtf(n2);
      axpy_features<NB>(act, w0t + a.p * D, nrm, lg);
      relu_into<NB>(act, act);
      if (a.fiber_out && live && lg == 0) {   // one lane per row keeps the fiber for the backward (16 or 32 bytes per edge)
        float f[8];
#pragma unroll
        for (int c = 0; c < 7; ++c) f[c] = c < a.p ? pi[c] - pj[c] : (c == a.p ? nrm : 0.f);
        f[7] = a.p == 7 ? nrm : 0.f;
        const int ld = fiber_ld(a.p);
        float4* dst = reinterpret_cast<float4*>(a.fiber_out + row * ld);
        dst[0] = make_float4(f[0], f[1], f[2], f[3]);
        if (ld == 8) dst[1] = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
  }

  // ---- MFMA stages.  The activation entering a stage is stored to HBM from inside that stage (mfma_stage).
  stamp();            // input stage done
  stamp();
  float* pending = (IN == IN_SMALL || IN == IN_EDGE) ? a.store_in : nullptr;   // uniform
  if (a.nstage == 0) {
    store_rows<NB, false>(act, pending, roff, lg);
    store_mask_bits<NB>(act, pending, a.R, roff, lg);
  }
  if (OUT == OUT_PLAIN2) {  // two Linears of the SAME rows (the edge MLP's two node projections): one launch, one read of x
    const float m = row_amax<NB>(act);
    note_amax(brow, 0, m, lane);
    const RowScale rs = scale_of(m);
    mfma_stage<NB, true, 2, LONE>(acc, act, rs, ring, slot, lane);
    store_rows<NB, false>(acc, a.y, roff, lg);
    mfma_stage<NB, true, 2, LONE>(acc, act, rs, ring, slot, lane);
    store_rows<NB, false>(acc, a.y2, roff, lg);
    continue;
  }
  for (int l = 0; l < a.nstage; ++l) {
    if constexpr (BF) {
      if (IN == IN_ROWS2 && l == 0) {   // BSMS_BF16_NODES: Linear over [x, x2], both rounded to bf16 as they enter; the bias rides in the FIRST pack
        float m = row_amax<NB>(act);
        load_rows<NB>(acc, a.x2 + rowc * D, lg);
        m = fmaxf(m, row_amax<NB>(acc));
        note_amax(brow, 0, m, lane);      // bound of the fp32 rows [x, x2]: operands of the first Linear's fp32 weight-gradient job
        mfma_stage_bf<NB>(acc, act, ring, slot, lane, true, nullptr, roff, 0);
        load_rows<NB>(act, a.x2 + rowc * D, lg);
        mfma_stage_bf<NB>(acc, act, ring, slot, lane, false, nullptr, roff, 0);
      } else {
        mfma_stage_bf<NB>(acc, act, ring, slot, lane, true, pending, roff, (a.store_mode & 4) ? 0 : a.R);   // acc = bias + W act
      }
    } else if (IN == IN_ROWS2 && l == 0) {
      // Linear over the concatenation [x, x2]: ONE row scale (the larger of the two rows' maxima; the second source is
      // read once more for it -- node-level rows, L2-resident) and one weight scale (PackDesc::mate), so the second half
      // continues the raw sums of the first; the bias rides in the second pack
      float m = row_amax<NB>(act);
      stamp();          // (timing builds) x has arrived
      if constexpr (KEEP2) {
        m = fmaxf(m, row_amax<NB>(x2t));
      } else {
        load_rows<NB>(acc, a.x2 + rowc * D, lg);
        m = fmaxf(m, row_amax<NB>(acc));
      }
      stamp();          // x2 has arrived
      note_amax(brow, 0, m, lane);
      const RowScale rs = scale_of(m);
      mfma_stage<NB, true, 0, LONE>(acc, act, rs, ring, slot, lane);
      stamp();          // first half of stage 0
      if constexpr (KEEP2) {
        mfma_stage<NB, false, 2, LONE>(acc, x2t, rs, ring, slot, lane);
      } else {
        load_rows<NB>(act, a.x2 + rowc * D, lg);
        mfma_stage<NB, false, 2, LONE>(acc, act, rs, ring, slot, lane);
      }
    } else {
      const float m = row_amax<NB>(act);
      note_amax(brow, l, m, lane);
      mfma_stage<NB, true, 2, LONE, TIMING>(acc, act, scale_of(m), ring, slot, lane, pending, roff, a.store_mode & 3,
                                      (a.store_mode & 4) ? 0 : a.R, &waited, row, (a.store_mode & 8) ? 0 : a.R);  // acc = bias + W act
    }
    stamp();          // stage l done
    pending = nullptr;
    const bool last = (l == a.nstage - 1);
    if (!last || OUT == OUT_SMALL) {
      relu_into<NB>(act, acc);
      if (!last) pending = a.store[l];
      else store_rows<NB, false>(act, a.store[l], roff, lg);
    }
  }
  stamp();
  if (TIMING && a.timing && tid == 0) {
    a.timing[int64_t(tile) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
    a.timing[int64_t(tile) * 16 + 11] = waited;
  }
  if (!live) continue;

  // ---- output
  if (OUT == OUT_LN) {  // LayerNorm(elementwise_affine=False), eps 1e-5  (ops/basic.py:18)
    // single-round launches: the residual rows are requested BEFORE the LayerNorm arithmetic (`act` and `x2t` are dead by
    // now) instead of one exposed round trip each after it; the additions below are the same, in the same order
    constexpr bool EARLY = LONE && !BF && NB == 8;
    if constexpr (EARLY) {
      if (a.resid) load_rows<NB>(act, a.resid + row * D, lg);
      if constexpr (KEEP2) { if (a.resid2) load_rows<NB>(x2t, a.resid2 + row * D, lg); }
    }
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
    if constexpr (BF && IN == IN_EDGE) {   // edge messages of the bf16 precision: stored (and consumed by the aggregation) as bf16
      store_rows_bf16<NB>(acc, a.y, roff, lg);
      if (a.rstd && lg == 0) a.rstd[row] = rstd;
      continue;
    }
    store_rows<NB, false>(acc, a.yln, roff, lg);
    if (a.rstd && lg == 0) a.rstd[row] = rstd;
    if (a.resid) {
      if constexpr (!EARLY) load_rows<NB>(act, a.resid + row * D, lg);
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] += act[t];
    }
    if (a.resid2) {   // (LN + x) + skip: the same two additions, in the same order, as GMP's `+ x` then BSGMP's `h + down_outs`
      if constexpr (EARLY && KEEP2) {
#pragma unroll
        for (int t = 0; t < NB; ++t) acc[t] += x2t[t];
      } else {
        load_rows<NB>(act, a.resid2 + row * D, lg);
#pragma unroll
        for (int t = 0; t < NB; ++t) acc[t] += act[t];
      }
    }
    store_rows<NB, false>(acc, a.y, roff, lg, a.out_mode);
    if (TIMING && a.timing && tid == 0) {
      __builtin_amdgcn_s_waitcnt(0);  // experiments: all of this wave's stores acknowledged
      a.timing[int64_t(tile) * 16 + 12] = __builtin_amdgcn_s_memrealtime();
    }
  } else if (OUT == OUT_PLAIN) {
    if (a.accumulate) store_rows<NB, true>(acc, a.y, roff, lg);
    else store_rows<NB, false>(acc, a.y, roff, lg);
  } else {  // OUT_SMALL: the narrow last Linear (decoder, models/model.py:22) on the VALU
    for (int c = 0; c < a.C; ++c) {
      const float v = dot_features<NB>(act, a.wout + c * D, lg);
      if (lg == 0) a.y[row * a.C + c] = v + a.bout[c];
    }
  }
  }  // tile loop
  flush_bounds(a.amax, kMaxStages + 1, brow, wave, lane);
}

// ------------------------------------------------------------------- small launches: feature-split forward chain ----
// A launch of a few thousand rows at most (every node-level MLP of a batch-1 step / rollout, the coarse levels of any
// step) is LATENCY, not throughput: in k_chain_fwd one wave per SIMD owns 16 rows x all 128 features and works through
// ~750-900 cycles per 32-feature chunk (profiles/census/stage_lone.hip: 24 MFMAs 427, its 16 ds_read_b128 of shared
// weight fragments 513 at the rate a lone wave gets, the two-way split 187, barrier-serialised), 4.5k cycles per Linear,
// 12 us for the node MLP whatever the row count -- and no loader / ring variation moves it (profiles/lone_timeline.py).
// Here the FEATURES of a 16-row tile are split over the four waves of a 256-thread workgroup: wave w owns output feature
// blocks 2w, 2w+1 of every Linear = a quarter of the MFMAs, of the split, of the epilogue arithmetic.  Its accumulator
// layout is exactly K block w of the next Linear's B operand (chain.h), so what the waves exchange through LDS per
// Linear is 2 KB of fp16 pieces each plus the row maximum -- two LDS barriers.  Each wave needs only ITS quarter of every
// weight chunk, and all four together read each weight byte once per tile: the fragments come straight from L2 into
// registers (one 1 KB global_load_dwordx4 per fragment), a whole Linear ahead -- no LDS ring, no loader wave, no chunk
// barriers.  Per-element arithmetic and its order are those of k_chain_fwd (same split, same three products per
// accumulator in the same order, same row scale, LayerNorm on the full row gathered through LDS): BIT-IDENTICAL results
// (tests/test_hip_parity.py::test_feature_split_kernels_equal_the_ring_kernels).  Weight traffic per row is 4-14x that
// of the persistent ring kernels, so the launcher takes this path only below kFsMaxRows rows.
// Same-box sweeps of the threshold (profiles/r04 fs_rows): airfoil B = 8 step 187.1 (never) / 188.0 (5000) / 187.3 (12288) /
// 185.2 (24000) steps/s; B = 1 rollout 1613 (never) / 1750 (3000) / 1783 (12288): the level-0 launches of a batch-1 step
// (5233 rows) gain, the 10 104 rows of level 2 at batch 8 do not.
constexpr int kFsMaxRows = 6144;    // 384 tiles of 16 rows: one and a half per CU
// The backward form re-reads the full dy / y rows in every wave and runs at 256 VGPRs: per level of the batch-1 / batch-8
// traces it wins up to ~2600 rows (14.3-16.6 us against ~19.6 for the single-round ring kernel) and loses at 4728-5233 rows
// (24.7-29.4 against 20-23 us).
constexpr int kFsMaxRowsBwd = 3072;

struct FsPack {           // one weight pack of the chain as wave `w` sees it
  float4 f[16];           // [chunk c][block i = 0, 1][plane h, l]  -> f[c * 4 + i * 2 + plane]
  float4 bias[2];         // bias of the own feature blocks (header of the last chunk), this lane's features
  float scale;            // 2^-k_w (header float kScaleSlot of chunk 0)
};
__device__ __forceinline__ void fs_request(FsPack& p, const float4* wp, int w, int lane) {
  using R = Ring<8>;
  const float4* body = wp + kChunkHdrFloats / 4 + lane;
#pragma unroll
  for (int c = 0; c < R::NCH; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) p.f[c * 4 + i * 2 + pl] = body[size_t(c) * R::CH4 + ((2 * w + i) * 2 + pl) * 64];
  const float* hdr_last = reinterpret_cast<const float*>(wp + size_t(R::NCH - 1) * R::CH4);
#pragma unroll
  for (int i = 0; i < 2; ++i) p.bias[i] = *reinterpret_cast<const float4*>(hdr_last + 16 * (2 * w + i) + 4 * (lane >> 4));
  p.scale = reinterpret_cast<const float*>(wp)[kScaleSlot];
}

// the four K blocks of the activation entering a Linear, as B operands
struct FsPieces { u32x4 h[4], l[4]; };

// One Linear on the own feature blocks.  ZERO / FIN as in mfma_stage.  `next` / `wnext` (nullable, uniform): the NEXT pack of
// the chain is requested chunk by chunk between this pack's MFMAs -- a lone wave issues a 1 KB global load per ~75 cycles
// (profiles/fs_timeline.py: 1.4k cycles for the 19 loads of a pack, against 430 for its 24 MFMAs), so the matrix
// instructions execute under the load issue instead of after it.
template <bool ZERO, int FIN>
__device__ __forceinline__ void fs_stage(f32x4 (&acc)[2], const FsPack& p, const FsPieces& x, int E, int lane,
                                         FsPack* next = nullptr, const float4* wnext = nullptr, int w = 0) {
  using R = Ring<8>;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const float4* nbody = wnext ? wnext + kChunkHdrFloats / 4 + lane : nullptr;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = mma(p.f[c * 4 + i * 2], x.l[c], (ZERO && c == 0) ? zero : acc[i]);
    if (wnext) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) next->f[c * 4 + i * 2 + pl] = nbody[size_t(c) * R::CH4 + ((2 * w + i) * 2 + pl) * 64];
      if (c == 3) {
        const float* hdr_last = reinterpret_cast<const float*>(wnext + size_t(R::NCH - 1) * R::CH4);
#pragma unroll
        for (int i = 0; i < 2; ++i) next->bias[i] = *reinterpret_cast<const float4*>(hdr_last + 16 * (2 * w + i) + 4 * (lane >> 4));
        next->scale = reinterpret_cast<const float*>(wnext)[kScaleSlot];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = mma(p.f[c * 4 + i * 2], x.h[c], acc[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = mma(p.f[c * 4 + i * 2 + 1], x.h[c], acc[i]);
  }
  if (FIN != 0) {   // finish_stage on the own blocks: same fast / slow path decision (wave-uniform over the same 16 rows)
    const int fw = int(__float_as_uint(p.scale) >> 23);
    const int f = E + fw - 139;
    if (__builtin_amdgcn_ballot_w64(unsigned(f - 1) >= 254u) == 0) {
      const float inv = __uint_as_float(unsigned(f) << 23);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (FIN == 2) acc[i] = f32x4{fmaf(acc[i][0], inv, p.bias[i].x), fmaf(acc[i][1], inv, p.bias[i].y), fmaf(acc[i][2], inv, p.bias[i].z), fmaf(acc[i][3], inv, p.bias[i].w)};
        else acc[i] *= inv;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 b = FIN == 2 ? p.bias[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[i] = f32x4{ldexpf(acc[i][0], f - 127) + b.x, ldexpf(acc[i][1], f - 127) + b.y, ldexpf(acc[i][2], f - 127) + b.z, ldexpf(acc[i][3], f - 127) + b.w};
      }
    }
  }
}

struct FsLds {
  float pmax[16][16];       // [row][4 w + g]: largest |value| of the row among the features held by (wave w, lane group g)
  u32x4 piece[4][2][64];    // [K block][plane][lane]
  float zrow[16][132];      // full rows for the LayerNorm / the narrow output layer (pitch 132: 16-byte aligned, spread over banks)
};

__device__ __forceinline__ float fs_amax2(const f32x4 (&v)[2]) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    m = fmaxf(fmaxf(m, fabsf(v[i][0])), fabsf(v[i][1]));
    m = fmaxf(fmaxf(m, fabsf(v[i][2])), fabsf(v[i][3]));
  }
  return m;
}
// row maximum over all 128 features: every (wave, lane group) publishes its part, everybody reads the 16 parts of its row
__device__ __forceinline__ float fs_row_max(FsLds& L, float mloc, int w, int lane) {
  L.pmax[lane & 15][4 * w + (lane >> 4)] = mloc;
  lds_barrier();
  const float4* p = reinterpret_cast<const float4*>(L.pmax[lane & 15]);
  const float4 a = p[0], b = p[1], c = p[2], d = p[3];
  return fmaxf(fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))),
               fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w))));
}
// own K block -> fp16 pieces, published; the other three are read back
__device__ __forceinline__ void fs_publish(FsLds& L, FsPieces& x, const f32x4 (&own)[2], float s, int w, int lane) {
  u32x4 h, l;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    unsigned hh, ll;
    split_h2(own[v >> 1][2 * (v & 1)], own[v >> 1][2 * (v & 1) + 1], s, hh, ll);
    h[v] = hh;
    l[v] = ll;
  }
  L.piece[w][0][lane] = h;
  L.piece[w][1][lane] = l;
  lds_barrier();
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    x.h[kb] = L.piece[kb][0][lane];
    x.l[kb] = L.piece[kb][1][lane];
  }
}
// largest |value| over the tile's rows -> this workgroup's entry of a bound slot (chain.h; wave 0 only: m is per row)
__device__ __forceinline__ void fs_note(float* slot, float m, int w, int lane) {
  if (!slot || w != 0) return;   // uniform
  int v = __float_as_int(m);
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));
  if (lane == 15) slot[int(blockIdx.x) * 8] = __int_as_float(v);
}
// saved activation (values + ReLU sign bits) of the own feature blocks, act_floats layout (chain.h)
__device__ __forceinline__ void fs_save(float* base, const f32x4 (&own)[2], int64_t R, int64_t row, bool live, int w, int lane, bool bits) {
  if (!base || !live) return;
  constexpr int D = 128;
  const int lg = lane >> 4;
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    __builtin_nontemporal_store(own[i], reinterpret_cast<f32x4*>(base + row * D + 16 * (2 * w + i) + 4 * lg));
#pragma unroll
    for (int r = 0; r < 4; ++r) m |= (__float_as_uint(own[i][r]) != 0u ? 1u : 0u) << (4 * i + r);   // post-ReLU value: positive iff non-zero bits
  }
  if (bits) reinterpret_cast<unsigned char*>(base + pad_rows(R) * D)[(row * 4 + lg) * 4 + w] = (unsigned char)m;   // bits 8w .. 8w+7 of the word of (row, group)
}

template <int IN, int OUT>
__global__ __launch_bounds__(256) void k_fs_fwd(ChainFwdArgs a) {
  constexpr int NB = 8, D = 128;
  __shared__ FsLds L;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lg = lane >> 4;
  const int64_t row = int64_t(blockIdx.x) * 16 + (lane & 15);
  const bool live = row < a.R;
  const int64_t rowc = live ? row : 0;   // what a lane past the end reads (its results are never stored)
#ifdef BSMS_EXPERIMENTS
  int stamp_i = 0;
  auto stamp = [&]() { if (a.timing && tid == 0 && stamp_i < 16) a.timing[int64_t(blockIdx.x) * 16 + stamp_i++] = __builtin_amdgcn_s_memtime(); };
#else
  auto stamp = [] {};
#endif
  stamp();
  FsPack pa, pb;                         // packs alternate between the two register sets, one Linear ahead
  fs_request(pa, a.wseq[0], w, lane);
  f32x4 own[2], acc[2];
  FsPieces x;
  float m;
  // ---- input stage (own feature blocks 2w, 2w+1 = K block w of the first Linear)
  f32x4 own2[2];
  if (IN == IN_ROWS || IN == IN_ROWS2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) own[i] = *reinterpret_cast<const f32x4*>(a.x + rowc * D + 16 * (2 * w + i) + 4 * lg);
    if (IN == IN_ROWS2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) own2[i] = *reinterpret_cast<const f32x4*>(a.x2 + rowc * D + 16 * (2 * w + i) + 4 * lg);
    }
  } else {  // IN_SMALL: the narrow first layer on the VALU, relu(b0 + sum_k x[k] W0[:, k])
#pragma unroll
    for (int i = 0; i < 2; ++i) own[i] = *reinterpret_cast<const f32x4*>(a.bias_in + 16 * (2 * w + i) + 4 * lg);
    for (int k = 0; k < a.K0; ++k) {
      const float xv = a.x[rowc * a.K0 + k];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 wv = *reinterpret_cast<const float4*>(a.w0t + k * D + 16 * (2 * w + i) + 4 * lg);
        own[i][0] = fmaf(xv, wv.x, own[i][0]);
        own[i][1] = fmaf(xv, wv.y, own[i][1]);
        own[i][2] = fmaf(xv, wv.z, own[i][2]);
        own[i][3] = fmaf(xv, wv.w, own[i][3]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) own[i][r] = __int_as_float(max(__float_as_int(own[i][r]), 0));
    fs_save(a.store_in, own, a.R, row, live, w, lane, !(a.store_mode & 4));
    if (a.nstage == 0) return;
  }
  float mloc = fs_amax2(own);
  if (IN == IN_ROWS2) mloc = fmaxf(mloc, fs_amax2(own2));
  stamp();   // loads issued
  m = fs_row_max(L, mloc, w, lane);
  stamp();   // input rows arrived, row maximum exchanged
  fs_note(a.amax[0], m, w, lane);
  RowScale rs = scale_of(m);
  fs_publish(L, x, own, rs.s, w, lane);
  stamp();   // pieces exchanged

  // ---- Linears.  `q` walks the pack sequence (a.wseq: IN_ROWS2 has two packs for its first Linear, OUT_PLAIN2 one per head)
  auto run = [&](FsPack& cur, FsPack& nxt, int q, int l) -> bool {   // returns false when the chain is finished
    const float4* wn = q + 1 < a.nseq ? a.wseq[q + 1] : nullptr;     // the next pack is requested between this pack's MFMAs
    if (OUT == OUT_PLAIN2) {   // two Linears of the SAME rows: stage q -> y (q = 0) / y2 (q = 1)
      fs_stage<true, 2>(acc, cur, x, rs.E, lane, &nxt, wn, w);
      float* y = q == 0 ? a.y : a.y2;
      if (live)
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(y + row * D + 16 * (2 * w + i) + 4 * lg) = acc[i];
      return q + 1 < a.nseq;
    }
    if (IN == IN_ROWS2 && q == 0) {   // first half of the Linear over [x, x2]: raw sums, continued by the second pack
      fs_stage<true, 0>(acc, cur, x, rs.E, lane, &nxt, wn, w);
      stamp();
      lds_barrier();                  // everybody has read the pieces of x
      fs_publish(L, x, own2, rs.s, w, lane);
      stamp();
      return true;
    }
    if (IN == IN_ROWS2 && q == 1) fs_stage<false, 2>(acc, cur, x, rs.E, lane, &nxt, wn, w);
    else fs_stage<true, 2>(acc, cur, x, rs.E, lane, &nxt, wn, w);
    stamp();   // MFMAs of the pack issued (the wave has its weights)
    const bool last = l == a.nstage - 1;
    if (!last || OUT == OUT_SMALL) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) own[i][r] = __int_as_float(max(__float_as_int(acc[i][r]), 0));
    }
    if (last) return false;
    fs_save(a.store[l], own, a.R, row, live, w, lane, !(a.store_mode & 4));
    lds_barrier();                    // the pieces of the previous activation have been read by everybody
    m = fs_row_max(L, fs_amax2(own), w, lane);
    fs_note(a.amax[l + 1], m, w, lane);
    rs = scale_of(m);
    fs_publish(L, x, own, rs.s, w, lane);
    stamp();   // next activation exchanged
    return true;
  };
  {
    int q = 0, l = 0;
    for (;;) {
      if (!run(pa, pb, q, l)) break;
      if (!(IN == IN_ROWS2 && q == 0) && OUT != OUT_PLAIN2) ++l;
      ++q;
      if (!run(pb, pa, q, l)) break;
      if (!(IN == IN_ROWS2 && q == 0) && OUT != OUT_PLAIN2) ++l;
      ++q;
    }
  }
  if (OUT == OUT_PLAIN2) return;

  // ---- output
  if (OUT == OUT_PLAIN) {
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4* p = reinterpret_cast<f32x4*>(a.y + row * D + 16 * (2 * w + i) + 4 * lg);
      f32x4 v = acc[i];
      if (a.accumulate) v += *p;
      *p = v;
    }
    return;
  }
  // OUT_LN / OUT_SMALL work on FULL rows: gather them through LDS in the chain layout, then exactly the arithmetic of k_chain_fwd
  lds_barrier();
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&L.zrow[lane & 15][16 * (2 * w + i) + 4 * lg]) = (OUT == OUT_SMALL) ? own[i] : acc[i];
  if (OUT == OUT_SMALL && a.store[a.nstage - 1] && live) {   // last hidden activation (plain rows, no sign bits: the backward masks by value)
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(a.store[a.nstage - 1] + row * D + 16 * (2 * w + i) + 4 * lg) = own[i];
  }
  lds_barrier();
  f32x4 z[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) z[t] = *reinterpret_cast<const f32x4*>(&L.zrow[lane & 15][16 * t + 4 * lg]);
  if (!live) return;
  if (OUT == OUT_LN) {  // LayerNorm(elementwise_affine=False), eps 1e-5  (ops/basic.py:18): every wave normalises the row, stores its quarter
    const float mean = row_sum<NB>(z) * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z[t][r] -= mean;
        ss = fmaf(z[t][r], z[t][r], ss);
      }
    ss = group_sum(ss);
    const float rstd = 1.f / sqrtf(ss * (1.f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = 2 * w + i;
      f32x4 v = z[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rstd;
      const int64_t o = row * D + 16 * t + 4 * lg;
      if (a.yln) *reinterpret_cast<f32x4*>(a.yln + o) = v;
      if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + o);
      if (a.resid2) v += *reinterpret_cast<const f32x4*>(a.resid2 + o);   // (LN + x) + skip, in this order
      *reinterpret_cast<f32x4*>(a.y + o) = v;
    }
    if (a.rstd && w == 0 && lg == 0) a.rstd[row] = rstd;
  } else {  // OUT_SMALL: the narrow last Linear (decoder, models/model.py:22) on the VALU; output channel c belongs to wave c % 4
    for (int c = w; c < a.C; c += 4) {
      const float v = dot_features<NB>(z, a.wout + c * D, lg);
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

template <int NB, int GIN, int FIRST, bool BF = false, bool LONE = false>
__global__ __launch_bounds__(kChainMaxThreads) __attribute__((amdgpu_waves_per_eu(NB <= 8 ? (LONE ? 2 : BSMS_CHAIN_WPE) : 2))) void k_chain_bwd(ChainBwdArgs a) {
  constexpr int D = NB * 16;
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4;
  const int cw = int(blockDim.x >> 6) - a.nload;   // compute waves of this launch (4..7, chosen by the launcher); the last wave(s) load
  if (wave >= cw) {  // loader wave (uniform branch)
    loader_dispatch<NB, BF ? 1 : kPL>(a.nload, wave - cw, a.wseq, a.nseq, lds, lane, a.ntiles, a.nring);
    return;
  }
  Slot slot{0, a.nring};  // ring slot of the next chunk, across this workgroup's tiles
  float4* const ring = lds + Ring<NB>::PRE4;
  unsigned* brow = bound_row<NB>(lds, wave, lane, any_slot(a.gmax));   // this wave's running magnitude bounds
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {  // persistent workgroups (see k_chain_fwd)
  const int64_t row = int64_t(tile) * (16 * cw) + wave * 16 + (lane & 15);
  const bool live = row < a.R;
  const int64_t rowc = live ? row : 0;   // what a lane past the end reads (its results are never stored)
  const int64_t roff = live ? row * D : -1;  // row offset for stores; negative = no store

  f32x4 g[NB], acc[NB];
  if (GIN == G_SMALL) {  // g = (dy . W_out) masked by the last hidden activation
    zero_tile<NB>(g);
    for (int c = 0; c < a.C; ++c) axpy_features<NB>(g, a.wout + c * D, a.dy[rowc * a.C + c], lg);
    mask_by<NB>(g, a.mask_in + rowc * D, lg);
  } else {
    const float* dyrow;
    if (GIN == G_EDGE_LN) {  // autograd of scatter_sum: gather the node gradient by target
      const int b = int(rowc / a.E), q = int(rowc - int64_t(b) * a.E);
      dyrow = a.dy + (int64_t(b) * a.N + a.dst[q]) * D;
    } else {
      dyrow = a.dy + rowc * D;
    }
    load_rows<NB>(g, dyrow, lg);
    if constexpr (BF && GIN == G_EDGE_LN) load_rows_bf16<NB>(acc, a.yln, rowc, lg);   // the bf16 messages the forward handed to the aggregation
    else load_rows<NB>(acc, a.yln + rowc * D, lg);  // acc = normalised output y
    const float rs = a.rstd[rowc];
    __builtin_amdgcn_sched_barrier(0);         // all 17 loads in flight before the first use (see k_chain_fwd)
    // LayerNorm backward (no affine): dz = rstd * (dy - mean(dy) - y * mean(dy * y))
    const float m1 = row_sum<NB>(g) * (1.f / D);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) s2 = fmaf(g[t][r], acc[t][r], s2);
    s2 = group_sum(s2);
    const float m2 = s2 * (1.f / D);
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) g[t][r] = rs * (g[t][r] - m1 - acc[t][r] * m2);
  }
  // The gradient entering a stage is stored to HBM from inside that stage (mfma_stage), so the store has a whole
  // stage to drain before the next vmcnt wait (the ReLU-mask rows at the end of the stage).
  float* pending = a.gstore[0];   // uniform

  for (int k = 0; k < a.nstage; ++k) {
    // ReLU sign bits of the activation that masks this stage's output: one small load, issued before the stage
    unsigned mbits[mask_words<NB>()];
#pragma unroll
    for (int w = 0; w < mask_words<NB>(); ++w)
      mbits[w] = a.mask[k] ? reinterpret_cast<const unsigned*>(a.mask[k] + (BF ? pad_rows(a.R) * D / 2 : pad_rows(a.R) * D))[rowc * (4 * mask_words<NB>()) + lg * mask_words<NB>() + w]
                           : 0xffffffffu;
    if constexpr (BF) {
      zero_tile<NB>(acc);
      mfma_stage_bf<NB>(acc, g, ring, slot, lane, false, pending, roff, 0);
    } else {
      const float m = row_amax<NB>(g);
      note_amax(brow, k, m, lane);
      mfma_stage<NB, true, 1, LONE>(acc, g, scale_of(m), ring, slot, lane, pending, roff, a.store_mode & 3, 0, nullptr, row,
                              (a.store_mode & 8) ? 0 : a.R);
    }
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // bit -> all-ones / zero mask (one v_bfe_i32), then one and
        const int keep = __builtin_amdgcn_sbfe((int)mbits[(4 * t + r) >> 5], (4 * t + r) & 31, 1);
        g[t][r] = __uint_as_float(__float_as_uint(acc[t][r]) & (unsigned)keep);
      }
    pending = a.gstore[k + 1];
  }

  if constexpr (FIRST != F_NONE && BF) {   // BSMS_BF16_NODES: gN[0] stays fp32 (its weight-gradient job multiplies the fp32 rows x / aggr)
    note_amax(brow, a.nstage, row_amax<NB>(g), lane);
    store_rows<NB, false>(g, pending, roff, lg);
    pending = nullptr;
    zero_tile<NB>(acc);
    mfma_stage_bf<NB>(acc, g, ring, slot, lane, false, nullptr, roff, 0);
    if (a.dres) {
      f32x4 r[NB];
      load_rows<NB>(r, a.dres + rowc * D, lg);
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] += r[t];
    }
    store_rows<NB, false>(acc, a.dx, roff, lg);
    if (FIRST == F_HEADS2) {
      zero_tile<NB>(acc);
      mfma_stage_bf<NB>(acc, g, ring, slot, lane, false, nullptr, roff, 0);
      store_rows<NB, false>(acc, a.dx2, roff, lg);
    }
  } else if (FIRST != F_NONE) {
    const float mh = row_amax<NB>(g);
    note_amax(brow, a.nstage, mh, lane);
    const RowScale rs = scale_of(mh);
    mfma_stage<NB, true, 1, LONE>(acc, g, rs, ring, slot, lane, pending, roff, 1, 0, nullptr, row, a.R);
    pending = nullptr;
    if (a.dres) {
      f32x4 r[NB];
      load_rows<NB>(r, a.dres + rowc * D, lg);
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] += r[t];
    }
    if (FIRST == F_HEADS2) {
      f32x4 acc2[NB];
      mfma_stage<NB, true, 1, LONE>(acc2, g, rs, ring, slot, lane);
      store_rows<NB, false>(acc, a.dx, roff, lg);
      store_rows<NB, false>(acc2, a.dx2, roff, lg);
    } else {
      store_rows<NB, false>(acc, a.dx, roff, lg);
    }
  }
  if (FIRST == F_NONE && a.gmax[a.nstage]) note_amax(brow, a.nstage, row_amax<NB>(g), lane);   // uniform
  if constexpr (BF) store_rows_bf16<NB>(g, pending, roff, lg);
  else store_rows<NB, false>(g, pending, roff, lg);
  }  // tile loop
  flush_bounds(a.gmax, kMaxStages + 1, brow, wave, lane);
}

// ------------------------------------------------------------------ small launches: feature-split backward chain ----
// k_chain_bwd with the features of a 16-row tile split over four waves (see k_fs_fwd).  The LayerNorm backward needs sums
// over the whole row in the association of k_chain_bwd (row_sum, then the sequential fmaf chain): every wave reads the full
// dy / y rows (L2-resident at these sizes) and repeats that arithmetic, then keeps its own feature blocks.  Bit-identical
// to k_chain_bwd.
template <int GIN, int FIRST>
__global__ __launch_bounds__(256) void k_fs_bwd(ChainBwdArgs a) {
  constexpr int NB = 8, D = 128;
  __shared__ FsLds L;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lg = lane >> 4;
  const int64_t row = int64_t(blockIdx.x) * 16 + (lane & 15);
  const bool live = row < a.R;
  const int64_t rowc = live ? row : 0;
  FsPack pa, pb;
  fs_request(pa, a.wseq[0], w, lane);
  f32x4 own[2], acc[2];
  if (GIN == G_SMALL) {  // g = (dy . W_out) masked by the last hidden activation
#pragma unroll
    for (int i = 0; i < 2; ++i) own[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < a.C; ++c) {
      const float dv = a.dy[rowc * a.C + c];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 wv = *reinterpret_cast<const float4*>(a.wout + c * D + 16 * (2 * w + i) + 4 * lg);
        own[i][0] = fmaf(dv, wv.x, own[i][0]);
        own[i][1] = fmaf(dv, wv.y, own[i][1]);
        own[i][2] = fmaf(dv, wv.z, own[i][2]);
        own[i][3] = fmaf(dv, wv.w, own[i][3]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 mv = *reinterpret_cast<const float4*>(a.mask_in + rowc * D + 16 * (2 * w + i) + 4 * lg);
      own[i][0] = mv.x > 0.f ? own[i][0] : 0.f;
      own[i][1] = mv.y > 0.f ? own[i][1] : 0.f;
      own[i][2] = mv.z > 0.f ? own[i][2] : 0.f;
      own[i][3] = mv.w > 0.f ? own[i][3] : 0.f;
    }
  } else {  // G_ROWS_LN: LayerNorm backward (no affine): dz = rstd * (dy - mean(dy) - y * mean(dy * y)) on the FULL row
    f32x4 gf[NB], yf[NB];
    load_rows<NB>(gf, a.dy + rowc * D, lg);
    load_rows<NB>(yf, a.yln + rowc * D, lg);
    const float rs = a.rstd[rowc];
    const float m1 = row_sum<NB>(gf) * (1.f / D);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) s2 = fmaf(gf[t][r], yf[t][r], s2);
    s2 = group_sum(s2);
    const float m2 = s2 * (1.f / D);
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) gf[t][r] = rs * (gf[t][r] - m1 - yf[t][r] * m2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // own feature blocks 2w, 2w+1 (w is wave-uniform: a select over the register tile)
      own[i] = gf[i];
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (w == q) own[i] = gf[2 * q + i];
    }
  }
  auto store_own = [&](float* base, const f32x4 (&v)[2]) {   // a layer gradient: rows [R, D] fp32, own quarter
    if (!base || !live) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4*>(base + row * D + 16 * (2 * w + i) + 4 * lg));
  };
  FsPieces x;
  RowScale rs{};
  auto enter = [&](int k) {   // the gradient entering pack k: store, bound, row scale, pieces
    store_own(a.gstore[k], own);
    if (k > 0) lds_barrier();   // the pieces of the previous gradient have been read by everybody
    const float m = fs_row_max(L, fs_amax2(own), w, lane);
    fs_note(a.gmax[k], m, w, lane);
    rs = scale_of(m);
    fs_publish(L, x, own, rs.s, w, lane);
  };
  auto stage = [&](FsPack& cur, FsPack& nxt, int q, int k) {   // dgrad through layer k, masked by the ReLU sign bits of its input activation
    unsigned mb = 0xffu;
    if (a.mask[k]) mb = reinterpret_cast<const unsigned char*>(a.mask[k] + pad_rows(a.R) * D)[(rowc * 4 + lg) * 4 + w];
    fs_stage<true, 1>(acc, cur, x, rs.E, lane, &nxt, q + 1 < a.nseq ? a.wseq[q + 1] : nullptr, w);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int keep = __builtin_amdgcn_sbfe((int)mb, 4 * i + r, 1);
        own[i][r] = __uint_as_float(__float_as_uint(acc[i][r]) & (unsigned)keep);
      }
  };
  // the two pack register sets alternate with STATIC roles (a run-time choice between them would put both in scratch)
  auto finish = [&](FsPack& cur, FsPack& nxt, int q) {
    if (FIRST == F_NONE) {
      if (a.gmax[a.nstage]) {   // uniform
        lds_barrier();
        fs_note(a.gmax[a.nstage], fs_row_max(L, fs_amax2(own), w, lane), w, lane);
      }
      store_own(a.gstore[a.nstage], own);
      return;
    }
    enter(a.nstage);
    fs_stage<true, 1>(acc, cur, x, rs.E, lane, &nxt, q + 1 < a.nseq ? a.wseq[q + 1] : nullptr, w);
    if (a.dres && live) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] += *reinterpret_cast<const f32x4*>(a.dres + row * D + 16 * (2 * w + i) + 4 * lg);
    }
    if (live)
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(a.dx + row * D + 16 * (2 * w + i) + 4 * lg) = acc[i];
    if (FIRST == F_HEADS2) {
      fs_stage<true, 1>(acc, nxt, x, rs.E, lane);
      if (live)
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(a.dx2 + row * D + 16 * (2 * w + i) + 4 * lg) = acc[i];
    }
  };
  for (int k = 0;;) {
    if (k == a.nstage) { finish(pa, pb, k); break; }
    enter(k);
    stage(pa, pb, k, k);
    ++k;
    if (k == a.nstage) { finish(pb, pa, k); break; }
    enter(k);
    stage(pb, pa, k, k);
    ++k;
  }
}

// store_pair_stream for the pipelined edge kernels: the tensor is non-null and padded (no tests), and the per-lane
// part of both addresses is a 32-bit byte offset WITHIN THE TILE, computed once per kernel (the tile's base is uniform
// 64-bit scalar arithmetic), so a pair costs its DPP exchange and two stores with SGPR base + VGPR offset + immediate.
struct PairOff { unsigned a, b; };
template <int NB>
__device__ __forceinline__ PairOff pair_offsets(int64_t row, int lane) {
  constexpr int D = NB * 16;
  const bool hi = (lane & 8) != 0;
  const int64_t rowA = row - (lane & 8);
  return PairOff{unsigned((rowA * D + 4 * (lane >> 4) + (hi ? 16 : 0)) * 4), unsigned(((rowA + 8) * D + 4 * (lane >> 4) + (hi ? 0 : 16)) * 4)};
}
template <int NB>
__device__ __forceinline__ void store_pair_nt(const f32x4 (&v)[NB], float* base, PairOff off, int lane, int t) {
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  const bool hi = (lane & 8) != 0;
  const i32x4 own = __builtin_bit_cast(i32x4, v[t + 1]);
  i32x4 got;
#pragma unroll
  for (int r = 0; r < 4; ++r) got[r] = __builtin_amdgcn_update_dpp(own[r], own[r], 0x128, 0xf, 0xf, false);
  const f32x4 x = __builtin_bit_cast(f32x4, got);
  const f32x4 dA = hi ? x : v[t], dB = hi ? v[t] : x;
  char* b = reinterpret_cast<char*>(base) + 64 * t;   // uniform
  __builtin_nontemporal_store(dA, reinterpret_cast<f32x4*>(b + off.a));
  __builtin_nontemporal_store(dB, reinterpret_cast<f32x4*>(b + off.b));
}

// ------------------------------------------------------------- edge MLP chains, software-pipelined ----
// The edge MLP (IN_EDGE / OUT_LN forward, G_EDGE_LN / F_NONE backward) is ~45 % of the training step.  In k_chain_fwd /
// k_chain_bwd every A-fragment pair is read from LDS right before its MFMAs (the 128-VGPR budget of two workgroups
// per CU leaves no room to prefetch), so an in-order wave exposes one LDS round trip per pair.
// Here a wave owns RB row blocks of 16 rows: ONE fragment pair feeds 2 RB x {2, 1} MFMAs, the next pair is in
// flight while they run, 2 RB independent accumulator chains interleave (no dependent back-to-back MFMAs), and the
// workgroup barrier + bias reads are paid once per RB x 64 rows.  RB = 2 at D = 128 (one workgroup per CU, 256-VGPR
// budget), RB = 1 at D = 256 (the 32 + 32 blocks of one row block already fill the budget).
// The arithmetic (order of the three partial products per accumulator, chain.h) is exactly mfma_stage's: bit-identical.
// The VALU work of a stage, cut into STEPS of 2-4 operations that are placed by hand between the MFMA pairs of the
// chunk before the one that needs them (sched_barrier fences keep hipcc from regrouping them: left alone it emits the
// split of a K block as one lump during which the matrix pipe drains, and its IGroupLP pipelines (sched_group_barrier)
// either explode in compile time or silently skip some regions).  Per row block:
//   P0..P3  streaming store of feature blocks 2c, 2c + 1 of the activation: DPP exchange (2 steps), select + store (2)
//   S0..S7  fp16 pieces of K block c + 1: per dword (two features) {h}, {l}  (two v_fma_mix each)
//   M0..M7  (last chunk of a saved activation instead of S) ReLU sign bits, then the store of the words
struct Pieces { unsigned h[4], l[4]; };
__device__ __forceinline__ u32x4 vec4(const unsigned (&d)[4]) { return u32x4{d[0], d[1], d[2], d[3]}; }
struct StepState { int got[4]; };

// SAVE: 0 nothing is stored; 1 the activation as fp32 rows (128-byte streaming pairs) + sign bits; 2 (round 6, the fused fp32 edge
// backward, efuse32.hip) the fp16 x 2 PIECES this stage multiplies anyway -- K block c of a row as 64 bytes of h pieces followed
// by 64 bytes of l pieces at byte 512 row + 128 c of the tensor (the four lane groups of a row write 16 bytes each), + sign bits;
// the row's scale exponent goes to a side array (stage_rb).  Same bytes as the fp32 row, no DPP exchange, two steps instead of four.
template <int NB, int RB, int SAVE, bool MASK>
__device__ __forceinline__ void valu_step(int s, int c, const f32x4 (&act)[RB][NB], Pieces (&pc)[RB][2], StepState (&st)[RB],
                                          const RowScale (&rs)[RB], unsigned (&mword)[RB][mask_words<NB>()], float* store_base,
                                          unsigned* bits_base, const PairOff (&off)[RB], const unsigned (&moff)[RB], int lane) {
  constexpr int W = mask_words<NB>(), NP = SAVE == 2 ? 2 : (SAVE ? 4 : 0), PER = NP + 8;
  const int rb = s / PER, q = s % PER;
  if (rb >= RB) return;
  const bool last = c + 1 == Ring<NB>::NCH;
  if (SAVE == 2 && q < NP) {   // ---- P steps, pieces: K block c is being multiplied right now (pc[.][c & 1]); S steps write the other slot
    const Pieces& pcs = pc[rb][c & 1];
    char* b = reinterpret_cast<char*>(store_base) + 128 * c + 64 * q;   // uniform
    *reinterpret_cast<u32x4*>(b + off[rb].a) = q == 0 ? vec4(pcs.h) : vec4(pcs.l);   // plain stores: 64-byte pieces per row (streaming stores of that size run at 3.1 TB/s, plain ones are pattern-insensitive; census/store_bw.hip)
    return;
  }
  if (q < NP) {   // ---- P steps
    const f32x4& own = act[rb][2 * c + 1];
    const bool hi = (lane & 8) != 0;
    if (q < 2) {
#pragma unroll
      for (int r = 2 * q; r < 2 * q + 2; ++r) {
        const int x = __float_as_int(own[r]);
        st[rb].got[r] = __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false);   // row_ror:8: partner's block 2c + 1
      }
    } else {
      const f32x4& mine = act[rb][2 * c];
      f32x4 d;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g = __int_as_float(st[rb].got[r]);
        d[r] = (q == 2) ? (hi ? g : mine[r]) : (hi ? mine[r] : g);
      }
      char* b = reinterpret_cast<char*>(store_base) + 128 * c;   // uniform; feature blocks 2c, 2c + 1
      __builtin_nontemporal_store(d, reinterpret_cast<f32x4*>(b + (q == 2 ? off[rb].a : off[rb].b)));
    }
    return;
  }
  const int ss = q - NP;
  if (!last) {    // ---- S steps: K block c + 1 -> pc[rb][(c + 1) & 1]
    const int v = ss >> 1, kb2 = c + 1;
    Pieces& o = pc[rb][kb2 & 1];
    const float x0 = act[rb][2 * kb2 + (v >> 1)][2 * (v & 1)], x1 = act[rb][2 * kb2 + (v >> 1)][2 * (v & 1) + 1];
    if ((ss & 1) == 0) {
      asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(o.h[v]) : "v"(x0), "v"(rs[rb].s));
      asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(o.h[v]) : "v"(x1), "v"(rs[rb].s));
    } else {
      asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(o.l[v]) : "v"(x0), "v"(rs[rb].s), "v"(o.h[v]));
      asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(o.l[v]) : "v"(x1), "v"(rs[rb].s), "v"(o.h[v]));
    }
  } else if (SAVE && MASK) {   // ---- M steps: highest element first, one shift-and-append per element (store_mask_bits)
    constexpr int EPS = (4 * NB + 7) / 8;
#pragma unroll
    for (int i = ss * EPS; i < (ss + 1) * EPS && i < 4 * NB; ++i) {
      const int e = 4 * NB - 1 - i;
      mword[rb][e >> 5] = __builtin_amdgcn_alignbit(mword[rb][e >> 5], 0u - __float_as_uint(act[rb][e >> 2][e & 3]), 31);
    }
    if (ss == 7) {
      unsigned* bits = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(bits_base) + moff[rb]);
#pragma unroll
      for (int w = 0; w < W; ++w) bits[w] = mword[rb][w];   // rows past R land in the padding (chain.h: act_floats)
    }
  }
}

// ZERO / FIN as in mfma_stage: ZERO = the accumulators start from zero (else they continue the raw sums of the previous
// call: same row scales, pack of the same weight scale); FIN = 0 leave raw sums, 1 un-scale, 2 un-scale + bias.
// `rs_ext` (nullable): row scales decided by the caller (a Linear over two concatenated sources); else the row maxima of
// `act` are taken here and noted in the running bounds (`brow`, stage index `stage`).
template <int NB, int RB, int SAVE, bool MASK, bool ZERO, int FIN, bool LONE = false>
__device__ __forceinline__ void stage_rb(f32x4 (&acc)[RB][NB], const f32x4 (&act)[RB][NB], float4* lds, Slot& slot, int lane,
                                         float* store_base, unsigned* bits_base, const PairOff (&off)[RB], const unsigned (&moff)[RB],
                                         unsigned* brow, int stage, const RowScale* rs_ext = nullptr,
                                         unsigned long long* waited = nullptr,   // experiments: cycles at the chunk barriers
                                         int* exps_tile = nullptr) {            // SAVE == 2: scale exponents of this tile's rows
  using Rg = Ring<NB>;
  constexpr int W = mask_words<NB>();
  constexpr int NSLOT = (NB / 2) * 3 * RB;            // MFMA pairs per chunk
  constexpr int NSTEP = RB * ((SAVE == 2 ? 2 : (SAVE ? 4 : 0)) + 8);    // VALU steps per chunk
  static_assert(NSTEP <= NSLOT, "at most one step per MFMA pair");
  Pieces pc[RB][2];                                   // pieces of K blocks c (slot c & 1) and c + 1
  StepState st[RB];
  RowScale rs[RB];
  unsigned mword[RB][W];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    if (rs_ext) {
      rs[rb] = rs_ext[rb];
    } else {
      const float m = row_amax<NB>(act[rb]);
      note_amax(brow, stage, m, lane);
      rs[rb] = scale_of(m);
    }
    u32x4 h, l;
    split_block<NB>(act[rb], 0, rs[rb].s, h, l);
#pragma unroll
    for (int v = 0; v < 4; ++v) { pc[rb][0].h[v] = h[v]; pc[rb][0].l[v] = l[v]; }
#pragma unroll
    for (int w = 0; w < W; ++w) mword[rb][w] = 0;
    if (SAVE == 2 && (lane >> 4) == 0) exps_tile[moff[rb] / (16 * W)] = rs[rb].E;   // moff = (row in tile * 4 W + group * W) * 4 bytes
  }
  int fw = 0;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if constexpr (LONE) {
    // ---- single-round launches (one workgroup per CU, one or two waves per SIMD, 256-register budget): nothing hides the
    // chunk barrier and the LDS round trip of a chunk's first fragments (~350 of ~900 cycles per chunk, profiles/census/
    // stage_lone.hip).  As in mfma_stage<LONE>: the wave passes the barrier of chunk c + 1 and requests its first fragments
    // at the START of chunk c's last block pair -- whose own fragments (plane l) were requested one pair early -- so the
    // round trip runs under six MFMAs per row block.  One barrier per chunk, in the same order; after barrier c + 1 this
    // wave has nothing left to read of chunk c (lds_barrier waits for its LDS reads), so the loader may overwrite it.
    static_assert(NB >= 4, "the early barrier needs two block pairs per chunk");
    const float4* cur = nullptr;
    const float4* body = nullptr;
    float4 f0, f1;
#pragma unroll
    for (int c = 0; c < Rg::NCH; ++c) {
      if (c == 0) {
        lds_barrier();
        cur = lds + slot.i * Rg::CH4;
        if (++slot.i == slot.nr) slot.i = 0;
        body = cur + kChunkHdrFloats / 4 + lane;
        f0 = body[0]; f1 = body[2 * 64];
        fw = int(__float_as_uint(reinterpret_cast<const float*>(cur)[kScaleSlot]) >> 23);
      }
      const int cb = c & 1;
      int islot = 0;
      auto pair = [&](int t, int rb, const float4& a0, const float4& a1, const unsigned (&piece)[4], bool first) {
        acc[rb][t] = mma(a0, vec4(piece), (ZERO && first && c == 0) ? zero : acc[rb][t]);
        acc[rb][t + 1] = mma(a1, vec4(piece), (ZERO && first && c == 0) ? zero : acc[rb][t + 1]);
        const int s = (islot * NSTEP + NSLOT - 1) / NSLOT;
        if (s < NSTEP && s * NSLOT / NSTEP == islot)
          valu_step<NB, RB, SAVE, MASK>(s, c, act, pc, st, rs, mword, store_base, bits_base, off, moff, lane);
        ++islot;
        __builtin_amdgcn_sched_barrier(0);
      };
      __builtin_amdgcn_sched_barrier(0);
      float4 m0 = f0, m1 = f1;   // plane l of the LAST block pair, requested one pair early
      float4 g0 = f0, g1 = f1;   // first fragments of the NEXT chunk
      const float4* ncur = cur;
      const float4* nbody = body;
#pragma unroll
      for (int t = 0; t < NB; t += 2) {
        float4 n0, n1;
        if (t == NB - 2) {
          n0 = m0; n1 = m1;
          if (c + 1 < Rg::NCH) {   // every read of this chunk has been issued: barrier of the next one, its first fragments
            lds_barrier();
            ncur = lds + slot.i * Rg::CH4;
            if (++slot.i == slot.nr) slot.i = 0;
            nbody = ncur + kChunkHdrFloats / 4 + lane;
            g0 = nbody[0]; g1 = nbody[2 * 64];
          }
        } else {
          n0 = body[(t * 2 + 1) * 64]; n1 = body[(t * 2 + 3) * 64];
          if (t == NB - 4) { m0 = body[((NB - 2) * 2 + 1) * 64]; m1 = body[((NB - 2) * 2 + 3) * 64]; }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) pair(t, rb, f0, f1, pc[rb][cb].l, true);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) pair(t, rb, f0, f1, pc[rb][cb].h, false);
        f0 = n0;
        f1 = n1;
        if (t + 2 < NB) {
          n0 = body[((t + 2) * 2) * 64];
          n1 = body[((t + 2) * 2 + 2) * 64];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) pair(t, rb, f0, f1, pc[rb][cb].h, false);
        f0 = n0;
        f1 = n1;
      }
      if (FIN != 0 && c == Rg::NCH - 1) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) finish_stage<NB, FIN == 2>(acc[rb], rs[rb].E, fw, reinterpret_cast<const float*>(cur), lane);
      }
      cur = ncur; body = nbody; f0 = g0; f1 = g1;
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < Rg::NCH; ++c) {
#ifdef BSMS_EXPERIMENTS
    if (waited) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      lds_barrier();
      *waited += __builtin_amdgcn_s_memtime() - t0;
    } else
#endif
    lds_barrier();                                         // chunk has landed (and my reads of the last one are done)
    const float4* cur = lds + slot.i * Rg::CH4;
    if (++slot.i == slot.nr) slot.i = 0;
    const float4* body = cur + kChunkHdrFloats / 4 + lane;
    float4 f0 = body[0], f1 = body[2 * 64];                // pair (t = 0, plane h)
    if (c == 0) fw = int(__float_as_uint(reinterpret_cast<const float*>(cur)[kScaleSlot]) >> 23);
    const int cb = c & 1;
    int islot = 0;   // MFMA pair within the chunk
    // one MFMA pair (feature blocks t, t + 1 of row block rb, one plane combination), then the VALU step that rides with it
    auto pair = [&](int t, int rb, const float4& a0, const float4& a1, const unsigned (&piece)[4], bool first) {
      acc[rb][t] = mma(a0, vec4(piece), (ZERO && first && c == 0) ? zero : acc[rb][t]);
      acc[rb][t + 1] = mma(a1, vec4(piece), (ZERO && first && c == 0) ? zero : acc[rb][t + 1]);
      const int s = (islot * NSTEP + NSLOT - 1) / NSLOT;          // the step whose place is this pair, if any
      if (s < NSTEP && s * NSLOT / NSTEP == islot)
        valu_step<NB, RB, SAVE, MASK>(s, c, act, pc, st, rs, mword, store_base, bits_base, off, moff, lane);
      ++islot;
      __builtin_amdgcn_sched_barrier(0);
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NB; t += 2) {
      float4 n0 = body[(t * 2 + 1) * 64], n1 = body[(t * 2 + 3) * 64];          // plane l of (t, t + 1): one pair ahead
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) pair(t, rb, f0, f1, pc[rb][cb].l, true);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) pair(t, rb, f0, f1, pc[rb][cb].h, false);
      f0 = n0;
      f1 = n1;
      if (t + 2 < NB) {                                                           // plane h of the next block pair
        n0 = body[((t + 2) * 2) * 64];
        n1 = body[((t + 2) * 2 + 2) * 64];
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) pair(t, rb, f0, f1, pc[rb][cb].h, false);
      f0 = n0;
      f1 = n1;
    }
    if (FIN != 0 && c == Rg::NCH - 1) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) finish_stage<NB, FIN == 2>(acc[rb], rs[rb].E, fw, reinterpret_cast<const float*>(cur), lane);
    }
  }
}

template <int NB, int RB>
struct EdgeTile {
  static constexpr int rows = kTileRows * RB;
  static constexpr int waves_per_eu = (NB * RB <= 8) ? 4 : 2;   // VGPR budget 128 / 256
  static constexpr int resident = (NB * RB <= 8) ? 2 : 1;       // workgroups per CU (see resident_per_cu)
};

// LONE: the instantiation for launches of at most one workgroup per CU (stage_rb<.., LONE>; 256-register budget)
template <int NB, int RB, int SAVE, bool LONE = false>
__global__ __launch_bounds__(kChainMaxThreads) __attribute__((amdgpu_waves_per_eu(LONE ? 2 : EdgeTile<NB, RB>::waves_per_eu)))
void k_edge_fwd(ChainFwdArgs a) {
  constexpr int D = NB * 16;
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4;
  const int cw = int(blockDim.x >> 6) - a.nload, tile_rows = 16 * RB * cw;   // compute waves of this launch (launcher's choice); the last wave(s) load
  if (wave >= cw) {  // loader wave (uniform branch)
    loader_dispatch<NB>(a.nload, wave - cw, a.wseq, a.nseq, lds, lane, a.ntiles, a.nring, a.w0t);
    return;
  }
  const float rcpE = 1.f / float(a.E);
  // plan-order endpoints of this lane's rows in a tile; fetched one tile ahead (two registers per row block), so a tile
  // starts with its row gathers instead of a dependent index round trip
  auto fetch_endpoints = [&](int tile, int (&i)[RB], int (&j)[RB], int (&b)[RB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int64_t r = int64_t(tile) * tile_rows + wave * (16 * RB) + rb * 16 + (lane & 15);
      const EdgeRef e = edge_ref(unsigned(r < a.R ? r : 0), unsigned(a.E), rcpE);   // a lane past the end reads row 0
      i[rb] = a.src[e.q];
      j[rb] = a.dst[e.q];
      b[rb] = e.b;
    }
  };
  int ni[RB], nj[RB], nbat[RB];
  fetch_endpoints(blockIdx.x, ni, nj, nbat);
  lds_barrier();
  const float* w0t = reinterpret_cast<const float*>(lds);   // fiber weights (LDS side table, see k_chain_fwd)
  Slot slot{0, a.nring};
  float4* const ring = lds + Ring<NB>::PRE4;
  unsigned* brow = bound_row<NB>(lds, wave, lane, any_slot(a.amax));   // this wave's running magnitude bounds
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int64_t row[RB];
    PairOff off[RB];
    unsigned moff[RB];   // byte offset of this lane's sign-bit words
    f32x4 act[RB][NB], acc[RB][NB];
    float pi[RB][7], pj[RB][7];
#ifdef BSMS_EXPERIMENTS
    int stamp_i = 0;
    unsigned long long waited = 0;
    auto stamp = [&]() { if (a.timing && tid == 0 && stamp_i < 16) a.timing[int64_t(tile) * 16 + stamp_i++] = __builtin_amdgcn_s_memtime(); };
#else
    auto stamp = [] {};
#endif
    stamp();
    // ---- input stage: relu(Ps[src] + Pd[dst] + Wf . [pos_i - pos_j, |pos_i - pos_j|])   (ops/basic.py:70-92)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      row[rb] = int64_t(tile) * tile_rows + wave * (16 * RB) + rb * 16 + (lane & 15);
      off[rb] = pair_offsets<NB>(wave * (16 * RB) + rb * 16 + (lane & 15), lane);                         // within the tile
      if (SAVE == 2) off[rb].a = unsigned((wave * (16 * RB) + rb * 16 + (lane & 15)) * (4 * D) + 16 * lg);    // pieces: this lane's 16 bytes of a K block's h run (valu_step)
      moff[rb] = unsigned(((wave * (16 * RB) + rb * 16 + (lane & 15)) * (4 * mask_words<NB>()) + lg * mask_words<NB>()) * 4);
      const int i = ni[rb], j = nj[rb], b = nbat[rb];
      load_rows<NB>(act[rb], a.Ps + (int64_t(b) * a.N + i) * D, lg);
      load_rows<NB>(acc[rb], a.Pd + (int64_t(b) * a.N + j) * D, lg);
      const float* pb = a.pos + b * a.pos_bstride;
      if (a.p == 2) {   // uniform; the common widths load whole points
        const float2 xi = *reinterpret_cast<const float2*>(pb + int64_t(i) * 2), xj = *reinterpret_cast<const float2*>(pb + int64_t(j) * 2);
        pi[rb][0] = xi.x; pi[rb][1] = xi.y; pj[rb][0] = xj.x; pj[rb][1] = xj.y;
#pragma unroll
        for (int c = 2; c < 7; ++c) pi[rb][c] = pj[rb][c] = 0.f;
      } else {
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          const int cc = c < a.p ? c : 0;   // uniform clamp: the loads stay unconditional
          pi[rb][c] = pb[int64_t(i) * a.p + cc];
          pj[rb][c] = pb[int64_t(j) * a.p + cc];
        }
      }
    }
    if (tile + int(gridDim.x) < a.ntiles) fetch_endpoints(tile + gridDim.x, ni, nj, nbat);   // uniform; lands under the stages
    __builtin_amdgcn_sched_barrier(0);   // all gathers of the tile are in flight before the first use
    stamp();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int t = 0; t < NB; ++t) act[rb][t] += acc[rb][t];
      float n2 = 0.f;
#pragma unroll
      for (int c = 0; c < 7; ++c)
        if (c < a.p) {
          const float rel = pi[rb][c] - pj[rb][c];
          n2 = fmaf(rel, rel, n2);
          axpy_features<NB>(act[rb], w0t + c * D, rel, lg);
        }
      const float nrm = sqrtf(n2);
      axpy_features<NB>(act[rb], w0t + a.p * D, nrm, lg);
      relu_into<NB>(act[rb], act[rb]);
      if (a.fiber_out && row[rb] < a.R && lg == 0) {   // one lane per row keeps the fiber for the backward (uniform: null in inference; the fused fp32 backward needs it from the launch that saves nothing else)
        float f[8];
#pragma unroll
        for (int c = 0; c < 7; ++c) f[c] = c < a.p ? pi[rb][c] - pj[rb][c] : (c == a.p ? nrm : 0.f);
        f[7] = a.p == 7 ? nrm : 0.f;
        const int ld = fiber_ld(a.p);
        float4* dst = reinterpret_cast<float4*>(a.fiber_out + row[rb] * ld);
        dst[0] = make_float4(f[0], f[1], f[2], f[3]);
        if (ld == 8) dst[1] = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
    // ---- MFMA stages; the activation entering a stage is stored (values + sign bits) from inside that stage
    float* pending = a.store_in;   // uniform; non-null when SAVE (launcher)
    int* pending_exp = SAVE == 2 ? a.store_exp[0] : nullptr;   // uniform; non-null when SAVE == 2 (launcher)
    stamp();
    for (int l = 0; l < a.nstage; ++l) {
      float* st_tile = SAVE ? pending + int64_t(tile) * (tile_rows * D) : nullptr;   // uniform
      unsigned* bits_tile = SAVE ? reinterpret_cast<unsigned*>(pending + pad_rows(a.R) * D) + int64_t(tile) * (tile_rows * 4 * mask_words<NB>()) : nullptr;
      int* exps_tile = SAVE == 2 ? pending_exp + int64_t(tile) * tile_rows : nullptr;
#ifdef BSMS_EXPERIMENTS
      stage_rb<NB, RB, SAVE, true, true, 2, LONE>(acc, act, ring, slot, lane, st_tile, bits_tile, off, moff, brow, l, nullptr, a.timing ? &waited : nullptr, exps_tile);
#else
      stage_rb<NB, RB, SAVE, true, true, 2, LONE>(acc, act, ring, slot, lane, st_tile, bits_tile, off, moff, brow, l, nullptr, nullptr, exps_tile);   // acc = bias + W act
#endif
      stamp();
      if (l + 1 < a.nstage) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) relu_into<NB>(act[rb], acc[rb]);
        pending = a.store[l];
        if (SAVE == 2) pending_exp = a.store_exp[l + 1];
      }
    }
    // ---- LayerNorm(elementwise_affine=False), eps 1e-5  (ops/basic.py:18)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const float mean = row_sum<NB>(acc[rb]) * (1.f / D);
      float ss = 0.f;
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[rb][t][r] -= mean;
          ss = fmaf(acc[rb][t][r], acc[rb][t][r], ss);
        }
      ss = group_sum(ss);
      const float rstd = 1.f / sqrtf(ss * (1.f / D) + 1e-5f);
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rb][t][r] *= rstd;
      const int64_t roff = row[rb] < a.R ? row[rb] * D : -1;
      store_rows<NB, false>(acc[rb], a.yln, roff, lg);
      if (a.rstd && roff >= 0 && lg == 0) a.rstd[row[rb]] = rstd;
      store_rows<NB, false>(acc[rb], a.y, roff, lg, a.out_mode);
    }
    stamp();
#ifdef BSMS_EXPERIMENTS
    if (a.timing && tid == 0) a.timing[int64_t(tile) * 16 + 11] = waited;
#endif
  }
  if (SAVE) flush_bounds(a.amax, kMaxStages + 1, brow, wave, lane);
}

template <int NB, int RB, bool LONE = false>
__global__ __launch_bounds__(kChainMaxThreads) __attribute__((amdgpu_waves_per_eu(LONE ? 2 : EdgeTile<NB, RB>::waves_per_eu)))
void k_edge_bwd(ChainBwdArgs a) {
  constexpr int D = NB * 16, W = mask_words<NB>();
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4;
  const int cw = int(blockDim.x >> 6) - a.nload, tile_rows = 16 * RB * cw;   // compute waves of this launch (launcher's choice); the last wave(s) load
  if (wave >= cw) {  // loader wave (uniform branch)
    loader_dispatch<NB>(a.nload, wave - cw, a.wseq, a.nseq, lds, lane, a.ntiles, a.nring);
    return;
  }
  const float rcpE = 1.f / float(a.E);
  auto fetch_targets = [&](int tile, int64_t (&node)[RB]) {   // node row (b * N + dst) of this lane's rows, one tile ahead
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int64_t r = int64_t(tile) * tile_rows + wave * (16 * RB) + rb * 16 + (lane & 15);
      const EdgeRef e = edge_ref(unsigned(r < a.R ? r : 0), unsigned(a.E), rcpE);
      node[rb] = int64_t(e.b) * a.N + a.dst[e.q];
    }
  };
  int64_t nnode[RB];
  fetch_targets(blockIdx.x, nnode);
  Slot slot{0, a.nring};
  float4* const ring = lds + Ring<NB>::PRE4;
  unsigned* brow = bound_row<NB>(lds, wave, lane, any_slot(a.gmax));   // this wave's running magnitude bounds
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int64_t row[RB], rowc[RB];
    PairOff off[RB];
    unsigned moff[RB];
    f32x4 g[RB][NB], acc[RB][NB];
    float rs[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {   // autograd of scatter_sum: gather the node gradient by target; y, rstd of the row
      row[rb] = int64_t(tile) * tile_rows + wave * (16 * RB) + rb * 16 + (lane & 15);
      rowc[rb] = row[rb] < a.R ? row[rb] : 0;
      off[rb] = pair_offsets<NB>(wave * (16 * RB) + rb * 16 + (lane & 15), lane);   // within the tile
      moff[rb] = 0;
      load_rows<NB>(g[rb], a.dy + nnode[rb] * D, lg);
      load_rows<NB>(acc[rb], a.yln + rowc[rb] * D, lg);
      rs[rb] = a.rstd[rowc[rb]];
    }
    if (tile + int(gridDim.x) < a.ntiles) fetch_targets(tile + gridDim.x, nnode);   // uniform; lands under the stages
    __builtin_amdgcn_sched_barrier(0);   // all loads in flight before the first use
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {   // LayerNorm backward (no affine): dz = rstd * (dy - mean(dy) - y * mean(dy * y))
      const float m1 = row_sum<NB>(g[rb]) * (1.f / D);
      float s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s2 = fmaf(g[rb][t][r], acc[rb][t][r], s2);
      s2 = group_sum(s2);
      const float m2 = s2 * (1.f / D);
#pragma unroll
      for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[rb][t][r] = rs[rb] * (g[rb][t][r] - m1 - acc[rb][t][r] * m2);
    }
    float* pending = a.gstore[0];   // uniform, non-null (launcher): the gradient entering a stage is stored inside it
    for (int k = 0; k < a.nstage; ++k) {
      unsigned mbits[RB][W];   // ReLU sign bits of the activation that masks this stage's output, loaded ahead of the stage
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int w = 0; w < W; ++w)
          mbits[rb][w] = reinterpret_cast<const unsigned*>(a.mask[k] + pad_rows(a.R) * D)[rowc[rb] * (4 * W) + lg * W + w];
#ifdef BSMS_EXPERIMENTS   // ablation bound of the fused dataflow (profiles/r05_fusion_bound.txt): the layer gradients of every tile land on tile 0
      float* const gtile = pending + int64_t(a.ablate ? 0 : tile) * (tile_rows * D);
#else
      float* const gtile = pending + int64_t(tile) * (tile_rows * D);
#endif
      stage_rb<NB, RB, 1, false, true, 1, LONE>(acc, g, ring, slot, lane, gtile, nullptr, off, moff, brow, k);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {   // bit -> all-ones / zero mask (one v_bfe_i32), then one and
            const int keep = __builtin_amdgcn_sbfe((int)mbits[rb][(4 * t + r) >> 5], (4 * t + r) & 31, 1);
            g[rb][t][r] = __uint_as_float(__float_as_uint(acc[rb][t][r]) & (unsigned)keep);
          }
      pending = a.gstore[k + 1];
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {  // gE[0]: read next by the scatter kernel, plain stores (stay in L2 / the memory-side cache)
      if (a.gmax[a.nstage]) note_amax(brow, a.nstage, row_amax<NB>(g[rb]), lane);   // uniform
      store_rows<NB, false>(g[rb], pending, row[rb] < a.R ? row[rb] * D : -1, lg);
    }
  }
  flush_bounds(a.gmax, kMaxStages + 1, brow, wave, lane);
}

// Workgroups of 5 waves the chip keeps resident per CU at each width.  NOT the occupancy API's answer: the SPI
// accounts a 5-wave workgroup like an 8-wave one (census: 320 threads x 120 VGPRs -> 2 per CU where the API says 3;
// profiles/census).  A grid larger than the residency would only queue, a smaller one idles slots.
template <int NB>
constexpr int resident_per_cu() { return NB <= 4 ? 4 : NB == 8 ? 2 : 1; }

// Compute waves per workgroup of a generic chain launch: 4 (64-row tiles), or up to 7 when that makes the launch fit ONE
// round of resident workgroups.  A workgroup streams the whole weight set once per tile (~23 us for the node MLP: the
// LDS-DMA rate of a CU), so a second, nearly empty round costs as much as the first: the level-0 node launches of the
// airfoil step (41 864 rows = 655 tiles of 64 on 512 slots) run as 437 tiles of 96 rows instead.  The SPI accounts a
// 5-wave workgroup like an 8-wave one anyway (resident_per_cu), so the extra waves use slots that were empty.
template <int NB>
int chain_compute_waves(int64_t R) {
  const int cus = device_cu_count();
  const int64_t slots = int64_t(cus) * resident_per_cu<NB>();
  if (NB < 8 || ceil_div(R, kTileRows) <= slots) return kComputeWaves;
  const int64_t need = ceil_div(R, slots * 16);   // waves per workgroup for one round
  return need <= 7 ? (int)need : kComputeWaves;
}

template <int NB>
unsigned persistent_grid(int64_t ntiles) {
  const int cus = device_cu_count();
  return (unsigned)std::min<int64_t>(ntiles, int64_t(cus) * resident_per_cu<NB>());
}


inline int device_cus() { return device_cu_count(); }   // common.h: cached per device

// Launch shape knobs.  Production values are the defaults; experiment builds read them from the environment for
// same-box sweeps (BSMS_EDGE_CW, BSMS_EDGE_NL, BSMS_CHAIN_NL, BSMS_EDGE_CW16, BSMS_EDGE_NL16).
inline int knob(const char* name, int dflt) {
#ifdef BSMS_EXPERIMENTS
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

// experiment builds (same-box A/B, profiles/edge_prof.sh): BSMS_EDGE_RB = 0 keeps the edge MLP on k_chain_fwd / k_chain_bwd,
// 1 / 2 force the number of row blocks per wave
inline int edge_rb_mode() {
#ifdef BSMS_EXPERIMENTS
  static const int m = [] { const char* e = getenv("BSMS_EDGE_RB"); return e ? atoi(e) : -1; }();
  return m;
#else
  return -1;
#endif
}

// Compute waves per workgroup of an edge launch: 7 (+ the loader: 8 waves, 112 x RB rows per tile).  A tile streams the
// whole weight set of the MLP through the loader's LDS-DMA (one wave delivers a 1 KB piece per 60-185 cycles,
// profiles/census/ldsdma_rate.hip), and since the fp32 products take three MFMAs per fragment pair that stream, not the
// matrix pipe, paces a stage: 7 compute waves spread it over 1.75x the rows of 4 (same-box +2.5 % steps/s at D = 128 with
// the fp16 x 2 arithmetic; +4.3 % at D = 256 already with the bf16 x 3 one; profiles/r02_edge_levels.md, r03).
template <int NB>
int edge_compute_waves() {
  static const int cw = NB >= 16 ? knob("BSMS_EDGE_CW16", 7) : knob("BSMS_EDGE_CW", 7);
  return cw;
}
// Compute waves of ONE edge launch.  A persistent workgroup runs ceil(tiles / grid) tiles one after the other and a tile
// costs t0 + t1 * (rows per wave-set): the fixed part is the weight stream of the whole MLP, the rest scales with the rows.
// Seven waves minimise the fixed part per row, but a launch whose last round is nearly empty pays a whole tile for it:
// cylinder level 0 (90 112 rows, 512 slots) runs 2 rounds with 7, 6 waves and 3 with 5, 4 -- 6 waves win by the shorter
// tile (same-box sweep: 4 / 5 / 6 / 7 waves = 400.1 / 392.6 / 404.4 / 397.2 steps/s, exactly this model's order).  The
// slope t1 / t0 = 0.53 per wave with one row block per wave comes from the airfoil sweep (7 against 4 waves: +2.5 %).
template <int NB, int RB>
int pick_edge_waves(int64_t R) {
  const int fixed = edge_compute_waves<NB>();
#ifdef BSMS_EXPERIMENTS
  if (getenv(NB >= 16 ? "BSMS_EDGE_CW16" : "BSMS_EDGE_CW") || getenv("BSMS_EDGE_CW_FIXED")) return fixed;
#endif
  const double slope = (NB >= 16 ? 0.25 : 0.53) * RB;
  const int64_t slots = int64_t(device_cus()) * EdgeTile<NB, RB>::resident;
  auto cost_of = [&](int cw) {
    const int64_t tiles = ceil_div(R, int64_t(16) * RB * cw);
    const int64_t per = ceil_div(tiles, std::min<int64_t>(tiles, slots));
    return double(per) * (1.0 + slope * cw);
  };
  int best = fixed;
  double best_cost = cost_of(fixed) * 0.92;   // the model is coarse: leave the measured default unless it predicts a clear gain
  for (int cw = 6; cw >= 4; --cw) {
    const double cost = cost_of(cw);
    if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = cw; }
  }
  return best;
}
template <int NB>
int edge_loader_waves() {
  static const int nl = NB >= 16 ? knob("BSMS_EDGE_NL16", 1) : knob("BSMS_EDGE_NL", 1);
  return nl;
}
inline int chain_loader_waves() {
  static const int nl = knob("BSMS_CHAIN_NL", 1);
  return nl;
}

// Weight stream of a launch: loader waves and ring depth (Ring).  A launch that fits ONE round of workgroups (at most
// one workgroup per CU: the coarse mesh levels, most node-level launches) has the CU's whole LDS and nothing to overlap
// its chunk latency with: deep ring (up to 6 slots = 5 chunks in flight) fed by two loader waves.  Anything larger keeps
// 3 slots so that two workgroups share a CU.  Limits: compute + loader waves <= 8; (nr - 2) x pieces per loader <= 63
// (vmcnt field); ring + side tables <= 160 KB.
template <int NB, int PL = kPL>
int max_ring() { return (int)std::min<size_t>(6, (size_t(160) * 1024 - Ring<NB, PL>::PRE_FLOATS * sizeof(float)) / (Ring<NB, PL>::CHF * sizeof(float))); }
template <int NB, int PL = kPL>
void pick_stream(int64_t ntiles, int cw, int nload_default, int& nload, int& nring) {
  static const int deep = knob("BSMS_RING_DEEP", 6), lone_nl = knob("BSMS_LONE_NL", 2), shared = knob("BSMS_RING", 3);
  nload = std::max(1, std::min(nload_default, 8 - cw));
  nring = std::min(shared, max_ring<NB, PL>());
  if (ntiles <= device_cus()) {
    nload = std::max(nload, std::min(lone_nl, 8 - cw));
    nring = std::min(deep, max_ring<NB, PL>());
  }
  const int mine = (Ring<NB, PL>::PER + nload - 1) / nload;
  while (nring > 3 && (nring - 2) * mine > 63) --nring;
}

template <int NB, int RB, int SAVE>
int launch_edge_fwd_t(ChainFwdArgs& a, hipStream_t s) {
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_edge_fwd<NB, RB, SAVE>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
  BSMS_REQUIRE(attr == hipSuccess, BSMS_E_HIP, "edge_fwd: cannot reserve %zu bytes of LDS", Ring<NB>::lds_bytes(max_ring<NB>()));
  const int cw = pick_edge_waves<NB, RB>(a.R);
  a.ntiles = (int)ceil_div(a.R, 16 * RB * cw);
  pick_stream<NB>(a.ntiles, cw, edge_loader_waves<NB>(), a.nload, a.nring);
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, int64_t(device_cus()) * EdgeTile<NB, RB>::resident);
  static const int lone = knob("BSMS_EDGE_LONE", 1);
  if (lone && a.ntiles <= device_cus() && !a.timing) {   // one round of workgroups: the variant that passes the chunk barrier early (stage_rb<LONE>)
    static DynLdsAttr lattr_dev;
  const hipError_t lattr = lattr_dev.ensure(reinterpret_cast<const void*>(&k_edge_fwd<NB, RB, SAVE, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
    BSMS_REQUIRE(lattr == hipSuccess, BSMS_E_HIP, "edge_fwd: cannot reserve LDS (single-round build)");
    hipLaunchKernelGGL((k_edge_fwd<NB, RB, SAVE, true>), dim3(grid), dim3((cw + a.nload) * 64), Ring<NB>::lds_bytes(a.nring), s, a);
    BSMS_LAUNCH_CHECK();
    return BSMS_OK;
  }
  hipLaunchKernelGGL((k_edge_fwd<NB, RB, SAVE>), dim3(grid), dim3((cw + a.nload) * 64), Ring<NB>::lds_bytes(a.nring), s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

// the software-pipelined edge kernels take the production configuration only; anything else stays on k_chain_fwd
template <int NB>
bool launch_edge_fwd(ChainFwdArgs& a, hipStream_t s, int& rc) {
  if (a.bf16 || a.nstage < 1 || a.store_mode != 1 || a.resid || a.resid2 || edge_rb_mode() == 0 || a.R >= (int64_t(1) << 31)) return false;
  const bool save = a.store_in != nullptr;
  for (int l = 0; l + 1 < a.nstage; ++l)
    if ((a.store[l] != nullptr) != save) return false;
  constexpr int RBIG = NB == 8 ? 2 : 1;
  // Measured per level (profiles/r02_edge_levels.md): with several tiles per workgroup two workgroups per CU of one row
  // block per wave win the forward (the random row gathers of one hide under the other's MFMA stages); a launch that
  // fits one round of workgroups is faster with two row blocks per wave, and so is the whole backward (its loads are
  // sequential or local).  Below half a round of 128-row tiles the narrow tile keeps more CUs busy.
  const int64_t cus = device_cus(), rows1 = 16 * edge_compute_waves<NB>();   // rows of a tile with one row block per wave
  bool big = RBIG == 2 && a.R >= cus * rows1 && ceil_div(a.R, rows1) <= cus * EdgeTile<NB, 1>::resident;
  if (edge_rb_mode() > 0) big = RBIG == 2 && edge_rb_mode() == 2;
#ifdef BSMS_EXPERIMENTS   // (the only reader of the pieces, experiments/efuse32.hip, is not in the product library: no instantiation there)
  if constexpr (NB == 8) {
    if (save && a.pieces) {   // the activations leave as the fp16 x 2 pieces of their own stage (valu_step SAVE == 2): efuse32.hip reads them back
      for (int l = 0; l < a.nstage; ++l)
        if (!a.store_exp[l]) return false;
      rc = big ? launch_edge_fwd_t<NB, RBIG, 2>(a, s) : launch_edge_fwd_t<NB, 1, 2>(a, s);
      return true;
    }
  }
#endif
  if (a.pieces) return false;
  if (big) rc = save ? launch_edge_fwd_t<NB, RBIG, 1>(a, s) : launch_edge_fwd_t<NB, RBIG, 0>(a, s);
  else rc = save ? launch_edge_fwd_t<NB, 1, 1>(a, s) : launch_edge_fwd_t<NB, 1, 0>(a, s);
  return true;
}

template <int NB, int RB>
int launch_edge_bwd_t(ChainBwdArgs& a, hipStream_t s) {
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_edge_bwd<NB, RB>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
  BSMS_REQUIRE(attr == hipSuccess, BSMS_E_HIP, "edge_bwd: cannot reserve %zu bytes of LDS", Ring<NB>::lds_bytes(max_ring<NB>()));
  const int cw = pick_edge_waves<NB, RB>(a.R);
  a.ntiles = (int)ceil_div(a.R, 16 * RB * cw);
  pick_stream<NB>(a.ntiles, cw, edge_loader_waves<NB>(), a.nload, a.nring);
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, int64_t(device_cus()) * EdgeTile<NB, RB>::resident);
  static const int lone = knob("BSMS_EDGE_LONE", 1);
  if (lone && a.ntiles <= device_cus()) {   // see launch_edge_fwd_t
    static DynLdsAttr lattr_dev;
  const hipError_t lattr = lattr_dev.ensure(reinterpret_cast<const void*>(&k_edge_bwd<NB, RB, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
    BSMS_REQUIRE(lattr == hipSuccess, BSMS_E_HIP, "edge_bwd: cannot reserve LDS (single-round build)");
    hipLaunchKernelGGL((k_edge_bwd<NB, RB, true>), dim3(grid), dim3((cw + a.nload) * 64), Ring<NB>::lds_bytes(a.nring), s, a);
    BSMS_LAUNCH_CHECK();
    return BSMS_OK;
  }
  hipLaunchKernelGGL((k_edge_bwd<NB, RB>), dim3(grid), dim3((cw + a.nload) * 64), Ring<NB>::lds_bytes(a.nring), s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

template <int NB>
bool launch_edge_bwd(ChainBwdArgs& a, hipStream_t s, int& rc) {
  if (a.bf16 || a.nstage < 1 || a.store_mode != 1 || edge_rb_mode() == 0 || a.R >= (int64_t(1) << 31)) return false;
  for (int k = 0; k <= a.nstage; ++k)
    if (!a.gstore[k] || (k < a.nstage && !a.mask[k])) return false;
  constexpr int RBIG = NB == 8 ? 2 : 1;
  bool big = RBIG == 2 && a.R >= int64_t(device_cus()) * 16 * edge_compute_waves<NB>();   // see launch_edge_fwd
  if (edge_rb_mode() > 0) big = RBIG == 2 && edge_rb_mode() == 2;
  rc = big ? launch_edge_bwd_t<NB, RBIG>(a, s) : launch_edge_bwd_t<NB, 1>(a, s);
  return true;
}

// Compute waves per workgroup of the bf16 edge chains (generic kernels, 16 rows per wave).  Round 4, same-box with the
// experiment build (profiles/r04_bfcw.sh): 7 waves against 4 -- airfoil batch 8 bf16 222.4 -> 230.1, bf16_nodes 231.5 -> 241.5,
// surface B=2 bf16 105.2 -> 110.9 / bf16_nodes 111.6 -> 118.0 steps/s (a workgroup streams the weights once per tile: 112 rows
// per pass instead of 64); a launch that fits one round of 64-row tiles keeps 4 (batch 1: 614 against 597 steps/s with 7).
template <int NB>
int bf_edge_waves(int64_t R) {
  static const int forced = knob("BSMS_BFEDGE_CW", 0);
  if (forced > 0) return forced;
  return R > int64_t(device_cus()) * resident_per_cu<NB>() * 16 * kComputeWaves ? 7 : kComputeWaves;
}

template <int NB, int IN, int OUT>
int launch_fwd_t(const ChainFwdArgs& a0, hipStream_t s) {
  ChainFwdArgs a = a0;
  a.nseq = 0;
  for (int l = 0; l < a.nstage; ++l) {  // the loader follows exactly the compute waves' stage order
    a.wseq[a.nseq++] = a.wp[l];
    if (IN == IN_ROWS2 && l == 0) a.wseq[a.nseq++] = a.wp0b;
  }
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_chain_fwd<NB, IN, OUT>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
  BSMS_REQUIRE(attr == hipSuccess, BSMS_E_HIP, "chain_fwd: cannot reserve %zu bytes of LDS", Ring<NB>::lds_bytes(max_ring<NB>()));
  if constexpr ((NB == 8 || NB == 16) && IN == IN_EDGE && OUT == OUT_LN) {
    int rc = BSMS_OK;
    if (launch_edge_fwd<NB>(a, s, rc)) return rc;
  }
  BSMS_REQUIRE(!a.pieces, BSMS_E_UNSUPPORTED, "chain_fwd: only the pipelined edge kernel at D = 128 saves fp16 x 2 pieces (R = %lld, stages %d)", (long long)a.R, a.nstage);
  if constexpr (NB == 8 && (IN == IN_ROWS || IN == IN_ROWS2 || IN == IN_SMALL)) {   // small launches: the feature-split kernel
    static const int fs_rows = knob("BSMS_FS_ROWS", kFsMaxRows);
    if (!a.bf16 && a.R <= fs_rows && a.nseq >= 1 && a.nseq <= kMaxStages + 1) {
      hipLaunchKernelGGL((k_fs_fwd<IN, OUT>), dim3((unsigned)ceil_div(a.R, 16)), dim3(256), 0, s, a);
      BSMS_LAUNCH_CHECK();
      return BSMS_OK;
    }
  }
  int cw = (IN == IN_EDGE) ? bf_edge_waves<NB>(a.R) : chain_compute_waves<NB>(a.R);
  a.ntiles = (int)ceil_div(a.R, 16 * cw);
  // One Linear over [x, x2] added into y (the input gradient through the two edge projections, gmp.hip): three more dependent row
  // loads per tile than a plain chain and only two packs of MFMAs to hide them under.  The single-round build keeps x2 in registers
  // (one round trip instead of three) -- so this launch always takes it, one 7-wave workgroup per CU striding over the tiles
  // (round 6, profiles/r06_rows2.txt).
  bool rows2_lone = false;
  if constexpr (NB == 8 && IN == IN_ROWS2 && OUT == OUT_PLAIN) {
    static const int on = knob("BSMS_ROWS2_LONE", 1);
    if (on && !a.bf16 && a.nstage == 1 && a.ntiles > device_cus()) {
      rows2_lone = true;
      cw = 7;
      a.ntiles = (int)ceil_div(a.R, 16 * cw);
    }
  }
  const int64_t stream_tiles = rows2_lone ? std::min<int64_t>(a.ntiles, device_cus()) : a.ntiles;   // ring depth / loaders of a one-workgroup-per-CU launch
  if (a.bf16) pick_stream<NB, 1>(stream_tiles, cw, chain_loader_waves(), a.nload, a.nring);
  else pick_stream<NB>(stream_tiles, cw, chain_loader_waves(), a.nload, a.nring);
  const dim3 threads((cw + a.nload) * 64);
  // the bf16 precision's chunks are one plane: its ring may be deeper than the fp32 one at the same D, size it as what it is
  const size_t lds = a.bf16 ? Ring<NB, 1>::lds_bytes(a.nring) : Ring<NB>::lds_bytes(a.nring);
  bool launched = false;
  if constexpr (NB == 8 && IN == IN_EDGE) {   // the only instantiation with stamps
    if (a.timing) {
      static DynLdsAttr tattr_dev;
  const hipError_t tattr = tattr_dev.ensure(reinterpret_cast<const void*>(&k_chain_fwd<NB, IN, OUT, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
      BSMS_REQUIRE(tattr == hipSuccess, BSMS_E_HIP, "chain_fwd: cannot reserve LDS (timing build)");
      hipLaunchKernelGGL((k_chain_fwd<NB, IN, OUT, true>), dim3(persistent_grid<NB>(a.ntiles)), threads, lds, s, a);
      launched = true;
    }
  }
#ifdef BSMS_EXPERIMENTS
  if constexpr (NB == 8 && IN == IN_ROWS2 && OUT == OUT_LN) {   // experiments: phase stamps of a single-round node chain (profiles/lone_timeline.py)
    if (a.timing && a.ntiles <= device_cus()) {
      static DynLdsAttr tattr_dev;
  const hipError_t tattr = tattr_dev.ensure(reinterpret_cast<const void*>(&k_chain_fwd<NB, IN, OUT, true, false, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
      BSMS_REQUIRE(tattr == hipSuccess, BSMS_E_HIP, "chain_fwd: cannot reserve LDS (timing build)");
      hipLaunchKernelGGL((k_chain_fwd<NB, IN, OUT, true, false, true>), dim3(persistent_grid<NB>(a.ntiles)), threads, lds, s, a);
      launched = true;
    }
  }
#endif
  if constexpr ((NB == 8 || NB == 16) && (IN == IN_EDGE || IN == IN_ROWS2) && OUT == OUT_LN) {   // the bf16 arithmetic: edge MLP (BSMS_BF16), node MLP (BSMS_BF16_NODES)
    if (a.bf16 && !launched) {
      static DynLdsAttr battr_dev;
  const hipError_t battr = battr_dev.ensure(reinterpret_cast<const void*>(&k_chain_fwd<NB, IN, OUT, false, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
      BSMS_REQUIRE(battr == hipSuccess, BSMS_E_HIP, "chain_fwd: cannot reserve LDS (bf16 build)");
      hipLaunchKernelGGL((k_chain_fwd<NB, IN, OUT, false, true>), dim3(persistent_grid<NB>(a.ntiles)), threads, lds, s, a);
      launched = true;
    }
  }
  BSMS_REQUIRE(launched || !a.bf16, BSMS_E_UNSUPPORTED, "chain_fwd: bf16 precision is built for the edge and node MLPs at D = 128 / 256 only");
  if constexpr (NB >= 8) {   // one round of workgroups = a single wave per SIMD: the variant that prefetches its fragments (mfma_stage)
    if (!launched && (a.ntiles <= device_cus() || rows2_lone)) {
      static DynLdsAttr lattr_dev;
  const hipError_t lattr = lattr_dev.ensure(reinterpret_cast<const void*>(&k_chain_fwd<NB, IN, OUT, false, false, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
      BSMS_REQUIRE(lattr == hipSuccess, BSMS_E_HIP, "chain_fwd: cannot reserve LDS (single-round build)");
      const unsigned grid = rows2_lone ? (unsigned)std::min<int64_t>(a.ntiles, device_cus()) : persistent_grid<NB>(a.ntiles);
      hipLaunchKernelGGL((k_chain_fwd<NB, IN, OUT, false, false, true>), dim3(grid), threads, lds, s, a);
      launched = true;
    }
  }
  if (!launched)
    hipLaunchKernelGGL((k_chain_fwd<NB, IN, OUT>), dim3(persistent_grid<NB>(a.ntiles)), threads, lds, s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
// Only the combinations the path uses are instantiated (each is a large unrolled kernel).
template <int NB>
int launch_fwd_n(int in_mode, int out_mode, const ChainFwdArgs& a, hipStream_t s) {
#define BSMS_FWD(I, O) \
  if (in_mode == I && out_mode == O) return launch_fwd_t<NB, I, O>(a, s)
  BSMS_FWD(IN_ROWS, OUT_PLAIN);   // x W^T
  BSMS_FWD(IN_ROWS, OUT_PLAIN2);  // the two node pre-projections of the edge MLP
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
  static DynLdsAttr attr_dev;
  const hipError_t attr = attr_dev.ensure(reinterpret_cast<const void*>(&k_chain_bwd<NB, GIN, FIRST>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
  BSMS_REQUIRE(attr == hipSuccess, BSMS_E_HIP, "chain_bwd: cannot reserve %zu bytes of LDS", Ring<NB>::lds_bytes(max_ring<NB>()));
  if constexpr (NB == 8 && (GIN == G_ROWS_LN || GIN == G_SMALL)) {   // small launches: the feature-split kernel (see launch_fwd_t)
    static const int fs_rows = knob("BSMS_FS_ROWS_BWD", kFsMaxRowsBwd);
    if (!a.bf16 && a.R <= fs_rows && a.nseq >= 1) {
      hipLaunchKernelGGL((k_fs_bwd<GIN, FIRST>), dim3((unsigned)ceil_div(a.R, 16)), dim3(256), 0, s, a);
      BSMS_LAUNCH_CHECK();
      return BSMS_OK;
    }
  }
  const int cw = (GIN == G_EDGE_LN) ? bf_edge_waves<NB>(a.R) : chain_compute_waves<NB>(a.R);
  a.ntiles = (int)ceil_div(a.R, 16 * cw);
  if (a.bf16) pick_stream<NB, 1>(a.ntiles, cw, chain_loader_waves(), a.nload, a.nring);
  else pick_stream<NB>(a.ntiles, cw, chain_loader_waves(), a.nload, a.nring);
  const dim3 threads((cw + a.nload) * 64);
  const size_t lds = a.bf16 ? Ring<NB, 1>::lds_bytes(a.nring) : Ring<NB>::lds_bytes(a.nring);   // see launch_fwd_t
  if constexpr ((NB == 8 || NB == 16) && GIN == G_EDGE_LN && FIRST == F_NONE) {
    int rc = BSMS_OK;
    if (launch_edge_bwd<NB>(a, s, rc)) return rc;
    a.ntiles = (int)ceil_div(a.R, 16 * cw);
    if (a.bf16) {
      static DynLdsAttr battr_dev;
  const hipError_t battr = battr_dev.ensure(reinterpret_cast<const void*>(&k_chain_bwd<NB, GIN, FIRST, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
      BSMS_REQUIRE(battr == hipSuccess, BSMS_E_HIP, "chain_bwd: cannot reserve LDS (bf16 build)");
      hipLaunchKernelGGL((k_chain_bwd<NB, GIN, FIRST, true>), dim3(persistent_grid<NB>(a.ntiles)), threads, lds, s, a);
      BSMS_LAUNCH_CHECK();
      return BSMS_OK;
    }
  }
  if constexpr ((NB == 8 || NB == 16) && GIN == G_ROWS_LN && FIRST == F_HEADS2) {   // node MLP of BSMS_BF16_NODES
    if (a.bf16) {
      static DynLdsAttr nattr_dev;
  const hipError_t nattr = nattr_dev.ensure(reinterpret_cast<const void*>(&k_chain_bwd<NB, GIN, FIRST, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
      BSMS_REQUIRE(nattr == hipSuccess, BSMS_E_HIP, "chain_bwd: cannot reserve LDS (bf16 node build)");
      hipLaunchKernelGGL((k_chain_bwd<NB, GIN, FIRST, true>), dim3(persistent_grid<NB>(a.ntiles)), threads, lds, s, a);
      BSMS_LAUNCH_CHECK();
      return BSMS_OK;
    }
  }
  BSMS_REQUIRE(!a.bf16, BSMS_E_UNSUPPORTED, "chain_bwd: bf16 precision is built for the edge and node MLPs at D = 128 / 256 only");
  if constexpr (NB >= 8) {   // see launch_fwd_t
    if (a.ntiles <= device_cus()) {
      static DynLdsAttr lattr_dev;
  const hipError_t lattr = lattr_dev.ensure(reinterpret_cast<const void*>(&k_chain_bwd<NB, GIN, FIRST, false, true>), (int)Ring<NB>::lds_bytes(max_ring<NB>()));
      BSMS_REQUIRE(lattr == hipSuccess, BSMS_E_HIP, "chain_bwd: cannot reserve LDS (single-round build)");
      hipLaunchKernelGGL((k_chain_bwd<NB, GIN, FIRST, false, true>), dim3(persistent_grid<NB>(a.ntiles)), threads, lds, s, a);
      BSMS_LAUNCH_CHECK();
      return BSMS_OK;
    }
  }
  hipLaunchKernelGGL((k_chain_bwd<NB, GIN, FIRST>), dim3(persistent_grid<NB>(a.ntiles)), threads, lds, s, a);
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

// experiments only (not in bsms_hip.h): what residency does the runtime compute for the D = 128 edge chains?
#ifdef BSMS_EXPERIMENTS
extern "C" int bsms_debug_occupancy(int* fwd_blocks_per_cu, int* bwd_blocks_per_cu) {
  hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(fwd_blocks_per_cu, k_chain_fwd<8, IN_EDGE, OUT_LN>,
                                                                kChainThreads, Ring<8>::lds_bytes(3));
  hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(bwd_blocks_per_cu, k_chain_bwd<8, G_EDGE_LN, F_NONE>,
                                                                kChainThreads, Ring<8>::lds_bytes(3));
  return (e1 == hipSuccess && e2 == hipSuccess) ? 0 : -4;
}
#endif

namespace bsms {

int launch_prepack(const PackTable& t, hipStream_t s) {
  if (t.n == 0) return BSMS_OK;
  int biggest = 0;
  for (int i = 0; i < t.n; ++i) biggest = biggest > t.d[i].N * t.d[i].K ? biggest : t.d[i].N * t.d[i].K;
  static const int fused = knob("BSMS_PACK_FUSED", 1);
  if (fused) {
    hipLaunchKernelGGL(k_prepack_fused, dim3((unsigned)std::min<int64_t>(ceil_div(biggest, 4096), 16), t.n), dim3(1024), 0, s, t);
    BSMS_LAUNCH_CHECK();
    return BSMS_OK;
  }
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(biggest, 256), 64);
  hipLaunchKernelGGL(k_pack_scale, dim3(t.n), dim3(1024), 0, s, t);
  BSMS_LAUNCH_CHECK();
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
