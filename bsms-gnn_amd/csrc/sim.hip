// Model-level glue of BSMS_Simulator (models/model.py:127-164) and the masked-RMSE loss (trainer/trainer.py:96-97) as
// three small fused kernels.  In the reference (and in the autograd mirror, bsms-gnn_amd/model.py) this is ~40
// element-wise / reduction launches per training step on [B,N,<=4] tensors -- ~0.35 ms of a 7 ms step on the GPU and
// most of the host time that is not the engine's own.  The arithmetic is kept op for op:
//   normalise    float( (double(x) - mean) / std ),  std = max(nan_to_num(sqrt(E[x^2] - mean^2)), eps)   (fp64, normalizer.py:40-52,88-90)
//   de-normalise float( double(y) * std + mean )                                                        (normalizer.py:80-83)
//   integrate    pred = state + delta * mask                                                            (model.py:160-163)
//   loss         sqrt( sum(se * mask) / sum(mask) / C )                                                 (trainer.py:96-97)
// This file is compiled with -ffp-contract=off (build.py): a fused multiply-add would change the fp64 roundings.
#include "common.h"

#pragma clang fp contract(off)

using namespace bsms;

namespace {

constexpr int kMaxC = 8;   // state / target channels handled in registers

__device__ __forceinline__ double std_eps(double mean, double meansq, double eps) {
  double s = sqrt(meansq - mean * mean);       // torch: sqrt(E2 - mean ** 2); pow(x, 2) is x * x
  if (isnan(s)) s = 0.0;                       // nan_to_num
  else if (isinf(s)) s = s > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
  return s > eps ? s : eps;                    // torch.max(std, eps)
}

// node_in [R, C+p+1] = [state(C) | mesh_pos(p) | node_type(1)]  ->  norm_in [R, C+1] = normalised [state | type], pos [R,p]
__global__ __launch_bounds__(256) void k_sim_prologue(const float* node_in, int64_t R, int C, int p, const double* mean,
                                                      const double* meansq, const double* eps, float* norm_in, float* pos) {
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (r >= R) return;
  const int W = C + p + 1;
  const float* row = node_in + r * W;
  const double e = *eps;
  for (int c = 0; c <= C; ++c) {
    const float x = c < C ? row[c] : row[W - 1];
    norm_in[r * (C + 1) + c] = float((double(x) - mean[c]) / std_eps(mean[c], meansq[c], e));
  }
  for (int c = 0; c < p; ++c) pos[r * p + c] = row[C + c];
}

// de-normalise, mask, integrate; optional loss partial sums; optional next rollout input
__global__ __launch_bounds__(256) void k_sim_epilogue(const float* norm_pred, const float* node_in, const float* mask,
                                                      const float* target, int64_t R, int C, int p, const double* mean,
                                                      const double* meansq, const double* eps, float* pred, float* next_in,
                                                      const float* ic, float2* partials) {
  __shared__ float2 red[256];
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  float s_se = 0.f, s_m = 0.f;
  if (r < R) {
    const int W = C + p + 1;
    const double e = *eps;
    const float m = mask[r];
    float pr[kMaxC], tail[kMaxC + 1];
    for (int c = 0; c < C; ++c) {
      const float delta = float(double(norm_pred[r * C + c]) * std_eps(mean[c], meansq[c], e) + mean[c]);
      const float dm = delta * m;
      pr[c] = node_in[r * W + c] + dm;
    }
    if (next_in)
      for (int c = 0; c <= p; ++c) tail[c] = node_in[r * W + C + c];    // read before a possibly aliasing write
    for (int c = 0; c < C; ++c) pred[r * C + c] = pr[c];
    if (target) {
      for (int c = 0; c < C; ++c) {
        const float d = pr[c] - target[r * C + c];
        const float se = d * d;
        s_se += se * m;
      }
      s_m = m;
    }
    if (next_in) {   // rollout_utils.py:57-62: next = cat[pred, mesh_pos | type]; Dirichlet nodes (mask == 0) keep the IC
      for (int c = 0; c < C; ++c) next_in[r * W + c] = (m == 0.f) ? ic[r * W + c] : pr[c];
      for (int c = 0; c <= p; ++c) next_in[r * W + C + c] = (m == 0.f) ? ic[r * W + C + c] : tail[c];
    }
  }
  if (partials) {
    red[threadIdx.x] = make_float2(s_se, s_m);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {   // fixed tree: deterministic
      if (threadIdx.x < s) {
        red[threadIdx.x].x += red[threadIdx.x + s].x;
        red[threadIdx.x].y += red[threadIdx.x + s].y;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
  }
}

__global__ __launch_bounds__(256) void k_sim_sum_partials(const float2* partials, int n, float* sums) {
  __shared__ double2 red[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { a += partials[i].x; b += partials[i].y; }
  red[threadIdx.x] = make_double2(a, b);
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[threadIdx.x].x += red[threadIdx.x + s].x;
      red[threadIdx.x].y += red[threadIdx.x + s].y;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[0] = float(red[0].x); sums[1] = float(red[0].y); }
}

// loss = sqrt(S / M / C) from the (possibly all-reduced) sums; grad w.r.t. the decoder output
__global__ __launch_bounds__(256) void k_sim_loss_bwd(const float* pred, const float* target, const float* mask, int64_t R,
                                                      int C, const double* mean, const double* meansq, const double* eps,
                                                      const float* sums, float* loss_out, float* grad_norm_pred) {
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const float S = sums[0], M = sums[1];
  const float loss = sqrtf(S / M / float(C));
  if (r == 0 && loss_out) *loss_out = loss;
  if (r >= R) return;
  const double e = *eps;
  const float m = mask[r];
  const float coef = 1.f / (loss * M * float(C));           // dL/dS * 2 = 1 / (L M C)
  for (int c = 0; c < C; ++c) {
    const float gp = (pred[r * C + c] - target[r * C + c]) * m * coef;   // dL/dpred
    const float gd = gp * m;                                              // through `delta * mask`
    grad_norm_pred[r * C + c] = float(double(gd) * std_eps(mean[c], meansq[c], e));   // through the fp64 de-normalisation
  }
}

}  // namespace

extern "C" int bsms_sim_prologue(const float* node_in, int64_t R, int64_t C, int64_t p, const double* mean,
                                 const double* meansq, const double* std_eps_dev, float* norm_in, float* pos,
                                 bsms_stream_t stream) {
  BSMS_REQUIRE(R >= 0 && C >= 1 && C <= kMaxC && p >= 1 && p <= 7, BSMS_E_UNSUPPORTED, "sim_prologue: R=%lld C=%lld p=%lld",
               (long long)R, (long long)C, (long long)p);
  if (R == 0) return BSMS_OK;
  BSMS_REQUIRE(node_in && mean && meansq && std_eps_dev && norm_in && pos, BSMS_E_INVALID_ARG, "sim_prologue: null argument");
  hipLaunchKernelGGL(k_sim_prologue, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, as_stream(stream), node_in, R, (int)C,
                     (int)p, mean, meansq, std_eps_dev, norm_in, pos);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}

extern "C" size_t bsms_sim_work_bytes(int64_t R) { return size_t(ceil_div(std::max<int64_t>(R, 1), 256)) * sizeof(float2) + 256; }

extern "C" int bsms_sim_epilogue(const float* norm_pred, const float* node_in, const float* mask, const float* target,
                                 int64_t R, int64_t C, int64_t p, const double* mean, const double* meansq,
                                 const double* std_eps_dev, float* pred, float* next_in, const float* ic, float* sums,
                                 void* work, bsms_stream_t stream) {
  BSMS_REQUIRE(R >= 0 && C >= 1 && C <= kMaxC && p >= 1 && p <= 7, BSMS_E_UNSUPPORTED, "sim_epilogue: R=%lld C=%lld p=%lld",
               (long long)R, (long long)C, (long long)p);
  BSMS_REQUIRE(!sums || (target && work), BSMS_E_INVALID_ARG, "sim_epilogue: the loss sums need target and work");
  BSMS_REQUIRE(!next_in || ic, BSMS_E_INVALID_ARG, "sim_epilogue: next_in needs the initial condition");
  hipStream_t s = as_stream(stream);
  if (R == 0) {
    if (sums) BSMS_HIP_CHECK(hipMemsetAsync(sums, 0, 2 * sizeof(float), s));
    return BSMS_OK;
  }
  BSMS_REQUIRE(norm_pred && node_in && mask && mean && meansq && std_eps_dev && pred, BSMS_E_INVALID_ARG, "sim_epilogue: null argument");
  const unsigned nb = (unsigned)ceil_div(R, 256);
  float2* partials = sums ? reinterpret_cast<float2*>(work) : nullptr;
  hipLaunchKernelGGL(k_sim_epilogue, dim3(nb), dim3(256), 0, s, norm_pred, node_in, mask, target, R, (int)C, (int)p, mean, meansq,
                     std_eps_dev, pred, next_in, ic, partials);
  BSMS_LAUNCH_CHECK();
  if (sums) {
    hipLaunchKernelGGL(k_sim_sum_partials, dim3(1), dim3(256), 0, s, (const float2*)partials, (int)nb, sums);
    BSMS_LAUNCH_CHECK();
  }
  return BSMS_OK;
}

extern "C" int bsms_sim_loss_bwd(const float* pred, const float* target, const float* mask, int64_t R, int64_t C,
                                 const double* mean, const double* meansq, const double* std_eps_dev, const float* sums,
                                 float* loss_out, float* grad_norm_pred, bsms_stream_t stream) {
  BSMS_REQUIRE(R >= 1 && C >= 1 && C <= kMaxC, BSMS_E_UNSUPPORTED, "sim_loss_bwd: R=%lld C=%lld", (long long)R, (long long)C);
  BSMS_REQUIRE(pred && target && mask && mean && meansq && std_eps_dev && sums && grad_norm_pred, BSMS_E_INVALID_ARG,
               "sim_loss_bwd: null argument");
  hipLaunchKernelGGL(k_sim_loss_bwd, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, as_stream(stream), pred, target, mask, R,
                     (int)C, mean, meansq, std_eps_dev, sums, loss_out, grad_norm_pred);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
