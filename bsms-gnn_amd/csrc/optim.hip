// Fused optimizer step on the flat parameter / gradient buffers: global-norm gradient clipping + AdamW.
// Replaces, for one training step, torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step over ~180 parameter
// tensors (trainer/trainer.py:150-152) by two launches over one contiguous 1.9 M-element array.  HBM bound:
// 4 streams read + 3 written per element.  The update follows torch.optim.AdamW (decoupled weight decay,
// bias-corrected moments) operation for operation.
#include <algorithm>
#include <cmath>

#include "common.h"

using namespace bsms;

namespace {
constexpr int NPART = 256;

__global__ __launch_bounds__(256) void k_sumsq_partials(const float* g, int64_t n, float* part) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(NPART) * 256) s = fmaf(g[i], g[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  int64_t n;
  float lr, beta1, beta2, eps, wd, bc1, sqrt_bc2, max_norm;
  const float* part;   // NPART partial sums of g^2 (null: no clipping)
  float* norm_out;     // optional device scalar: total gradient norm before clipping
};

__global__ __launch_bounds__(256) void k_adamw(AdamArgs a) {
  __shared__ float red[256];
  float clip = 1.f;
  if (a.part) {  // every block re-reduces the 256 partials in the same fixed order: identical value everywhere
    red[threadIdx.x] = a.part[threadIdx.x];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    const float total = sqrtf(red[0]);
    if (a.norm_out && blockIdx.x == 0 && threadIdx.x == 0) *a.norm_out = total;
    if (a.max_norm > 0.f) clip = fminf(a.max_norm / (total + 1e-6f), 1.f);   // clip_grad_norm_ semantics
  }
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < a.n; i += int64_t(gridDim.x) * 256) {
    const float g = a.g[i] * clip;
    float p = a.p[i] * (1.f - a.lr * a.wd);                   // p.mul_(1 - lr * weight_decay)
    const float m = a.m[i] + (g - a.m[i]) * (1.f - a.beta1);  // exp_avg.lerp_(grad, 1 - beta1)
    const float v = a.v[i] * a.beta2 + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.sqrt_bc2 + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);                        // p.addcdiv_(exp_avg, denom, value=-lr/bc1)
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
  }
}
}  // namespace

extern "C" size_t bsms_adamw_work_bytes(void) { return NPART * sizeof(float); }

extern "C" int bsms_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int64_t step,
                               float max_grad_norm, float* grad_norm_out, void* work, bsms_stream_t stream) {
  BSMS_REQUIRE((params && grads && exp_avg && exp_avg_sq) || n == 0, BSMS_E_INVALID_ARG, "adamw_step: null argument");
  BSMS_REQUIRE(n >= 0 && step >= 1, BSMS_E_SHAPE, "adamw_step: n=%lld step=%lld (step counts from 1)", (long long)n, (long long)step);
  const bool need_norm = max_grad_norm > 0.f || grad_norm_out != nullptr;
  BSMS_REQUIRE(!need_norm || work, BSMS_E_INVALID_ARG, "adamw_step: work buffer needed for the gradient norm");
  if (n == 0) return BSMS_OK;
  hipStream_t s = as_stream(stream);
  AdamArgs a{};
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.bc1 = 1.f - (float)std::pow((double)beta1, (double)step);
  a.sqrt_bc2 = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
  a.max_norm = max_grad_norm; a.norm_out = grad_norm_out;
  if (need_norm) {
    a.part = reinterpret_cast<float*>(work);
    hipLaunchKernelGGL(k_sumsq_partials, dim3(NPART), dim3(256), 0, s, grads, n, reinterpret_cast<float*>(work));
    BSMS_LAUNCH_CHECK();
  }
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n, 256), 2048);
  hipLaunchKernelGGL(k_adamw, dim3(grid), dim3(256), 0, s, a);
  BSMS_LAUNCH_CHECK();
  return BSMS_OK;
}
