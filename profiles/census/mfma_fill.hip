// Experiment: how many independent single-issue instructions ("fillers") hide behind one MFMA, for the two bf16 shapes,
// one wave per SIMD (the configuration of the RB = 2 edge kernels).  A wave runs N MFMAs on 4 independent accumulators
// with F fillers after each: F VALU (v_and / v_perm mix) or F-1 VALU + 1 ds_read_b128.  Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int SHAPE, int F, bool LDSR>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
  __shared__ float4 lds[2048];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_float4(i, 1, 2, 3);
  __syncthreads();
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  f32x4 c4[4] = {};
  f32x16 c16[4] = {};
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 7 + i;
  float4 l = lds[lane];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (SHAPE == 16) c4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c4[m & 3], 0, 0, 0);
      else c16[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c16[m & 3], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if (LDSR && f == 0) { l = lds[(lane + 64 * ((m + it) & 15)) & 2047]; }
        else if (f & 1) v[(m + f) & 7] = __builtin_amdgcn_perm(v[(m + f + 1) & 7], v[(m + f) & 7], 0x07060302u);
        else v[(m + f) & 7] &= 0xffff0f0fu + f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = l.x;
  for (int i = 0; i < 4; ++i) s += c4[i][0] + c16[i][0];
  unsigned x = 0;
  for (int i = 0; i < 8; ++i) x ^= v[i];
  if (s == 1234.5f || x == 0x12345u) sink[threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int SHAPE, int F, bool LDSR>
void run(unsigned long long* d, float* sink) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<SHAPE, F, LDSR>), dim3(256), dim3(256), 0, 0, d, sink, iters);
  hipLaunchKernelGGL((k<SHAPE, F, LDSR>), dim3(256), dim3(256), 0, 0, d, sink, iters);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%dx%d  fillers %d%s : %6.1f cycles per MFMA\n", SHAPE, SHAPE, F, LDSR ? " (one is a ds_read_b128)" : "", double(h) / (iters * 8.0));
}
int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 64); hipMalloc(&sink, 4096);
  run<16, 0, false>(d, sink); run<16, 1, false>(d, sink); run<16, 2, false>(d, sink); run<16, 3, false>(d, sink); run<16, 2, true>(d, sink); run<16, 3, true>(d, sink);
  run<32, 0, false>(d, sink); run<32, 2, false>(d, sink); run<32, 3, false>(d, sink); run<32, 4, false>(d, sink); run<32, 5, false>(d, sink); run<32, 6, false>(d, sink);
  run<32, 4, true>(d, sink); run<32, 5, true>(d, sink);
  return 0;
}
