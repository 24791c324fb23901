// Experiment: HBM write bandwidth of 16-byte-per-lane stores in two lane->address patterns:
//  A  lane-linear (a wave writes 1 KB contiguous per instruction)
//  B  the chain kernels' pattern: lane (row = lane & 15, g = lane >> 4) writes 16 B at row * 512 + 64 t + 16 g
//     (a wave instruction writes 16 separate 64-byte pieces; eight instructions complete 16 rows of 512 B)
// each plain and non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int PATTERN, bool NT>
__global__ __launch_bounds__(256) void k_store(float* out, long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 v = {1.f, 2.f, 3.f, float(lane)};
  for (long tile = blockIdx.x; tile * 64 < rows; tile += gridDim.x) {
    const long row0 = tile * 64 + wave * 16;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x4* p;
      if (PATTERN == 0) p = reinterpret_cast<f32x4*>(out + row0 * 128 + t * 256 + lane * 4);          // 1 KB contiguous
      else if (PATTERN == 1) p = reinterpret_cast<f32x4*>(out + (row0 + (lane & 15)) * 128 + 16 * t + 4 * (lane >> 4));   // chain layout
      else if (PATTERN == 2)   // 8 rows x 128 B per instruction: lane (r = lane & 7, seg = lane >> 3), t -> (row half, 128-B column)
        p = reinterpret_cast<f32x4*>(out + (row0 + 8 * (t & 1) + (lane & 7)) * 128 + 32 * (t >> 1) + 4 * (lane >> 3));
      else if (PATTERN == 3)   // 4 rows x 256 B per instruction
        p = reinterpret_cast<f32x4*>(out + (row0 + 4 * (t & 3) + (lane & 3)) * 128 + 64 * (t >> 2) + 4 * (lane >> 2));
      else {                   // 32x32 MFMA accumulator layout: lane (row = lane & 31, h = lane >> 5), 32-byte pieces; two passes of 8
        const long r32 = tile * 64 + (wave >> 1) * 32 + (lane & 31);
        p = reinterpret_cast<f32x4*>(out + r32 * 128 + 8 * (t + 8 * (wave & 1)) + 4 * (lane >> 5));
      }
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
  }
}
template <int PATTERN>
__global__ __launch_bounds__(256) void k_load(const float* in, float* sink, long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long tile = blockIdx.x; tile * 64 < rows; tile += gridDim.x) {
    const long row0 = tile * 64 + wave * 16;
    f32x4 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4* p;
      if (PATTERN == 0) p = reinterpret_cast<const f32x4*>(in + row0 * 128 + t * 256 + lane * 4);
      else if (PATTERN == 1) p = reinterpret_cast<const f32x4*>(in + (row0 + (lane & 15)) * 128 + 16 * t + 4 * (lane >> 4));
      else p = reinterpret_cast<const f32x4*>(in + (row0 + 8 * (t & 1) + (lane & 7)) * 128 + 32 * (t >> 1) + 4 * (lane >> 3));
      v[t] = *p;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) acc += v[t];
  }
  if (acc[0] == 12345.678f) sink[threadIdx.x] = acc[1] + acc[2] + acc[3];
}
template <int P> void runl(float* d, long rows, const char* name) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_load<P>), dim3(2048), dim3(256), 0, 0, d, d, rows);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_load<P>), dim3(2048), dim3(256), 0, 0, d, d, rows);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-40s %7.1f us per 512 MB  = %.2f TB/s\n", name, ms * 100, rows * 512.0 * 10 / (ms * 1e-3) / 1e12);
}
template <int P, bool NT> void run(float* d, long rows, const char* name) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_store<P, NT>), dim3(2048), dim3(256), 0, 0, d, rows);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_store<P, NT>), dim3(2048), dim3(256), 0, 0, d, rows);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-40s %7.1f us per 512 MB  = %.2f TB/s\n", name, ms * 100, rows * 512.0 * 10 / (ms * 1e-3) / 1e12);
}
int main() {
  const long rows = 1 << 20;  // 512 MB
  float* d; hipMalloc(&d, rows * 512);
  run<0, false>(d, rows, "lane-linear, plain");
  run<0, true>(d, rows, "lane-linear, non-temporal");
  run<1, false>(d, rows, "chain layout (64 B pieces), plain");
  run<1, true>(d, rows, "chain layout (64 B pieces), nt");
  run<2, false>(d, rows, "8 rows x 128 B pieces, plain");
  run<2, true>(d, rows, "8 rows x 128 B pieces, nt");
  run<3, false>(d, rows, "4 rows x 256 B pieces, plain");
  run<3, true>(d, rows, "4 rows x 256 B pieces, nt");
  run<4, false>(d, rows, "32x32 layout (32 B pieces), plain");
  run<4, true>(d, rows, "32x32 layout (32 B pieces), nt");
  runl<0>(d, rows, "LOAD lane-linear");
  runl<1>(d, rows, "LOAD chain layout (64 B pieces)");
  runl<2>(d, rows, "LOAD 8 rows x 128 B pieces");
  return 0;
}
