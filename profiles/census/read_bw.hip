// What does a READ-ONLY stream of the aggregation's size reach on cold data?  Six 128 MB buffers in rotation (768 MB > the
// 256 MB memory-side cache), one event pair per launch, median.  Variants: 16-byte loads in flight per lane, plain / nontemporal.
//   hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw && ./read_bw
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ x, float* out, size_t n4) {
  const size_t base = (size_t(blockIdx.x) * 256 * U) + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t i = base + size_t(u) * 256;
    v[u] = i < n4 ? (NT ? __builtin_nontemporal_load(x + i) : x[i]) : f4{0.f, 0.f, 0.f, 0.f};
  }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
  if (s == 12345.678f) out[0] = s;   // never true: keeps the loads
}
template <int U, bool NT>
void run(const char* name, std::vector<f4*>& bufs, float* out, size_t n4) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> t;
  const unsigned grid = unsigned((n4 + 256 * U - 1) / (256 * U));
  for (int it = 0; it < 66; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(256), 0, 0, bufs[it % bufs.size()], out, n4);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (it >= 6) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  const double mb = n4 * 16 / 1e6;
  printf("%-28s grid %7u  median %6.1f us = %.2f TB/s   (p10 %.1f us)\n", name, grid, t[t.size() / 2], mb / t[t.size() / 2], t[t.size() / 10]);
}
int main() {
  const size_t bytes = size_t(8) * 31354 * 128 * 4, n4 = bytes / 16;   // the airfoil L0 message tensor
  std::vector<f4*> bufs(6);
  for (auto& p : bufs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
  float* out; hipMalloc(&out, 64);
  hipDeviceSynchronize();
  run<2, false>("2 loads in flight", bufs, out, n4);
  run<4, false>("4 loads in flight", bufs, out, n4);
  run<8, false>("8 loads in flight", bufs, out, n4);
  run<16, false>("16 loads in flight", bufs, out, n4);
  run<8, true>("8 loads, nontemporal", bufs, out, n4);
  run<16, true>("16 loads, nontemporal", bufs, out, n4);
  // the same bytes as a device copy (read + write), for reference
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> t;
  for (int it = 0; it < 36; ++it) {
    hipEventRecord(a); hipMemcpyAsync(bufs[(it + 3) % 6], bufs[it % 6], bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (it >= 6) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  printf("%-28s               median %6.1f us = %.2f TB/s (read + write)\n", "hipMemcpy D2D", t[t.size() / 2], 2 * bytes / 1e6 / t[t.size() / 2]);
  return 0;
}
