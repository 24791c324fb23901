// Residency census (experiments, not part of the library): how many workgroups of a given shape does one CU hold?
// Every workgroup spins ~20 us, stamps its start/end on the chip-wide 100 MHz clock together with HW_ID / XCC_ID;
// the host computes the maximum number of concurrently resident workgroups per CU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

template <int VGPRS>
__global__ void k_census(unsigned long long* out, int spin_ticks) {
  extern __shared__ float lds[];
  if (VGPRS >= 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
  if (VGPRS >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (VGPRS >= 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
  if (VGPRS >= 120) asm volatile("v_mov_b32 v119, 0" ::: "v119");
  if (VGPRS >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) lds[0] = 1.f;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t0;
    out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 4 + 2] = (uint64_t(__builtin_amdgcn_s_getreg(63508)) << 32) | uint32_t(__builtin_amdgcn_s_getreg(63492));
  }
}

template <int V>
void run(int threads, int ldsb) {
  const int nwg = 256 * 24;
  unsigned long long* d;
  hipMalloc(&d, nwg * 32);
  hipMemset(d, 0, nwg * 32);
  hipFuncSetAttribute((const void*)k_census<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  int api = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k_census<V>, threads, ldsb);
  hipLaunchKernelGGL(k_census<V>, dim3(nwg), dim3(threads), ldsb, 0, d, 2000);
  hipError_t e = hipDeviceSynchronize();
  std::vector<unsigned long long> h(nwg * 4);
  hipMemcpy(h.data(), d, nwg * 32, hipMemcpyDeviceToHost);
  hipFree(d);
  std::map<uint64_t, std::vector<std::pair<uint64_t, int>>> ev;
  for (int i = 0; i < nwg; ++i) {
    const uint64_t hw = h[i * 4 + 2];
    const uint64_t key = ((hw >> 32) & 0xF) << 16 | (hw & 0xFF00);  // xcc, se/sh/cu
    ev[key].push_back({h[i * 4 + 0], +1});
    ev[key].push_back({h[i * 4 + 1], -1});
  }
  int hist[40] = {0};
  for (auto& kv : ev) {
    std::sort(kv.second.begin(), kv.second.end());
    int c = 0, m = 0;
    for (auto& p : kv.second) { c += p.second; m = std::max(m, c); }
    hist[std::min(m, 39)]++;
  }
  printf("vgpr<=%3d threads %4d lds %6d : API %d blocks/CU; census max resident per CU:", V, threads, ldsb, api);
  for (int i = 0; i < 40; ++i) if (hist[i]) printf(" %d x%d", i, hist[i]);
  printf("  (%s)\n", hipGetErrorString(e));
}

int main() {
  const int shapes[][2] = {{256, 0}, {256, 33792}, {320, 0}, {320, 33792}, {320, 16896}, {512, 0}, {512, 33792}, {576, 0}, {576, 33792},
                           {384, 33792}, {448, 33792}, {192, 33792}, {128, 33792}};
  for (auto& s : shapes) {
    run<64>(s[0], s[1]);
    run<96>(s[0], s[1]);
    run<120>(s[0], s[1]);
    run<128>(s[0], s[1]);
  }
  return 0;
}
