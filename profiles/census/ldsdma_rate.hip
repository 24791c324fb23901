// What limits the weight stream of the chain kernels?  Every workgroup streams the same L2-resident buffer (a weight
// set: 204 KB) into a 3-slot LDS ring, 17 KB chunks, with 1 / 2 / 4 loader waves (pieces interleaved between them) and
// nothing else running; variants: LDS-DMA (global_load_lds_dwordx4) or loads into registers + ds_write_b128.
// Prints bytes / cycle / CU at 1 and 2 workgroups per CU.   hipcc --offload-arch=gfx950 -O3 ldsdma_rate.hip -o ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int PIECES = 17, CHUNK = PIECES * 1024, NR = 3, NCHUNK_SET = 12;   // 12 chunks = 204 KB set

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NW, int DEPTH>   // NW loader waves; DEPTH chunks in flight
__global__ __launch_bounds__(256) void k_dma(const float4* src, int nchunks, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  constexpr int MINE = (PIECES + NW - 1) / NW;     // pieces of a chunk this wave issues (upper bound)
  auto issue = [&](int j) {
    const float4* s = src + size_t(j % NCHUNK_SET) * (CHUNK / 16) + lane;
    const unsigned dst = lds0 + unsigned(j % NR) * CHUNK;
#pragma unroll
    for (int i = 0; i < MINE; ++i) {
      const int pc = i * NW + wave;
      if (pc < PIECES) glds16(s + pc * 64, dst + pc * 1024);
    }
  };
  const int mine = (PIECES - wave + NW - 1) / NW;   // exact count for this wave
  for (int j = 0; j < DEPTH && j < nchunks; ++j) issue(j);
  for (int j = 0; j < nchunks; ++j) {
    // chunk j landed when at most (DEPTH - 1) younger chunks of MINE pieces are outstanding
    if (mine == MINE) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * MINE) : "memory"); }
    else { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * (MINE - 1)) : "memory"); }
    if (NW > 1) __syncthreads();
    if (j + DEPTH < nchunks) issue(j + DEPTH);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int NW>   // register path: global_load_dwordx4 -> ds_write_b128, one chunk ahead
__global__ __launch_bounds__(256) void k_reg(const float4* src, int nchunks, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  constexpr int MINE = (PIECES + NW - 1) / NW;
  float4 r[2][MINE];
  auto fetch = [&](int j, int set) {
    const float4* s = src + size_t(j % NCHUNK_SET) * (CHUNK / 16) + lane;
#pragma unroll
    for (int i = 0; i < MINE; ++i) { const int pc = i * NW + wave; r[set][i] = s[(pc < PIECES ? pc : 0) * 64]; }
  };
  auto put = [&](int j, int set) {
    float4* d = lds + size_t(j % NR) * (CHUNK / 16) + lane;
#pragma unroll
    for (int i = 0; i < MINE; ++i) { const int pc = i * NW + wave; if (pc < PIECES) d[pc * 64] = r[set][i]; }
  };
  fetch(0, 0);
  for (int j = 0; j < nchunks; j += 2) {
    fetch(j + 1, 1); put(j, 0); if (NW > 1) __syncthreads();
    fetch(j + 2, 0); put(j + 1, 1); if (NW > 1) __syncthreads();
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0 + (unsigned long long)(r[0][0].x == 1e30f);
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float4* src; unsigned long long* cyc;
  hipMalloc(&src, size_t(NCHUNK_SET) * CHUNK + 65536); hipMemset(src, 0, size_t(NCHUNK_SET) * CHUNK + 65536);
  hipMalloc(&cyc, 4096 * 8);
  const int nch = 1200;
  const size_t ldsb = NR * CHUNK;
  std::vector<unsigned long long> h(4096);
  auto run = [&](const char* name, auto kern, int threads, int wg_per_cu) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const int grid = cus * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), ldsb, 0, src, nch, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), ldsb, 0, src, nch, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < grid; ++i) avg += double(h[i]); avg /= grid;
    const double bytes = double(nch) * CHUNK;
    printf("%-34s %d WG/CU: %.1f us, %.2f B/memtime-tick/WG (x100 MHz: %.1f GB/s per WG), %.1f GB/s per CU, chip %.2f TB/s\n", name, wg_per_cu, ms * 1e3,
           bytes / avg, bytes / avg * 0.1, bytes * wg_per_cu / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12);
  };
  for (int w = 1; w <= 2; ++w) {
    run("LDS-DMA 1 loader, 2 in flight", k_dma<1, 2>, 64, w);
    run("LDS-DMA 2 loaders, 2 in flight", k_dma<2, 2>, 128, w);
    run("LDS-DMA 4 loaders, 2 in flight", k_dma<4, 2>, 256, w);
    run("registers + ds_write 1 wave", k_reg<1>, 64, w);
    run("registers + ds_write 2 waves", k_reg<2>, 128, w);
    run("registers + ds_write 4 waves", k_reg<4>, 256, w);
  }
  return 0;
}
