// Experiment: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 values equal to their own element index;
// lane L supplies byte address 8 L (pattern A) or a row-major 16x16 tile address (pattern B); print what each lane gets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

__global__ void k_tr(unsigned* out, int pattern) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int L = threadIdx.x;
  unsigned addr;
  if (pattern == 0) addr = 8 * L;                        // lane L -> elements 4L..4L+3
  else addr = ((L & 15) * 64 + (L >> 4) * 4) * 2;        // row (L&15) of a [16 rows][64 cols] u16 matrix, cols 4(L>>4)..+3
  addr += (unsigned)(uintptr_t)lds;
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[2 * L] = v[0];
  out[2 * L + 1] = v[1];
}

int main() {
  unsigned* d;
  hipMalloc(&d, 512);
  for (int pattern = 0; pattern < 2; ++pattern) {
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d, pattern);
    hipDeviceSynchronize();
    std::vector<unsigned> h(128);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("pattern %d (lane: 4 element indices received)\n", pattern);
    for (int L = 0; L < 64; ++L) {
      printf("  L%02d: %4u %4u %4u %4u", L, h[2 * L] & 0xffff, h[2 * L] >> 16, h[2 * L + 1] & 0xffff, h[2 * L + 1] >> 16);
      if (L % 4 == 3) printf("\n");
    }
  }
  return 0;
}
