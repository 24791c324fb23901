// Experiment: LDS-DMA (global_load_lds_dwordx4) into LDS offsets beyond 64 KB; mfma_f32_16x16x32_bf16 K-slot check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ void k_glds(const float4* src, float4* out, unsigned off_bytes) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int lane = threadIdx.x;
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds) + off_bytes;
  glds16(src + lane, base);
  glds16(src + 64 + lane, base + 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[lane] = lds[off_bytes / 16 + lane];
  out[64 + lane] = lds[off_bytes / 16 + 64 + lane];
}

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
// D[m][n] = sum_k A[m][k] B[k][n]: lane holds A[m=lane&15][slot (lane>>4, i)], B[slot][n=lane&15].
__global__ void k_mfma(const float* A, const float* B, float* D) {  // A [16][32], B [32][16] row-major, exact in bf16
  const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)A[m * 32 + 8 * g + i];
    b[i] = (__bf16)B[(8 * g + i) * 16 + m];
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + m] = c[r];
}

int main() {
  std::vector<float> h(128 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = float(i) + 0.5f;
  float4 *src, *out;
  hipMalloc(&src, 2048); hipMalloc(&out, 2048);
  hipMemcpy(src, h.data(), 2048, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k_glds, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (unsigned off : {0u, 32768u, 65536u - 2048u, 65536u, 70000u / 16 * 16, 100000u / 16 * 16, 150000u / 16 * 16, 163840u - 2048u}) {
    hipMemset(out, 0, 2048);
    hipLaunchKernelGGL(k_glds, dim3(1), dim3(64), 163840, 0, src, out, off);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> r(128 * 4);
    hipMemcpy(r.data(), out, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < r.size(); ++i) bad += r[i] != h[i];
    printf("glds to LDS offset %6u: %s (%d mismatches) %s\n", off, bad ? "FAIL" : "ok", bad, hipGetErrorString(e));
  }
  // MFMA check with asymmetric integer matrices
  std::vector<float> A(16 * 32), B(32 * 16), Dh(256), Dref(256, 0.f);
  for (int m = 0; m < 16; ++m) for (int k = 0; k < 32; ++k) A[m * 32 + k] = float((m * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) B[k * 16 + n] = float((k * 5 + n * 13) % 9 - 4);
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int k = 0; k < 32; ++k) Dref[m * 16 + n] += A[m * 32 + k] * B[k * 16 + n];
  float *dA, *dB, *dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipDeviceSynchronize();
  hipMemcpy(Dh.data(), dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += Dh[i] != Dref[i];
  printf("mfma_f32_16x16x32_bf16 with A[m=lane&15][k=8(lane>>4)+i], B[k][n=lane&15], D[row=4(lane>>4)+r][col=lane&15]: %s (%d)\n", bad ? "FAIL" : "ok", bad);
  return 0;
}
