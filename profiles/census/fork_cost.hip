// What a fork point costs the caller's stream: hipEventRecord + hipStreamWaitEvent (a barrier packet in the caller's queue)
// against the event bound to the kernel's own dispatch packet (hipExtLaunchKernelGGL stopEvent) and against no fork.
//   hipcc --offload-arch=gfx950 -O3 fork_cost.hip -o fork_cost && ./fork_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

int main() {
  float *a, *b;
  CK(hipMalloc(&a, 256 * 256 * 4)); CK(hipMalloc(&b, 256 * 256 * 4));
  CK(hipMemset(a, 0, 256 * 256 * 4)); CK(hipMemset(b, 0, 256 * 256 * 4));
  hipStream_t m, s;
  CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev, jn;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
  CK(hipEventCreateWithFlags(&jn, hipEventDisableTiming | hipEventDisableSystemFence));
  const int N = 400;
  for (int iters : {2000, 20000}) {
    for (int mode = 0; mode < 4; ++mode) {
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
          if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, m, nullptr, ev, 0, a, iters);
          else hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, m, a, iters);
          if (mode == 1 || mode == 3) CK(hipEventRecord(ev, m));
          if (mode >= 1) {
            CK(hipStreamWaitEvent(s, ev, 0));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, b, iters / 2);
          }
          hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, m, a, iters);
          if (mode == 3) {   // + a join: the caller's stream waits for the side stream
            CK(hipEventRecord(jn, s));
            CK(hipStreamWaitEvent(m, jn, 0));
          }
        }
        CK(hipStreamSynchronize(m)); CK(hipStreamSynchronize(s));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        if (us < best) best = us;
      }
      const char* names[4] = {"no fork", "hipEventRecord + wait", "stopEvent on the dispatch + wait", "record + wait + join"};
      printf("iters %6d  %-34s %8.2f us per pair of kernels\n", iters, names[mode], best);
    }
  }
  return 0;
}
