// What does ONE chunk of a single-round chain stage cost a lone wave per SIMD?  4 compute waves (one per SIMD) of one
// workgroup per CU run the inner loop of mfma_stage (chain.hip): per chunk a barrier, 16 ds_read_b128 of A fragments
// (17 KB chunk in LDS), 24 v_mfma_f32_16x16x32_f16 on 8 accumulators (three products each) and the two-way split of the
// next K block (16 v_fma_mix).  Components are switched off one at a time; optional loader waves stream the chunks with
// LDS-DMA as the real kernel does.     hipcc --offload-arch=gfx950 -O3 stage_lone.hip -o stage_lone && ./stage_lone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
constexpr int CHF = 256 + 8 * 256 * 2;   // floats per chunk (1 KB header + 16 fragments of 1 KB)
constexpr int CH4 = CHF / 4;
constexpr int NR = 6;

__device__ __forceinline__ f32x4 mma(const float4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void split_h2(float x0, float x1, float s, unsigned& h, unsigned& l) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}
// MODE bits: 1 barrier, 2 fragment reads from LDS, 4 MFMAs, 8 split VALU, 16 loader waves stream chunks (LDS-DMA)
template <int MODE>
__global__ __launch_bounds__(512) void k_stage(const float4* weights, float* out, unsigned long long* cyc, int nchunk, int nload, float scale) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= 4) {   // loader wave(s): the ring protocol of chain.hip (nr - 1 chunks in flight, one barrier per chunk)
    if (!(MODE & 16)) { if (MODE & 1) for (int j = 0; j < nchunk; ++j) asm volatile("s_barrier" ::: "memory"); return; }
    const int li = __builtin_amdgcn_readfirstlane(wave - 4), mine = (17 - li + nload - 1) / nload;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    int islot = 0, ic = 0;
    auto issue = [&]() {
      const float4* src = weights + size_t(ic % 20) * CH4 + lane;
      const unsigned dst = lds0 + unsigned(islot) * unsigned(CHF * 4);
      for (int i = 0; i < mine; ++i) glds16(src + (li + i * nload) * 64, __builtin_amdgcn_readfirstlane(dst + (li + i * nload) * 1024));
      ++ic; if (++islot == NR) islot = 0;
    };
    for (int j = 0; j < NR - 1 && j < nchunk; ++j) issue();
    for (int j = 0; j < nchunk; ++j) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the real loader counts; this is the conservative form)
      if (MODE & 1) asm volatile("s_barrier" ::: "memory");
      if (j + NR - 1 < nchunk) issue();
    }
    return;
  }
  f32x4 acc[8], act[8];
  for (int t = 0; t < 8; ++t) { acc[t] = f32x4{0, 0, 0, 0}; const float4 w0 = weights[lane + 64 * t]; act[t] = f32x4{w0.x + lane, w0.y + 1.f, w0.z + 2.f, w0.w + t}; }
  u32x4 bh = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, bl = {0, 0, 0, 0};
  float4 g[16];
  for (int k = 0; k < 16; ++k) g[k] = make_float4(1.f, 2.f, 3.f, float(k));
  int slot = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < nchunk; ++c) {
    if (MODE & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float4* body = lds + slot * CH4 + 64 + lane;
    if (++slot == NR) slot = 0;
    if (MODE & 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) g[k] = body[k * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 4) {
#pragma unroll
      for (int t = 0; t < 8; t += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t + k] = mma(g[2 * (t + k)], bl, acc[t + k]);
        if (t == 0 && (MODE & 8)) {
          u32x4 nh, nl;
#pragma unroll
          for (int v = 0; v < 4; ++v) { unsigned h, l; split_h2(act[(c & 3) * 2 + (v >> 1)][2 * (v & 1)], act[(c & 3) * 2 + (v >> 1)][2 * (v & 1) + 1], scale, h, l); nh[v] = h; nl[v] = l; }
          bh = nh; bl = nl;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t + k] = mma(g[2 * (t + k)], bh, acc[t + k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[t + k] = mma(g[2 * (t + k) + 1], bh, acc[t + k]);
      }
    } else if (MODE & 8) {
      u32x4 nh, nl;
#pragma unroll
      for (int v = 0; v < 4; ++v) { unsigned h, l; split_h2(act[(c & 3) * 2 + (v >> 1)][2 * (v & 1)], act[(c & 3) * 2 + (v >> 1)][2 * (v & 1) + 1], scale, h, l); nh[v] = h; nl[v] = l; }
      bh = nh; bl = nl;
    }
    if (!(MODE & 4) && (MODE & 2)) { for (int k = 0; k < 16; ++k) acc[k & 7][0] += g[k].x; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  s += __uint_as_float(bh[0]) + __uint_as_float(bl[1]);
  if (s == 1234.5f) out[0] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int MODE>
void run(const char* name, const float4* w, float* out, unsigned long long* cyc, int nload) {
  const int nchunk = 400, grid = 256;
  const size_t ldsb = size_t(NR) * CHF * 4 + 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stage<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_stage<MODE>), dim3(grid), dim3((4 + nload) * 64), ldsb, 0, w, out, cyc, nchunk, nload, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-64s %7.0f cycles per chunk (median wave; p90 %.0f)\n", name, double(h[h.size() / 2]) / nchunk, double(h[h.size() * 9 / 10]) / nchunk);
}
int main() {
  float4* w; float* out; unsigned long long* cyc;
  hipMalloc(&w, size_t(20) * CHF * 4); hipMemset(w, 0, size_t(20) * CHF * 4);
  hipMalloc(&out, 64); hipMalloc(&cyc, 256 * 4 * 8);
  run<4>("24 MFMA only", w, out, cyc, 1);
  run<4 | 8>("24 MFMA + split (16 v_fma_mix)", w, out, cyc, 1);
  run<2 | 4>("16 ds_read_b128 + 24 MFMA", w, out, cyc, 1);
  run<2 | 4 | 8>("16 ds_read_b128 + 24 MFMA + split", w, out, cyc, 1);
  run<1 | 2 | 4 | 8>("barrier + reads + MFMA + split (no loader traffic)", w, out, cyc, 1);
  run<1 | 2 | 4 | 8 | 16>("barrier + reads + MFMA + split + 1 loader wave (LDS-DMA)", w, out, cyc, 1);
  run<1 | 2 | 4 | 8 | 16>("barrier + reads + MFMA + split + 2 loader waves", w, out, cyc, 2);
  run<1 | 2 | 4 | 8 | 16>("barrier + reads + MFMA + split + 4 loader waves", w, out, cyc, 4);
  run<1 | 2>("barrier + reads only", w, out, cyc, 1);
  run<2>("reads only", w, out, cyc, 1);
  run<1>("barrier only", w, out, cyc, 1);
  run<1 | 16>("barrier + 2 loader waves only", w, out, cyc, 2);
  return 0;
}
