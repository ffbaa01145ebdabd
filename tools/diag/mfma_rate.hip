// Diagnostic: what a stream of DEPENDENT v_mfma_f32_32x32x2_f32 (the accumulator chain of the mesh GEMM kernels) sustains on the
// whole chip, against the 157 TFLOP/s the exact-f32 kernels are priced against (MI355X_MICROARCH.md), and the same for
// v_mfma_f32_32x32x16_bf16.  Waves per SIMD 1 and 2 (the mesh kernels run 2), 256 and 1024 workgroups.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int KIND, int CHAINS> __global__ __launch_bounds__(512) void stream(float* out, int iters, unsigned long long* clocks) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long m0, m1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[c], 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m1)::"memory");
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = m1 - m0; }
}
template <int KIND, int CHAINS> void run(const char* name, int blocks, int threads, int iters) {
  float* out; unsigned long long* clocks;
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&clocks, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((stream<KIND, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, iters / 10, clocks);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((stream<KIND, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, iters, clocks);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost);
  const double mfmas = (double)blocks * (threads / 64) * iters * 16.0 * CHAINS;
  const double flops = mfmas * (KIND == 0 ? 4096.0 : 32768.0);
  printf("%-34s %4d x %4d threads, %d chain(s): %8.3f ms  %7.1f TFLOP/s  | wave 0: %.1f cycles per MFMA (readcyclecounter), %.1f s_memtime ticks per MFMA, %.2f ticks/ns\n",
         name, blocks, threads, CHAINS, ms, flops / (ms * 1e-3) / 1e12, (double)h[0] / (iters * 16.0 * CHAINS), (double)h[1] / (iters * 16.0 * CHAINS), (double)h[1] / (ms * 1e6));
  hipFree(out); hipFree(clocks);
}
int main() {
  const int iters = 4000;
  run<0, 1>("f32 32x32x2, 1 wave / SIMD", 256, 256, iters);
  run<0, 1>("f32 32x32x2, 2 waves / SIMD", 256, 512, iters);
  run<0, 2>("f32 32x32x2, 2 waves / SIMD", 256, 512, iters);
  run<0, 1>("f32 32x32x2, 4 x 256 workgroups", 1024, 512, iters);
  run<1, 1>("bf16 32x32x16, 1 wave / SIMD", 256, 256, iters);
  run<1, 1>("bf16 32x32x16, 2 waves / SIMD", 256, 512, iters);
  run<1, 2>("bf16 32x32x16, 2 waves / SIMD", 256, 512, iters);
  return 0;
}
