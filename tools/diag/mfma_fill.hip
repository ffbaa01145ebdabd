// Diagnostic: does a lone wave's VALU / LDS work hide behind its own v_mfma_f32_32x32x2_f32 chain (as it does behind the bf16
// MFMAs, MI355X_MICROARCH.md "vector-instruction ISSUE cost"), or does the f32 MFMA -- whose rate IS the f32 vector rate --
// occupy the vector ALU?  One wave per SIMD, a dependent MFMA chain, F independent fillers between consecutive MFMAs.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_fill mfma_fill.hip && ./mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
// KIND 0: f32 32x32x2, 1: bf16 32x32x16.  FILL 0: v_fma_f32, 1: ds_read_b32 (LDS), 2: v_accvgpr_read (AGPR -> VGPR)
template <int KIND, int F, int FILL> __global__ __launch_bounds__(256) void stream(float* out, int iters, unsigned long long* clocks) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1.f;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = a + i;
  const float m = 1.0000001f, c = 1e-9f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < F; ++j) {
        if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(m), "v"(c));
        if (FILL == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(f[j]) : "v"((int)(threadIdx.x * 4 + j * 4)) : "memory");
      }
      if (FILL == 1 && F > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = t1 - t0;
}
template <int KIND, int F, int FILL> void run() {
  const int blocks = 256, threads = 256, iters = 2000;
  float* out; unsigned long long* clocks;
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&clocks, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((stream<KIND, F, FILL>), dim3(blocks), dim3(threads), 0, 0, out, iters / 10, clocks);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((stream<KIND, F, FILL>), dim3(blocks), dim3(threads), 0, 0, out, iters, clocks);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, clocks, 8, hipMemcpyDeviceToHost);
  const double mfmas = (double)iters * 16.0;
  printf("%s MFMA chain + %2d %s between MFMAs: %7.1f ns per MFMA, %6.1f readcyclecounter ticks per MFMA\n", KIND == 0 ? "f32 32x32x2 " : "bf16 32x32x16",
         F, FILL == 0 ? "v_fma_f32  " : "ds_read_b32", ms * 1e6 / mfmas, (double)h / mfmas);
  hipFree(out); hipFree(clocks);
}
int main() {
  run<0, 0, 0>(); run<0, 2, 0>(); run<0, 4, 0>(); run<0, 8, 0>(); run<0, 12, 0>(); run<0, 16, 0>();
  run<0, 1, 1>(); run<0, 2, 1>(); run<0, 4, 1>(); run<0, 8, 1>();
  run<1, 0, 0>(); run<1, 2, 0>(); run<1, 4, 0>(); run<1, 6, 0>(); run<1, 8, 0>();
  return 0;
}
