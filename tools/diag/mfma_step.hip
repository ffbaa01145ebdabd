// Diagnostic: what ONE STEP of the pipelined ICNN kernels costs by construction -- four back-to-back v_mfma_f32_32x32x2_f32
// (one dependent chain) followed by the step's side work -- as a function of where the operands live and what the side work is.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_step mfma_step.hip && ./mfma_step
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
// BSRC 0: B in VGPRs, 1: B in AGPRs.  ACC 0: accumulator in VGPRs, 1: in AGPRs.  NV: v_fma_f32 per step.  NL: ds_read_b128 per
// step (waited for one step later).  XLDS: the A operands come from an LDS read issued one step ahead (else constant registers)
template <int BSRC, int ACC, int NV, int NL, int XLDS> __global__ __launch_bounds__(256) void stream(float* out, int iters, unsigned long long* clocks) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.f + i * 1e-6f;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f;
  float w[4] = {a, a + 1, a + 2, a + 3};
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = a + i;
  const float m = 1.0000001f, c = 1e-9f;
  const f32x4* lp = (const f32x4*)lds + (threadIdx.x & 63);
  f32x4 x = lp[0], xn = lp[64], l0 = lp[128], l1 = lp[192], l2 = lp[0];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (XLDS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xn) : "v"((int)((threadIdx.x & 63) * 16)), "i"(1024 * (u & 3)) : "memory");
      if (NL >= 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(l0) : "v"((int)((threadIdx.x & 63) * 16)) : "memory");
      if (NL >= 2) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(l1) : "v"(0) : "memory");
      if (NL >= 3) asm volatile("ds_read_b128 %0, %1 offset:8208" : "=v"(l2) : "v"(0) : "memory");
#define MF4(CA, CB)                                                                                                                  \
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %5, %0\n\tv_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"                              \
                   "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\tv_mfma_f32_32x32x2_f32 %0, %4, %8, %0"                                  \
                   : "+" CA(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), CB(w[0]), CB(w[1]), CB(w[2]), CB(w[3]))
      if (BSRC == 0 && ACC == 0) MF4("v", "v");
      if (BSRC == 1 && ACC == 0) MF4("v", "a");
      if (BSRC == 0 && ACC == 1) MF4("a", "v");
      if (BSRC == 1 && ACC == 1) MF4("a", "a");
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(m), "v"(c));
      if (XLDS || NL) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (XLDS) x = xn;
        if (NL >= 1) f[0] += l0[1];
        if (NL >= 2) f[1] += l1[2];
        if (NL >= 3) f[2] += l2[3];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + x[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = t1 - t0;
}
template <int BSRC, int ACC, int NV, int NL, int XLDS> void run() {
  const int blocks = 256, threads = 256, iters = 2000;
  float* out; unsigned long long* clocks;
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&clocks, 16);
  hipLaunchKernelGGL((stream<BSRC, ACC, NV, NL, XLDS>), dim3(blocks), dim3(threads), 0, 0, out, iters / 10, clocks);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((stream<BSRC, ACC, NV, NL, XLDS>), dim3(blocks), dim3(threads), 0, 0, out, iters, clocks);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, clocks, 8, hipMemcpyDeviceToHost);
  printf("B in %s, acc in %s, %2d v_fma + %d ds_read_b128 per step, A from %s: %6.1f ticks per step of 4 MFMAs (256 = the MFMAs alone)\n",
         BSRC ? "AGPR" : "VGPR", ACC ? "AGPR" : "VGPR", NV, NL, XLDS ? "LDS " : "regs", (double)h / (iters * 8.0));
  hipFree(out); hipFree(clocks);
}
int main() {
  run<0, 0, 0, 0, 0>(); run<1, 0, 0, 0, 0>(); run<0, 1, 0, 0, 0>(); run<1, 1, 0, 0, 0>();
  run<0, 0, 8, 0, 0>(); run<1, 0, 8, 0, 0>(); run<1, 1, 8, 0, 0>();
  run<1, 0, 0, 0, 1>(); run<1, 0, 8, 0, 1>(); run<1, 0, 8, 1, 1>(); run<1, 0, 8, 2, 1>(); run<1, 0, 8, 3, 1>();
  run<1, 0, 16, 3, 1>();
  return 0;
}
