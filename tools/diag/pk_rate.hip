// Diagnostic: issue cost of v_pk_fma_f32 against v_fma_f32 for ONE wave per SIMD (the regime of the loss kernel).
//   hipcc -O3 --offload-arch=gfx950 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void rate(float* out, unsigned long long* cycles, int iters, int active_lanes) {
  f2 a[8]; float b[16];
  for (int i = 0; i < 8; ++i) a[i] = (f2){threadIdx.x * 1e-3f + i, 1.f + i};
  for (int i = 0; i < 16; ++i) b[i] = threadIdx.x * 1e-3f + i;
  const f2 m = (f2){1.0001f, 0.9999f}, c = (f2){1e-3f, -1e-3f};
  unsigned long long t0 = __builtin_readcyclecounter();
  if ((int)threadIdx.x < active_lanes)
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 8 independent packed FMAs = 16 flops-pairs
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    } else if (MODE == 1) {  // 16 independent scalar FMAs (same arithmetic)
#pragma unroll
      for (int i = 0; i < 16; ++i) b[i] = __builtin_fmaf(b[i], 1.0001f, 1e-3f);
    } else {  // one dependent chain of packed FMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) a[0] = __builtin_elementwise_fma(a[0], m, c);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  for (int i = 0; i < 16; ++i) s += b[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; unsigned long long h[256];
  hipMalloc(&out, 256 * 64 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 4096;
  const char* names[3] = {"8 independent v_pk_fma_f32", "16 independent v_fma_f32", "8 dependent v_pk_fma_f32"};
  for (int lanes = 64; lanes >= 16; lanes /= 2)
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(64), 0, 0, out, cyc, iters, lanes);
      if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(64), 0, 0, out, cyc, iters, lanes);
      if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(64), 0, 0, out, cyc, iters, lanes);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
    printf("active lanes %2d  %-32s %.2f cycles per loop body (%.2f per instruction)\n", lanes, names[mode], mean / iters, mean / iters / (mode == 1 ? 16 : 8));
  }
  return 0;
}
