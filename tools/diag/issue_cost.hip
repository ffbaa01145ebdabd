// Diagnostic: issue cost (cycles per instruction) of the instruction kinds of the Newton loop for ONE wave per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// MODE 11 / 12: the same v_fma_f32 stream as MODE 0 but as 32 KB / 8 KB of straight-line code per loop trip (the Newton loop's shape:
// long unrolled bodies that do not fit the wave's instruction buffer), to see what the waves of a CU cost each other in
// instruction fetch
template <int MODE> __global__ void cost(float* out, unsigned long long* cycles, int iters) {
  float a[16]; double d[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3f + i + 1.f; d[i] = threadIdx.x * 1e-3 + i + 1.0; }
  unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE >= 11) {
    constexpr int kBody = MODE == 11 ? 256 : 64;  // x 16 instructions x 8 bytes
    for (int it = 0; it < iters / kBody; ++it) {
#pragma unroll
      for (int r = 0; r < kBody; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f + 1e-6f * (r + 1), 1e-3f);
    }
  } else
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 1e-3f);
      if (MODE == 1) d[i] = __builtin_fma(d[i], 1.0001, 1e-3);
      if (MODE == 2) d[i] = d[i] + 1e-3;
      if (MODE == 3) a[i] = __builtin_amdgcn_rsqf(a[i]) + 1.f;               // rsq + add
      if (MODE == 4) a[i] = (float)((double)a[i] + 1e-3);                    // cvt f32->f64, add f64, cvt back
      if (MODE == 5) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xF, 0xF, true));  // DPP add
      if (MODE == 6) a[i] = a[i] > 8.f ? a[i] * 0.5f : a[i] + 1.f;           // cmp + cndmask + 2 alu
      if (MODE == 7) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x141, 0xF, 0xF, true));  // row_half_mirror
      if (MODE == 8) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x111, 0xF, 0xF, true));  // row_shr:1
      if (MODE == 9) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a[i]), 0x80B1));                   // ds_swizzle quad xor 1 + add
      if (MODE == 10) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)((threadIdx.x ^ 1) << 2), __builtin_bit_cast(int, a[i])));  // ds_bpermute + add
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 256;  // one-wave workgroups: 256 = one per CU, 1024 = one per SIMD

  float* out; unsigned long long* cyc; static unsigned long long h[8192];
  (void)hipMalloc(&out, 8192 * 64 * 4); (void)hipMalloc(&cyc, 8192 * 8);
  const int iters = argc > 2 ? atoi(argv[2]) : 2048;  // (long runs: what the part's clock does under the load, see `wall`)
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[13] = {"v_fma_f32", "v_fma_f64", "v_add_f64", "v_rsq_f32 + v_add_f32", "cvt f32->f64 + v_add_f64 + cvt f64->f32", "v_add_f32_dpp (quad_perm)", "v_cmp + v_cndmask + mul + add", "v_add_f32_dpp (row_half_mirror)", "v_add_f32_dpp (row_shr:1)", "ds_swizzle_b32 + v_add_f32", "ds_bpermute_b32 + v_add_f32", "v_fma_f32, 32 KB straight-line body", "v_fma_f32, 8 KB straight-line body"};
  const int per[13] = {1, 1, 1, 2, 3, 1, 4, 1, 1, 2, 2, 1, 1};
  for (int mode = 0; mode < 13; ++mode) {
    float wall_ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, 0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(cost<0>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 1: hipLaunchKernelGGL(cost<1>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 2: hipLaunchKernelGGL(cost<2>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 3: hipLaunchKernelGGL(cost<3>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 4: hipLaunchKernelGGL(cost<4>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 5: hipLaunchKernelGGL(cost<5>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 6: hipLaunchKernelGGL(cost<6>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 7: hipLaunchKernelGGL(cost<7>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 8: hipLaunchKernelGGL(cost<8>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 9: hipLaunchKernelGGL(cost<9>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 11: hipLaunchKernelGGL(cost<11>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        case 12: hipLaunchKernelGGL(cost<12>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
        default: hipLaunchKernelGGL(cost<10>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); break;
      }
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&wall_ms, e0, e1);
    }
    (void)hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0; for (int i = 0; i < grid; ++i) { mean += (double)h[i]; mx = h[i] > mx ? (double)h[i] : mx; } mean /= grid;
    printf("grid %d  %-44s %.2f (slowest wave %.2f) cycles per group of %d instruction(s) (16 independent chains); wall %.3f ns per group (kernel %.3f ms)\n", grid, names[mode], mean / iters / 16, mx / iters / 16, per[mode], wall_ms * 1e6 / iters / 16, wall_ms);
  }
  return 0;
}
