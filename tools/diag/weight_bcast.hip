// Diagnostic: how long 256 workgroups of 512 threads take to pull the SAME 256 KB (the weight fragments of a mesh GEMM kernel: 32 x
// dwordx4 per lane) out of the L2s, and whether the order in which they ask matters.  MODE 0: every workgroup reads fragments
// 0..31 in that order (what the kernels do); MODE 1: workgroup b starts at fragment 4 (b & 7) (registers assigned statically per
// instantiation: the whole body is instantiated per offset); MODE 2: every workgroup reads its OWN 256 KB (no sharing: 64 MB of
// distinct lines); MODE 3: as 0 but the 8 waves of a workgroup start 4 fragments apart.
//   hipcc -O3 --offload-arch=gfx950 -o weight_bcast weight_bcast.hip && ./weight_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int R> __device__ __forceinline__ float body(const f32x4* __restrict__ f) {
  f32x4 v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) { const int q = (k + R) & 31; v[q] = f[q * 64]; }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) s += v[q][0] + v[q][1] + v[q][2] + v[q][3];
  return s;
}
template <int MODE> __global__ __launch_bounds__(512) void pull(const float* __restrict__ W, float* __restrict__ out, unsigned long long* ticks) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* base = MODE == 2 ? W + (size_t)blockIdx.x * 65536 : W;
  const f32x4* f = (const f32x4*)base + (wv * 32) * 64 + lane;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float s;
  if (MODE == 1) {
    switch (blockIdx.x & 7) {
      case 0: s = body<0>(f); break; case 1: s = body<4>(f); break; case 2: s = body<8>(f); break; case 3: s = body<12>(f); break;
      case 4: s = body<16>(f); break; case 5: s = body<20>(f); break; case 6: s = body<24>(f); break; default: s = body<28>(f); break;
    }
  } else if (MODE == 3) {
    switch (wv) {
      case 0: s = body<0>(f); break; case 1: s = body<4>(f); break; case 2: s = body<8>(f); break; case 3: s = body<12>(f); break;
      case 4: s = body<16>(f); break; case 5: s = body<20>(f); break; case 6: s = body<24>(f); break; default: s = body<28>(f); break;
    }
  } else {
    s = body<0>(f);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, const float* W, bool warm) {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 4 * 256 * 512); hipMalloc(&ticks, 8 * 256);
  float best = 1e9f; unsigned long long med = 0;
  for (int rep = 0; rep < 5; ++rep) {
    if (!warm) { static float* trash = nullptr; if (!trash) hipMalloc(&trash, 512u << 20); hipMemsetAsync(trash, rep, 512u << 20, 0); }  // evict the caches
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((pull<MODE>), dim3(256), dim3(512), 0, 0, W, out, ticks);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    unsigned long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int i = 0; i < 256; ++i) mx = h[i] > mx ? h[i] : mx;
    med = mx;
  }
  printf("%-52s %s: kernel %.2f us, slowest workgroup %llu ticks from first request to last arrival\n", name, warm ? "L2 warm" : "cold  ", best * 1e3, med);
  hipFree(out); hipFree(ticks);
}
int main() {
  float* W; hipMalloc(&W, (size_t)256 * 65536 * 4); hipMemset(W, 0, (size_t)256 * 65536 * 4);
  for (int warm = 1; warm >= 0; --warm) {
    run<0>("same 256 KB, same order", W, warm);
    run<1>("same 256 KB, workgroups start 4 fragments apart", W, warm);
    run<3>("same 256 KB, waves start 4 fragments apart", W, warm);
    run<2>("each workgroup its own 256 KB (64 MB)", W, warm);
  }
  return 0;
}
