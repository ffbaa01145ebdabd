// dpll_icnn_pipe_api.hpp -- launchers of the one-wave-per-SIMD, software-pipelined ICNN GEMM kernels (dpll_icnn_pipe.hip)
// for the mesh pipeline in dpll_kernels.hip.  Same operands and results as the 8-wave kernels of dpll_mesh_kernels.hpp
// (DeepSupportConvex / HomogeneousICNN, geometry.py:309-325, deep_support_function.py:213-266) except that U0 travels
// between fwd2 and bwd1 in the accumulator layout of the MFMA (pipe_u0_floats() floats) instead of row-major.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "dpll_icnn.hpp"

namespace dpll_pipe {

constexpr int kTileRows = 32;   // rows (support queries) per tile
constexpr int kMaxBlocks = 256; // one 4-wave workgroup per CU: its 256 x 256 weight block lives in the CU's registers

inline long long tiles(long long N) { return (N + kTileRows - 1) / kTileRows; }
inline int blocks(long long N) { const long long t = tiles(N); return (int)(t < kMaxBlocks ? t : kMaxBlocks); }
inline size_t u0_floats(long long N) { return (size_t)tiles(N) * kTileRows * 256; }  // whole tiles

// PRE1 = act(Q Wd0) |Wh| + Q Wd1 -> mask bits M1 (N, 8)
int fwd1(hipStream_t stream, const float* x, long long ld, long long N, const dpll::IcnnWeights<float>& w, const float* Af,
         uint32_t* M1);
// U1 = |wout| . m1;  V = U1 |Wh|^T;  U0 = V . m0 (-> U0t, accumulator layout);  P = U1 Wd1^T + U0 Wd0^T
int fwd2(hipStream_t stream, const float* x, long long ld, long long N, const dpll::IcnnWeights<float>& w, const float* ATf,
         const float* a, const uint32_t* M1, float* U0t, float* P);
// Vb = (RB Wd0) . m0 (-> operand tiles VbT);  U1b = Vb |Wh| + RB Wd1;  partial rows [d|wout| | dWd1 | dWd0] per workgroup
int bwd1(hipStream_t stream, const float* x, long long ld, long long N, const dpll::IcnnWeights<float>& w, const float* Af,
         const float* a, const uint32_t* M1, const float* U0t, const float* RB, double* partial, float* VbT);
// the same three on the 16-bit matrix cores, operands split into 2 planes: bf16 (dpll_solver_opts_t.mesh_gemm = 2) or, f16 = true,
// fp16 with the low plane scaled by 2^11 (mesh_gemm = 4: f32-grade products, dpll_mesh_bf16.hpp); Ab / ATb = the weight planes of
// icnn_prep_bf16_kernel<2, f16>; Vb leaves row-major for icnn_bwd2_bf16
int fwd1_bf16(hipStream_t stream, const float* x, long long ld, long long N, const dpll::IcnnWeights<float>& w, const void* Ab, uint32_t* M1, bool f16 = false);
int fwd2_bf16(hipStream_t stream, const float* x, long long ld, long long N, const dpll::IcnnWeights<float>& w, const void* ATb,
              const float* a, const uint32_t* M1, float* U0t, float* P, bool f16 = false);
int bwd1_bf16(hipStream_t stream, const float* x, long long ld, long long N, const dpll::IcnnWeights<float>& w, const void* Ab,
              const float* a, const uint32_t* M1, const float* U0t, const float* RB, double* partial, float* Vb, bool f16 = false,
              unsigned* rbmax = nullptr);  // (f16: the launch's largest |r_bar| entry, float bits, atomicMax into *rbmax)

}  // namespace dpll_pipe
