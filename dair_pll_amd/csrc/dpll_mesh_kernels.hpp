// dpll_mesh_kernels.hpp -- ICNN (DeepSupportConvex) kernels of the mesh-geometry path; included by
// dpll_kernels.hip.  One floating body with a learned convex shape against the ground: N = 4 * batch support
// queries per launch, query n = 4 * item + s.
//
// Pipeline of one loss call (dpll_contactnets_loss_mesh):
//   icnn_prep      A = |Wh|, A^T, a = |wout|                                   (65,792 elements)
//   icnn_fwd1      H0 = act(Q Wd0);  PRE1 = H0 A + Q Wd1  -> mask bits M1     GEMM  N x 256 x 256
//   icnn_fwd2      U1 = a . m1;  V = U1 A^T;  U0 = V . m0;  P = U1 Wd1^T + U0 Wd0^T   GEMM  N x 256 x 256
//   loss_kernel_mesh   the ContactNets loss with witnesses P; emits r_bar (N x 3)
//   icnn_bwd1      Vb = (RB Wd0) . m0;  U1b = Vb A + RB Wd1;  partial d|wout|, dWd1, dWd0   GEMM  N x 256 x 256
//   icnn_bwd2      d|Wh| = Vb^T U1                                                          GEMM  256 x 256 x N
//   icnn_reduce    fixed-order sums of all partials, sign chain, cast to the parameter dtype
//
// The GEMM kernels exist in two forms: a generic register-tiled VALU form (any T; the float64 path and the
// checker of the MFMA form) and, for float, an MFMA form on v_mfma_f32_32x32x2_f32 (exact f32, so both forms
// agree to rounding order).
#pragma once

#include "dpll_icnn.hpp"

namespace {

using namespace dpll;

constexpr int kW = kIcnnWidth;      // 256
constexpr int kTileRows = 16;       // queries per tile in the generic kernels
constexpr int kMaskWords = kW / 32; // 8 mask words per query
constexpr int kB1Cols = kW + 3 * kW + 3 * kW;  // partial row of icnn_bwd1: [d|wout| | dWd1 | dWd0]

template <typename T> struct MeshBuffers {
  // workspace carve-up (device pointers)
  T* A;         // (256, 256) |Wh|
  T* AT;        // (256, 256) |Wh|^T
  T* a;         // (256,) |wout|
  T* P;         // (N, 3) support points
  T* RB;        // (N, 3) weighted r_bar
  uint32_t* M1; // (N, 8) mask bits of layer 1
  T* U0;        // (N, 256)
  double* rows; // loss kernel rows (blocks, 16)
  double* b1;   // icnn_bwd1 partial rows (b1_blocks, kB1Cols)
  T* slabs;     // icnn_bwd2 slabs (n_slabs, 256 * 256)
  int b1_blocks, n_slabs, loss_blocks;
};

__device__ __forceinline__ float mask_factor(uint32_t word, int bit) { return ((word >> bit) & 1u) ? 1.0f : float(kIcnnSlope); }

template <typename T>
__global__ __launch_bounds__(256) void icnn_prep_kernel(IcnnWeights<T> w, T* __restrict__ A, T* __restrict__ AT,
                                                        T* __restrict__ a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < kW * kW) {
    const int k = idx / kW, j = idx % kW;
    const T v = tabs(w.Wh[idx]);
    A[idx] = v;
    AT[j * kW + k] = v;
  }
  if (idx < kW) a[idx] = tabs(w.wout[idx]);
}

// query directions of a tile of rows into LDS; state row of item i starts at x + i * ld, quaternion first
template <typename T>
__device__ __forceinline__ void load_queries(const T* __restrict__ x, long long ld, const T* __restrict__ pert,
                                             long long n0, long long N, T (*Qs)[3]) {
  if (threadIdx.x < kTileRows) {
    const long long n = n0 + threadIdx.x;
    T q[3] = {T(0), T(0), T(1)};
    if (n < N) {
      T quat[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) quat[i] = x[(n >> 2) * ld + i];
      icnn_query<T>(quat, pert + 3 * (n & 3), q);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) Qs[threadIdx.x][i] = q[i];
  }
}

// ---- generic (VALU) GEMM kernels: 256 threads = 256 output columns, 16 rows per tile --------------
template <typename T>
__global__ __launch_bounds__(256) void icnn_fwd1_kernel(const T* __restrict__ x, long long ld, long long N,
                                                        IcnnWeights<T> w, const T* __restrict__ A,
                                                        uint32_t* __restrict__ M1) {
  __shared__ T Qs[kTileRows][3];
  __shared__ __attribute__((aligned(16))) T Hs[kW][kTileRows];
  const int j = threadIdx.x;
  const T wd0[3] = {w.Wd0[j], w.Wd0[kW + j], w.Wd0[2 * kW + j]};
  const T wd1[3] = {w.Wd1[j], w.Wd1[kW + j], w.Wd1[2 * kW + j]};
  const long long tiles = (N + kTileRows - 1) / kTileRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kTileRows;
    __syncthreads();
    load_queries<T>(x, ld, w.pert, n0, N, Qs);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) Hs[j][r] = icnn_act(Qs[r][0] * wd0[0] + Qs[r][1] * wd0[1] + Qs[r][2] * wd0[2]);
    __syncthreads();
    T acc[kTileRows];
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) acc[r] = Qs[r][0] * wd1[0] + Qs[r][1] * wd1[1] + Qs[r][2] * wd1[2];
    for (int k = 0; k < kW; ++k) {
      const T wk = A[k * kW + j];
#pragma unroll
      for (int r = 0; r < kTileRows; ++r) acc[r] += Hs[k][r] * wk;
    }
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) {
      const unsigned long long b = __ballot(acc[r] > T(0));
      if ((threadIdx.x & 63) == 0 && n0 + r < N) {
        M1[(n0 + r) * kMaskWords + 2 * (threadIdx.x >> 6)] = (uint32_t)(b & 0xffffffffull);
        M1[(n0 + r) * kMaskWords + 2 * (threadIdx.x >> 6) + 1] = (uint32_t)(b >> 32);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void icnn_fwd2_kernel(const T* __restrict__ x, long long ld, long long N,
                                                        IcnnWeights<T> w, const T* __restrict__ AT,
                                                        const T* __restrict__ a, const uint32_t* __restrict__ M1,
                                                        T* __restrict__ U0, T* __restrict__ P) {
  __shared__ T Qs[kTileRows][3];
  __shared__ __attribute__((aligned(16))) T U1s[kW][kTileRows];
  __shared__ __attribute__((aligned(16))) T U0s[kW][kTileRows];
  __shared__ T Pp[4][kTileRows][3];
  const int j = threadIdx.x;
  const T wd0[3] = {w.Wd0[j], w.Wd0[kW + j], w.Wd0[2 * kW + j]};
  const T wd1[3] = {w.Wd1[j], w.Wd1[kW + j], w.Wd1[2 * kW + j]};
  const T aj = a[j];
  const long long tiles = (N + kTileRows - 1) / kTileRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kTileRows;
    __syncthreads();
    load_queries<T>(x, ld, w.pert, n0, N, Qs);
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) {
      const uint32_t word = (n0 + r < N) ? M1[(n0 + r) * kMaskWords + (j >> 5)] : 0u;
      U1s[j][r] = aj * T(mask_factor(word, j & 31));
    }
    __syncthreads();
    T acc[kTileRows];
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) acc[r] = T(0);
    for (int c = 0; c < kW; ++c) {  // V[r][k=j] = sum_c A[j][c] U1[r][c] = sum_c AT[c][j] U1[r][c]
      const T wc = AT[c * kW + j];
#pragma unroll
      for (int r = 0; r < kTileRows; ++r) acc[r] += U1s[c][r] * wc;
    }
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) {
      const T pre0 = Qs[r][0] * wd0[0] + Qs[r][1] * wd0[1] + Qs[r][2] * wd0[2];
      const T u0 = acc[r] * icnn_mask(pre0);
      U0s[j][r] = u0;
      if (n0 + r < N) U0[(n0 + r) * kW + j] = u0;
    }
    __syncthreads();
    // P[r][i] = sum_c Wd1[i][c] U1[r][c] + Wd0[i][c] U0[r][c]: 192 threads = (part 0..3, r, i), 64 columns each
    if (threadIdx.x < 4 * kTileRows * 3) {
      const int part = threadIdx.x / (kTileRows * 3), r = (threadIdx.x / 3) % kTileRows, i = threadIdx.x % 3;
      T s = T(0);
      for (int c = part * 64; c < part * 64 + 64; ++c) s += w.Wd1[i * kW + c] * U1s[c][r] + w.Wd0[i * kW + c] * U0s[c][r];
      Pp[part][r][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < kTileRows * 3) {
      const int r = threadIdx.x / 3, i = threadIdx.x % 3;
      if (n0 + r < N) P[(n0 + r) * 3 + i] = (Pp[0][r][i] + Pp[1][r][i]) + (Pp[2][r][i] + Pp[3][r][i]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void icnn_bwd1_kernel(const T* __restrict__ x, long long ld, long long N,
                                                        IcnnWeights<T> w, const T* __restrict__ A,
                                                        const T* __restrict__ a, const uint32_t* __restrict__ M1,
                                                        const T* __restrict__ U0, const T* __restrict__ RB,
                                                        double* __restrict__ partial) {
  __shared__ T Qs[kTileRows][3];
  __shared__ T Rs[kTileRows][3];
  __shared__ __attribute__((aligned(16))) T Vs[kW][kTileRows];
  const int j = threadIdx.x;
  const T wd0[3] = {w.Wd0[j], w.Wd0[kW + j], w.Wd0[2 * kW + j]};
  const T wd1[3] = {w.Wd1[j], w.Wd1[kW + j], w.Wd1[2 * kW + j]};
  const T aj = a[j];
  double abar = 0.0, g1[3] = {0.0, 0.0, 0.0}, g0[3] = {0.0, 0.0, 0.0};
  const long long tiles = (N + kTileRows - 1) / kTileRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kTileRows;
    __syncthreads();
    load_queries<T>(x, ld, w.pert, n0, N, Qs);
    if (threadIdx.x < kTileRows * 3) {
      const int r = threadIdx.x / 3, i = threadIdx.x % 3;
      Rs[r][i] = (n0 + r < N) ? RB[(n0 + r) * 3 + i] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) {
      const T pre0 = Qs[r][0] * wd0[0] + Qs[r][1] * wd0[1] + Qs[r][2] * wd0[2];
      Vs[j][r] = (Rs[r][0] * wd0[0] + Rs[r][1] * wd0[1] + Rs[r][2] * wd0[2]) * icnn_mask(pre0);
    }
    __syncthreads();
    T acc[kTileRows];
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) acc[r] = Rs[r][0] * wd1[0] + Rs[r][1] * wd1[1] + Rs[r][2] * wd1[2];
    for (int k = 0; k < kW; ++k) {  // U1b[r][j] += Vb[r][k] A[k][j]
      const T wk = A[k * kW + j];
#pragma unroll
      for (int r = 0; r < kTileRows; ++r) acc[r] += Vs[k][r] * wk;
    }
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) {
      if (n0 + r < N) {
        const uint32_t word = M1[(n0 + r) * kMaskWords + (j >> 5)];
        const T mf = T(mask_factor(word, j & 31));
        const T u1 = aj * mf;
        const T u0 = U0[(n0 + r) * kW + j];
        abar += double(acc[r] * mf);
#pragma unroll
        for (int i = 0; i < 3; ++i) { g1[i] += double(Rs[r][i] * u1); g0[i] += double(Rs[r][i] * u0); }
      }
    }
  }
  double* row = partial + (long long)blockIdx.x * kB1Cols;
  row[j] = abar;
#pragma unroll
  for (int i = 0; i < 3; ++i) { row[kW + i * kW + j] = g1[i]; row[4 * kW + i * kW + j] = g0[i]; }
}

// d|Wh|[k][j] = sum_n Vb[n][k] U1[n][j]; grid (16 k-tiles, n_slabs); thread j keeps 16 k's
constexpr int kB2Chunk = 64;
template <typename T>
__global__ __launch_bounds__(256) void icnn_bwd2_kernel(const T* __restrict__ x, long long ld, long long N,
                                                        IcnnWeights<T> w, const T* __restrict__ a,
                                                        const uint32_t* __restrict__ M1, const T* __restrict__ RB,
                                                        T* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) T Vs[kB2Chunk][16];
  const int j = threadIdx.x;
  const int k0 = blockIdx.x * 16;
  const long long per = (N + gridDim.y - 1) / gridDim.y;
  const long long n_begin = (long long)blockIdx.y * per, n_end = (n_begin + per < N) ? n_begin + per : N;
  const T aj = a[j];
  T acc[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc[kk] = T(0);
  for (long long c0 = n_begin; c0 < n_end; c0 += kB2Chunk) {
    __syncthreads();
    {  // 256 threads: 64 queries x 4 groups of 4 k's -> Vb[n][k0..k0+15]
      const int q = threadIdx.x >> 2, g = threadIdx.x & 3;
      const long long n = c0 + q;
      T val[4] = {T(0), T(0), T(0), T(0)};
      if (n < n_end) {
        T quat[4], qd[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) quat[i] = x[(n >> 2) * ld + i];
        icnn_query<T>(quat, w.pert + 3 * (n & 3), qd);
        const T r0 = RB[n * 3], r1 = RB[n * 3 + 1], r2 = RB[n * 3 + 2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = k0 + 4 * g + t;
          const T d0 = w.Wd0[k], d1 = w.Wd0[kW + k], d2 = w.Wd0[2 * kW + k];
          val[t] = (r0 * d0 + r1 * d1 + r2 * d2) * icnn_mask(qd[0] * d0 + qd[1] * d1 + qd[2] * d2);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) Vs[q][4 * g + t] = val[t];
    }
    __syncthreads();
    const int lim = (n_end - c0 < kB2Chunk) ? (int)(n_end - c0) : kB2Chunk;
    for (int q = 0; q < lim; ++q) {
      const uint32_t word = M1[(c0 + q) * kMaskWords + (j >> 5)];
      const T u1 = aj * T(mask_factor(word, j & 31));
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc[kk] += Vs[q][kk] * u1;
    }
  }
  T* slab = slabs + (long long)blockIdx.y * kW * kW;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) slab[(k0 + kk) * kW + j] = acc[kk];
}

// final fixed-order reduction + sign chain.  grad layout: [theta(10) | friction(2) | Wh | Wd0 | Wd1 | wout]
template <typename T>
__global__ __launch_bounds__(256) void icnn_reduce_kernel(IcnnWeights<T> w, const double* __restrict__ rows,
                                                          int n_rows, const double* __restrict__ b1, int b1_blocks,
                                                          const T* __restrict__ slabs, int n_slabs,
                                                          T* __restrict__ grad, T* __restrict__ loss_total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  constexpr int kHead = 12;
  if (idx < kW * kW) {
    double s = 0.0;
    for (int b = 0; b < n_slabs; ++b) s += double(slabs[(long long)b * kW * kW + idx]);
    const T raw = w.Wh[idx];
    grad[kHead + idx] = T(s * (raw > T(0) ? 1.0 : (raw < T(0) ? -1.0 : 0.0)));
  } else if (idx < kW * kW + 7 * kW) {
    const int c = (int)(idx - kW * kW);  // 0..767 Wd0, 768..1535 Wd1, 1536..1791 wout
    int col;
    if (c < 3 * kW) col = 4 * kW + c;            // dWd0 lives at [4W, 7W) of the b1 row
    else if (c < 6 * kW) col = kW + (c - 3 * kW); // dWd1 at [W, 4W)
    else col = c - 6 * kW;                       // d|wout| at [0, W)
    double s = 0.0;
    for (int b = 0; b < b1_blocks; ++b) s += b1[(long long)b * kB1Cols + col];
    if (c >= 6 * kW) {
      const T raw = w.wout[c - 6 * kW];
      s *= (raw > T(0) ? 1.0 : (raw < T(0) ? -1.0 : 0.0));
    }
    grad[kHead + kW * kW + c] = T(s);
  } else if (idx < kW * kW + 7 * kW + 16) {
    const int c = (int)(idx - kW * kW - 7 * kW);  // 0: loss, 1..12: theta, friction
    if (c <= kHead) {
      double s = 0.0;
      for (int r = 0; r < n_rows; ++r) s += rows[(long long)r * 16 + c];
      if (c == 0) {
        if (loss_total) *loss_total = T(s);
      } else {
        grad[c - 1] = T(s);
      }
    }
  }
}

}  // namespace
