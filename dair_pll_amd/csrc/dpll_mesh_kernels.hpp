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
//   icnn_bwd2      d|Wh| = Vb^T U1  (float: Vb (N x 256) is stored by icnn_bwd1)                GEMM  256 x 256 x N
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

// position of B-operand element (k, j) of a 256 x 256 matrix in the "fragment order" the MFMA row-tile kernels load
// their weight block in: wave j / 32 of a block keeps, per lane (half = k & 1, column j & 31), the 128 values k / 2 = 0..127;
// four consecutive ones (one float4) of all 64 lanes are 1 KB contiguous, so a wave fetches its 32 KB with 32 coalesced
// dwordx4 loads instead of 128 dword loads (two rounds of the 64-deep load queue: ~2 us of the kernels' fixed cost)
__device__ __forceinline__ int frag_index(int k, int j) {
  const int kk = k >> 1, lane = (k & 1) * 32 + (j & 31);
  return (((j >> 5) * 32 + (kk >> 2)) * 64 + lane) * 4 + (kk & 3);
}

template <typename T>
__global__ __launch_bounds__(256) void icnn_prep_kernel(IcnnWeights<T> w, T* __restrict__ A, T* __restrict__ AT,
                                                        T* __restrict__ a, T* __restrict__ Af, T* __restrict__ ATf) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < kW * kW) {
    const int k = idx / kW, j = idx % kW;
    const T v = tabs(w.Wh[idx]);
    A[idx] = v;
    AT[j * kW + k] = v;
    if (Af) {  // float path: the same two matrices in fragment order
      Af[frag_index(k, j)] = v;
      ATf[frag_index(j, k)] = v;
    }
  }
  if (idx < kW) a[idx] = tabs(w.wout[idx]);
}

// query directions of a tile of rows into LDS; state row of item i starts at x + i * ld, quaternion first
// direction of query n: read from `dirs` when given, else from the item's quaternion and the perturbation row (4 per item)
template <typename T>
__device__ __forceinline__ void query_direction(const T* __restrict__ x, long long ld, const IcnnWeights<T>& w, long long n, T (&q)[3]) {
  if (w.dirs) {
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = w.dirs[3 * n + i];
    return;
  }
  T quat[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) quat[i] = x[(n >> 2) * ld + i];
  icnn_query<T>(quat, w.pert + 3 * (n & 3), q);
}
template <typename T>
__device__ __forceinline__ void load_queries(const T* __restrict__ x, long long ld, const IcnnWeights<T>& w,
                                             long long n0, long long N, T (*Qs)[3]) {
  if (threadIdx.x < kTileRows) {
    const long long n = n0 + threadIdx.x;
    T q[3] = {T(0), T(0), T(1)};
    if (n < N) query_direction<T>(x, ld, w, n, q);
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
    load_queries<T>(x, ld, w, n0, N, Qs);
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
  const T aj = a[j];
  const long long tiles = (N + kTileRows - 1) / kTileRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kTileRows;
    __syncthreads();
    load_queries<T>(x, ld, w, n0, N, Qs);
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
      if (n0 + r < N) P[icnn_point_index(n0 + r, w) + i] = (Pp[0][r][i] + Pp[1][r][i]) + (Pp[2][r][i] + Pp[3][r][i]);
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
    load_queries<T>(x, ld, w, n0, N, Qs);
    if (threadIdx.x < kTileRows * 3) {
      const int r = threadIdx.x / 3, i = threadIdx.x % 3;
      Rs[r][i] = (n0 + r < N) ? RB[icnn_point_index(n0 + r, w) + i] : T(0);
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
        T qd[3];
        query_direction<T>(x, ld, w, n, qd);
        const long long pn = icnn_point_index(n, w);
        const T r0 = RB[pn], r1 = RB[pn + 1], r2 = RB[pn + 2];
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
// The sums are over 64 slabs of 256 KB and 256 partial rows: a block owns 64 consecutive outputs and splits the
// summands over 16 thread groups (thread = 4 consecutive outputs x every 16th summand, all of its loads in flight
// at once), then combines the 16 groups through LDS in a fixed order => bitwise reproducible at full bandwidth.
constexpr int kRedOut = 64;     // outputs per block
constexpr int kRedGroups = 16;  // summand groups per block
constexpr int kRedWhBlocks = kW * kW / kRedOut;
constexpr int kRedB1Blocks = 7 * kW / kRedOut;
constexpr int kRedBlocks = kRedWhBlocks + kRedB1Blocks + 1;

template <typename S>
__device__ __forceinline__ void reduce_strided4(const S* __restrict__ base, long long stride, int n, int first, double (&acc)[4]) {
  // summands first, first + 16, ... < n of 4 consecutive columns; unrolled by 4 so that 4 vector loads are in flight
  struct alignas(4 * sizeof(S)) V4 { S v[4]; };
  int s = first;
  // eight vector loads in flight (the 256 partial rows of icnn_bwd1 are 16 summands per thread: two round trips, not four)
  for (; s + 7 * kRedGroups < n; s += 8 * kRedGroups) {
    V4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const V4*)(base + (long long)(s + u * kRedGroups) * stride);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[i] += ((double(v[0].v[i]) + double(v[1].v[i])) + (double(v[2].v[i]) + double(v[3].v[i]))) +
                ((double(v[4].v[i]) + double(v[5].v[i])) + (double(v[6].v[i]) + double(v[7].v[i])));
  }
  for (; s + 3 * kRedGroups < n; s += 4 * kRedGroups) {
    const V4 a = *(const V4*)(base + (long long)s * stride);
    const V4 b = *(const V4*)(base + (long long)(s + kRedGroups) * stride);
    const V4 c = *(const V4*)(base + (long long)(s + 2 * kRedGroups) * stride);
    const V4 d = *(const V4*)(base + (long long)(s + 3 * kRedGroups) * stride);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += (double(a.v[i]) + double(b.v[i])) + (double(c.v[i]) + double(d.v[i]));
  }
  for (; s < n; s += kRedGroups) {
    const V4 a = *(const V4*)(base + (long long)s * stride);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += double(a.v[i]);
  }
}

// NB bodies, each with its own network: one launch per network.  `grad_w` = where this network's weight gradients go
// ([Wh | Wd0 | Wd1 | wout], 67,328 entries); `grad_head` (first network's launch only, else nullptr) = the head of the
// gradient, [theta (10 NB) | friction (1 + NB)], chained from the loss kernel's iota-space rows (row stride
// `row_stride`, the chain matrix behind the rows).
template <typename T, int NB>
__global__ __launch_bounds__(256) void icnn_reduce_kernel(IcnnWeights<T> w, const double* __restrict__ rows,
                                                          int n_rows, int row_stride, const double* __restrict__ b1, int b1_blocks,
                                                          const T* __restrict__ slabs, int n_slabs,
                                                          T* __restrict__ grad_w, T* __restrict__ grad_head, T* __restrict__ loss_total,
                                                          AdamArgs adam, long long w_offset) {
  // adam (the fused training step, dpll_contactnets_train_step_mesh): the thread that writes a gradient entry applies Adam to
  // its parameter -- entry i of grad_w is parameter w_offset + i of the flat buffer, entry k of grad_head parameter k.  The
  // optimizer state is only READ here (several launches, many workgroups per step): adam_advance_kernel moves it once.
  constexpr int kHead = 10 * NB + 1 + NB;     // theta, friction
  constexpr int kCols = 1 + 10 * NB + 4 * NB; // columns of a row: [loss | iota | mu_pair | lengths (unused)]
  static_assert(kCols <= 32 && kHead < 64, "head must fit the wave that chains it");
  __shared__ double red[kRedGroups][kRedOut + 1];
  const int cg = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int b = blockIdx.x;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (b < kRedWhBlocks) {
    reduce_strided4<T>(slabs + (long long)b * kRedOut + 4 * cg, (long long)kW * kW, n_slabs, sg, acc);
  } else if (b < kRedWhBlocks + kRedB1Blocks) {
    // output c: 0..767 Wd0 (b1 columns [4W, 7W)), 768..1535 Wd1 ([W, 4W)), 1536..1791 wout ([0, W)); 64 | 768, 256
    const int c0 = (b - kRedWhBlocks) * kRedOut;
    const int col0 = c0 < 3 * kW ? 4 * kW + c0 : (c0 < 6 * kW ? kW + (c0 - 3 * kW) : c0 - 6 * kW);
    reduce_strided4<double>(b1 + col0 + 4 * cg, (long long)kB1Cols, b1_blocks, sg, acc);
  } else if (grad_head && cg < 8) {  // head: up to 32 columns of the loss kernel's rows = 8 column groups (scalar loads: odd strides)
    // (clamped addresses, eight rows' loads in flight, the values masked afterwards: a 256-row launch is two round trips)
    for (int r0 = sg; r0 < n_rows; r0 += 8 * kRedGroups) {
      double v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u * kRedGroups < n_rows ? r0 + u * kRedGroups : n_rows - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[u][i] = rows[(long long)r * row_stride + (4 * cg + i < kCols ? 4 * cg + i : 0)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += (r0 + u * kRedGroups < n_rows && 4 * cg + i < kCols) ? v[u][i] : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[sg][4 * cg + i] = acc[i];
  __syncthreads();
  if (threadIdx.x >= kRedOut) return;
  double s = 0.0;
#pragma unroll
  for (int g = 0; g < kRedGroups; ++g) s += red[g][threadIdx.x];
  double steps = 0.0, pow1 = 0.0, pow2 = 0.0;
  if (adam.params) adam_powers(adam, steps, pow1, pow2);
  if (b < kRedWhBlocks) {
    const long long idx = (long long)b * kRedOut + threadIdx.x;
    const T raw = w.Wh[idx];
    const T gv = T(s * (raw > T(0) ? 1.0 : (raw < T(0) ? -1.0 : 0.0)));
    grad_w[idx] = gv;
    if (adam.params) adam_apply<T>(adam, w_offset + idx, double(gv), pow1, pow2);
  } else if (b < kRedWhBlocks + kRedB1Blocks) {
    const int c = (b - kRedWhBlocks) * kRedOut + threadIdx.x;
    if (c >= 6 * kW) {
      const T raw = w.wout[c - 6 * kW];
      s *= (raw > T(0) ? 1.0 : (raw < T(0) ? -1.0 : 0.0));
    }
    grad_w[kW * kW + c] = T(s);
    if (adam.params) adam_apply<T>(adam, w_offset + kW * kW + c, double(T(s)), pow1, pow2);
  } else if (grad_head) {
    // head: the rows are in iota space, the chain matrix to (theta, friction) sits behind them; threads 0..63 are one
    // wave: every lane gathers the column totals, thread 1 + k chains parameter k
    double tot[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) tot[i] = i < kCols ? __shfl(s, i) : 0.0;
    const int k = (int)threadIdx.x - 1;
    if (threadIdx.x == 0) {
      if (loss_total) *loss_total = T(s);
    } else if (k < kHead) {
      const T gv = T(apply_chain<NB, NB>(tot, rows + (long long)n_rows * row_stride, k));
      grad_head[k] = gv;
      if (adam.params) adam_apply<T>(adam, k, double(gv), pow1, pow2);
    }
  }
}
// the optimizer state [steps, beta1^steps, beta2^steps] one step on: after the reduce launches of every network have read it
__global__ void adam_advance_kernel(AdamArgs adam) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double steps, pow1, pow2;
    adam_powers(adam, steps, pow1, pow2);
    adam.state[0] = steps; adam.state[1] = pow1; adam.state[2] = pow2;
  }
}

}  // namespace

// =================================================================================================
// MFMA forms (float): v_mfma_f32_32x32x2_f32, exact f32 FMA chains.  Operand maps (cdna guide, section 3):
//   A: lane l holds A[i = l & 31][k = l >> 5]     B: lane l holds B[k = l >> 5][j = l & 31]
//   C/D: 16 registers, col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)
// Row-tile kernels: 512 threads = 8 waves, wave v owns output columns [32 v, 32 v + 32) of a 32-row tile; the
// X tile sits in LDS with a 257-float row stride (conflict-free column reads), the 256 x 32 column block of
// the weight matrix is held in 128 VGPRs per lane for the whole launch.
// =================================================================================================
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kMfmaRows = 32;
constexpr int kXs = kW + 1;  // padded LDS row stride of tiles that are not MFMA operands
// LDS image of a 32 x 256 MFMA A-operand tile: element (row, k) at xop(row, k); the four values lane (row, half)
// feeds to the steps 4 kq .. 4 kq + 3 (k = 2 (4 kq + e) + half) are one aligned float4, the 64 lanes of a wave read a
// contiguous 1 KB with one ds_read_b128 per four MFMAs; the 8-float pad per kq keeps the column-wise fills
// (64 consecutive k of one row per wave) on 64 distinct banks.
constexpr int kXq = 32 * 8 + 8;
constexpr int kXopFloats = 32 * kXq;
__device__ __forceinline__ int xop(int row, int k) { return (k >> 3) * kXq + row * 8 + (k & 1) * 4 + ((k >> 1) & 3); }

__device__ __forceinline__ int mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// "Operand-tile" layout of an (N, 256) activation matrix for icnn_bwd2_mfma: 4 KB blocks (row tile t of 32 rows,
// column tile c of 32 columns) at float offset (t * 8 + c) * 1024, inside a block the float4 with index
// (q * 32 + col) * 2 + half holds rows 2 (4 q + e) + half, e = 0..3, of column col -- exactly what lane
// (col, half) feeds to MFMA steps 4 q .. 4 q + 3, so a wave fetches four steps with one contiguous 1 KB read.
// A thread of the tile kernels owns column c and rows r0 .. r0 + 15 (r0 = 0 or 16): four float4 stores.
template <class ValueOfRow>
__device__ __forceinline__ void fill_operand_tile(float* __restrict__ base, long long tile, int c, int r0, float* __restrict__ Xs,
                                                  ValueOfRow value_of_row) {
  // rows in float4 order (four values live at a time): LDS image Xs in the xop layout, global operand tile if `base`
  f32x4* blk = base ? (f32x4*)(base + (tile * 8 + (c >> 5)) * 1024) : nullptr;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      f32x4 val;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rr = r0 + 2 * (4 * qq + e) + h;
        val[e] = value_of_row(rr);
        Xs[xop(rr, c)] = val[e];
      }
      if (blk) blk[((r0 / 8 + qq) * 32 + (c & 31)) * 2 + h] = val;
    }
}

// queries of a 32-row tile into LDS (threads 0..31)
__device__ __forceinline__ void load_queries32(const float* __restrict__ x, long long ld, const IcnnWeights<float>& w,
                                               long long n0, long long N, float (*Qs)[3]) {
  if (threadIdx.x < kMfmaRows) {
    const long long n = n0 + threadIdx.x;
    float q[3] = {0.f, 0.f, 1.f};
    if (n < N) query_direction<float>(x, ld, w, n, q);
#pragma unroll
    for (int i = 0; i < 3; ++i) Qs[threadIdx.x][i] = q[i];
  }
}

// this lane's 128 weights of its wave's 256 x 32 column block, from a matrix in fragment order (frag_index)
__device__ __forceinline__ void load_weight_fragment(const float* __restrict__ F, int wv, int lane, float (&bfrag)[kW / 2]) {
  const f32x4* f = (const f32x4*)F + (wv * 32) * 64 + lane;
#pragma unroll
  for (int q = 0; q < kW / 8; ++q) {
    const f32x4 v = f[q * 64];
#pragma unroll
    for (int e = 0; e < 4; ++e) bfrag[4 * q + e] = v[e];
  }
}

// C(32 x 32 of this wave) = Xs(32 x 256, LDS in the xop layout) * Wfrag(256 x 32, registers)
__device__ __forceinline__ f32x16 mfma_tile(const float* __restrict__ Xs, const float (&bfrag)[kW / 2], int l31, int half) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const f32x4* xq = (const f32x4*)(Xs + l31 * 8 + half * 4);
#pragma unroll
  for (int kq = 0; kq < kW / 8; ++kq) {
    const f32x4 x4 = xq[kq * (kXq / 4)];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4[e], bfrag[4 * kq + e], acc, 0, 0, 0);
  }
  return acc;
}

__global__ __launch_bounds__(512) void icnn_fwd1_mfma(const float* __restrict__ x, long long ld, long long N,
                                                      IcnnWeights<float> w, const float* __restrict__ A,
                                                      uint32_t* __restrict__ M1) {
  __shared__ float Qs[kMfmaRows][3];
  __shared__ __attribute__((aligned(16))) float Xs[kXopFloats];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int col = 32 * wv + l31;
  float bfrag[kW / 2];
  load_weight_fragment(A, wv, lane, bfrag);  // A: |Wh| in fragment order
  const float wd1[3] = {w.Wd1[col], w.Wd1[kW + col], w.Wd1[2 * kW + col]};
  const long long tiles = (N + kMfmaRows - 1) / kMfmaRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kMfmaRows;
    __syncthreads();
    load_queries32(x, ld, w, n0, N, Qs);
    __syncthreads();
    {  // H0 tile: thread -> column c = t & 255, rows (t >> 8) * 16 ..
      const int c = threadIdx.x & 255, r0 = (threadIdx.x >> 8) * 16;
      const float d0 = w.Wd0[c], d1 = w.Wd0[kW + c], d2 = w.Wd0[2 * kW + c];
#pragma unroll
      for (int r = 0; r < 16; ++r) Xs[xop(r0 + r, c)] = icnn_act(Qs[r0 + r][0] * d0 + Qs[r0 + r][1] * d1 + Qs[r0 + r][2] * d2);
    }
    __syncthreads();
    const f32x16 acc = mfma_tile(Xs, bfrag, l31, half);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mfma_row(reg, half);
      const float pre1 = acc[reg] + Qs[row][0] * wd1[0] + Qs[row][1] * wd1[1] + Qs[row][2] * wd1[2];
      const unsigned long long b = __ballot(pre1 > 0.f);
      // low word: rows of half 0, high word: rows of half 1 (= +4)
      if (lane == 0) {
        const int ra = mfma_row(reg, 0), rb = mfma_row(reg, 1);
        if (n0 + ra < N) M1[(n0 + ra) * kMaskWords + wv] = (uint32_t)(b & 0xffffffffull);
        if (n0 + rb < N) M1[(n0 + rb) * kMaskWords + wv] = (uint32_t)(b >> 32);
      }
    }
  }
}

// U0T: U0 leaves in the accumulator layout the pipelined icnn_bwd1 (dpll_icnn_pipe.hip) reads -- block (tile, wave pair), per lane
// the float4 (4 (wave & 1) + reg / 4) -- instead of row-major
template <bool U0T>
__global__ __launch_bounds__(512) void icnn_fwd2_mfma(const float* __restrict__ x, long long ld, long long N,
                                                      IcnnWeights<float> w, const float* __restrict__ AT,
                                                      const float* __restrict__ a, const uint32_t* __restrict__ M1,
                                                      float* __restrict__ U0, float* __restrict__ P,
                                                      float* __restrict__ U1t) {
  __shared__ float Qs[kMfmaRows][3];
  __shared__ __attribute__((aligned(16))) float Xs[kXopFloats];   // U1 tile, xop layout (also written as operand tiles `U1t` when the backward will run)
  __shared__ float Ys[kMfmaRows * kXs];   // U0 tile
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int col = 32 * wv + l31;
  float bfrag[kW / 2];
  load_weight_fragment(AT, wv, lane, bfrag);  // AT: |Wh|^T in fragment order
  const float wd0[3] = {w.Wd0[col], w.Wd0[kW + col], w.Wd0[2 * kW + col]};
  // input weights of both layers by column for the support-point product below: [c] = (Wd1[0..2][c], 0 | Wd0[0..2][c], 0);
  // read back as two float4 per column (the 96 global loads per thread and tile they replace kept the vector-memory
  // pipe busier than the MFMAs)
  __shared__ f32x4 Wds[kW][2];
  if (threadIdx.x < kW) {
    const int c = threadIdx.x;
    Wds[c][0] = f32x4{w.Wd1[c], w.Wd1[kW + c], w.Wd1[2 * kW + c], 0.f};
    Wds[c][1] = f32x4{w.Wd0[c], w.Wd0[kW + c], w.Wd0[2 * kW + c], 0.f};
  }
  const long long tiles = (N + kMfmaRows - 1) / kMfmaRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kMfmaRows;
    __syncthreads();
    load_queries32(x, ld, w, n0, N, Qs);
    {  // U1 tile
      const int c = threadIdx.x & 255, r0 = (threadIdx.x >> 8) * 16;
      const float ac = a[c];
      fill_operand_tile(U1t, tile, c, r0, Xs, [&](int rr) {
        const uint32_t word = (n0 + rr < N) ? M1[(n0 + rr) * kMaskWords + (c >> 5)] : 0u;
        return ac * mask_factor(word, c & 31);
      });
    }
    __syncthreads();
    const f32x16 acc = mfma_tile(Xs, bfrag, l31, half);
    f32x4 u0t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mfma_row(reg, half);
      const float pre0 = Qs[row][0] * wd0[0] + Qs[row][1] * wd0[1] + Qs[row][2] * wd0[2];
      const float u0 = acc[reg] * icnn_mask(pre0);
      Ys[row * kXs + col] = u0;
      if (U0T) {
        u0t[reg & 3] = u0;
        if ((reg & 3) == 3) ((f32x4*)(U0 + (tile * 4 + (wv >> 1)) * 2048))[(4 * (wv & 1) + (reg >> 2)) * 64 + lane] = u0t;
      } else if (n0 + row < N) {
        U0[(n0 + row) * kW + col] = u0;
      }
    }
    __syncthreads();
    {  // P[row][i]: thread -> (row = t >> 4, part = t & 15), columns part + 16 m
      const int row = threadIdx.x >> 4, part = threadIdx.x & 15;
      float s[3] = {0.f, 0.f, 0.f};
#pragma unroll 2  // (the weight fragment leaves few registers: a full unroll hoists 32 float4 loads and spills)
      for (int m = 0; m < 16; ++m) {
        const int c = part + 16 * m;
        const float u1 = Xs[xop(row, c)], u0 = Ys[row * kXs + c];
        const f32x4 w1 = Wds[c][0], w0 = Wds[c][1];
#pragma unroll
        for (int i = 0; i < 3; ++i) s[i] += w1[i] * u1 + w0[i] * u0;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {  // sum over the 16 parts = one DPP row
        float v = s[i];
        v += dpp_mov<kQuadXor1>(v);
        v += dpp_mov<kQuadXor2>(v);
        v += dpp_mov<kRowHalfMirror>(v);
        v += dpp_mov<kRowMirror>(v);
        if (part == 0 && n0 + row < N) P[icnn_point_index(n0 + row, w) + i] = v;
      }
    }
  }
}

__global__ __launch_bounds__(512) void icnn_bwd1_mfma(const float* __restrict__ x, long long ld, long long N,
                                                      IcnnWeights<float> w, const float* __restrict__ A,
                                                      const float* __restrict__ a, const uint32_t* __restrict__ M1,
                                                      const float* __restrict__ U0, const float* __restrict__ RB,
                                                      double* __restrict__ partial, float* __restrict__ Vb) {
  __shared__ float Qs[kMfmaRows][3];
  __shared__ float Rs[kMfmaRows][3];
  __shared__ __attribute__((aligned(16))) float Xs[kXopFloats];  // Vb tile, xop layout (also written to `Vb` as operand tiles for icnn_bwd2_mfma)
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int col = 32 * wv + l31;
  float bfrag[kW / 2];
  load_weight_fragment(A, wv, lane, bfrag);  // A: |Wh| in fragment order
  const float wd1[3] = {w.Wd1[col], w.Wd1[kW + col], w.Wd1[2 * kW + col]};
  const float acol = a[col];
  double abar = 0.0, g1[3] = {0.0, 0.0, 0.0}, g0[3] = {0.0, 0.0, 0.0};
  const long long tiles = (N + kMfmaRows - 1) / kMfmaRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kMfmaRows;
    __syncthreads();
    load_queries32(x, ld, w, n0, N, Qs);
    if (threadIdx.x >= 64 && threadIdx.x < 64 + kMfmaRows * 3) {
      const int t = threadIdx.x - 64, r = t / 3, i = t % 3;
      Rs[r][i] = (n0 + r < N) ? RB[icnn_point_index(n0 + r, w) + i] : 0.f;
    }
    __syncthreads();
    {  // Vb tile
      const int c = threadIdx.x & 255, r0 = (threadIdx.x >> 8) * 16;
      const float d0 = w.Wd0[c], d1 = w.Wd0[kW + c], d2 = w.Wd0[2 * kW + c];
      fill_operand_tile(Vb, tile, c, r0, Xs, [&](int rr) {
        const float pre0 = Qs[rr][0] * d0 + Qs[rr][1] * d1 + Qs[rr][2] * d2;
        return (Rs[rr][0] * d0 + Rs[rr][1] * d1 + Rs[rr][2] * d2) * icnn_mask(pre0);  // rows past N: RB = 0
      });
    }
    __syncthreads();
    // epilogue operands first (independent loads in flight during the MFMAs)
    float mfv[16], u0v[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mfma_row(reg, half);
      const bool ok = n0 + row < N;
      mfv[reg] = ok ? mask_factor(M1[(n0 + row) * kMaskWords + wv], l31) : 0.f;
      u0v[reg] = ok ? U0[(n0 + row) * kW + col] : 0.f;
    }
    const f32x16 acc = mfma_tile(Xs, bfrag, l31, half);
    float t_abar = 0.f, t1[3] = {0.f, 0.f, 0.f}, t0[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mfma_row(reg, half);
      const float u1b = acc[reg] + Rs[row][0] * wd1[0] + Rs[row][1] * wd1[1] + Rs[row][2] * wd1[2];
      t_abar += u1b * mfv[reg];  // rows past N: mask factor 0 and RB = 0
      const float u1 = acol * mfv[reg];
#pragma unroll
      for (int i = 0; i < 3; ++i) { t1[i] += Rs[row][i] * u1; t0[i] += Rs[row][i] * u0v[reg]; }
    }
    abar += double(t_abar);
#pragma unroll
    for (int i = 0; i < 3; ++i) { g1[i] += double(t1[i]); g0[i] += double(t0[i]); }
  }
  // the two halves of the wave hold different rows of the same column
  abar += __shfl_xor(abar, 32);
#pragma unroll
  for (int i = 0; i < 3; ++i) { g1[i] += __shfl_xor(g1[i], 32); g0[i] += __shfl_xor(g0[i], 32); }
  if (half == 0) {
    double* row = partial + (long long)blockIdx.x * kB1Cols;
    row[col] = abar;
#pragma unroll
    for (int i = 0; i < 3; ++i) { row[kW + i * kW + col] = g1[i]; row[4 * kW + i * kW + col] = g0[i]; }
  }
}

// d|Wh|[k][j] = sum_n Vb[n][k] U1[n][j] as a split-K GEMM.  Grid (4 = k-half x j-half, n_slabs): a block owns a 128 x 128
// piece over its slab of row tiles; wave v the k-tile v >> 1 and the two j-tiles 2 (v & 1), 2 (v & 1) + 1 of the piece (two
// accumulator sets).  Vb comes as the operand tiles icnn_bwd1_mfma left behind (global -> registers -> LDS, double buffered:
// the next tile's loads are in flight during this tile's 32 MFMAs per wave, one barrier per tile).  U1 = |wout| (mask ? 1 :
// 1/2) has two values per column, so it is NOT stored (round 2 wrote and re-read 16 MB of U1 operand tiles per call): a lane
// rebuilds its B elements from the tile's mask words in LDS -- a broadcast read, a shift and a select per MFMA, well inside
// the 16 issue slots an f32 MFMA leaves.
constexpr int kB2Pieces = 4;
__global__ __launch_bounds__(512) void icnn_bwd2_mfma(long long N, const float* __restrict__ VbT, const uint32_t* __restrict__ M1,
                                                      const float* __restrict__ a, float* __restrict__ slabs) {
  __shared__ f32x4 Ls[2][4][256];  // [buffer][Vb k-tile of the piece][4 KB operand tile]
  __shared__ uint32_t Ms[2][kMfmaRows][4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int kq = blockIdx.x & 1, jq = blockIdx.x >> 1;
  const int ktl = wv >> 1, jtl = 2 * (wv & 1);
  const long long n_tiles = (N + kMfmaRows - 1) / kMfmaRows;
  const long long per = (n_tiles + gridDim.y - 1) / gridDim.y;
  const long long t_begin = (long long)blockIdx.y * per, t_end = (t_begin + per < n_tiles) ? t_begin + per : n_tiles;
  const float a0 = a[128 * jq + 32 * jtl + l31], a1 = a[128 * jq + 32 * (jtl + 1) + l31];
  const float h0 = a0 * float(kIcnnSlope), h1 = a1 * float(kIcnnSlope);
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  // copy assignment: float4 number threadIdx.x + 512 m of the chunk's 1024 (slot = number >> 8, index = number & 255)
  f32x4 stage[2];
  uint32_t mstage = 0u;
  auto fetch = [&](long long t) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int slot = 2 * m + (threadIdx.x >> 8), idx = threadIdx.x & 255;
      stage[m] = ((const f32x4*)(VbT + (t * 8 + 4 * kq + slot) * 1024))[idx];
    }
    if (threadIdx.x < kMfmaRows * 4) {
      const long long n = t * kMfmaRows + (threadIdx.x >> 2);
      mstage = n < N ? M1[n * kMaskWords + 4 * jq + (threadIdx.x & 3)] : 0u;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int m = 0; m < 2; ++m) Ls[buf][2 * m + (threadIdx.x >> 8)][threadIdx.x & 255] = stage[m];
    if (threadIdx.x < kMfmaRows * 4) Ms[buf][threadIdx.x >> 2][threadIdx.x & 3] = mstage;
  };
  if (t_begin < t_end) { fetch(t_begin); commit(0); }
  __syncthreads();
  for (long long t = t_begin; t < t_end; ++t) {
    const int cur = (int)((t - t_begin) & 1);
    const bool more = t + 1 < t_end;
    if (more) fetch(t + 1);
    // this lane's B elements of the tile's 16 MFMA steps, all rebuilt before the first MFMA (row of step i: 2 i + half)
    float b0[16], b1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t w0 = Ms[cur][2 * i + half][jtl], w1 = Ms[cur][2 * i + half][jtl + 1];
      b0[i] = ((w0 >> l31) & 1u) ? a0 : h0;
      b1[i] = ((w1 >> l31) & 1u) ? a1 : h1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 a4 = Ls[cur][ktl][(q * 32 + l31) * 2 + half];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b0[4 * q + e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b1[4 * q + e], acc1, 0, 0, 0);
      }
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
  }
  float* slab = slabs + (long long)blockIdx.y * kW * kW;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int k = 128 * kq + 32 * ktl + mfma_row(reg, half);
    slab[k * kW + 128 * jq + 32 * jtl + l31] = acc0[reg];
    slab[k * kW + 128 * jq + 32 * (jtl + 1) + l31] = acc1[reg];
  }
}

}  // namespace
