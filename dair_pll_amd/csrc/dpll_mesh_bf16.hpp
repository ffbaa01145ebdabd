// dpll_mesh_bf16.hpp -- the ICNN GEMM kernels on the bf16 matrix cores with SPLIT operands; included by dpll_kernels.hip
// behind dpll_mesh_kernels.hpp.  Forms of the mesh pipeline's GEMMs (dpll_solver_opts_t.mesh_gemm): 2 / 3 bf16 planes (below) and, since
// the end of round 5 the DEFAULT, 4 = two fp16 planes (further down: f32-grade products at the 2-plane cost).  The f32 MFMA kernels
// (mesh_gemm = 0 / 1) remain selectable and are what every earlier round measured.
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 rate (64 cycles for 4 kflop against 32 cycles for 32 kflop).  An
// f32 number is the exact sum of three bf16 numbers (8 + 8 + 8 significand bits), x = x1 + x2 + x3, and a product of two
// bf16 numbers is exact in f32, so
//     x y  =  x1 y1 + (x1 y2 + x2 y1) + (x1 y3 + x2 y2 + x3 y1) + O(2^-24 |x y|)
// PLANES = 3: those six products per k-step (6 x 32 cycles against 8 x 64 of the f32 form: 2.7 x fewer matrix-core cycles)
//             at f32-grade accuracy -- the dropped terms are of the size of one f32 rounding;
// PLANES = 2: x = x1 + x2 to 16 bits, three products (the "bf16 x 3" form: 5.3 x fewer cycles), relative error 2^-16 per
//             product.
// The small terms go to a second accumulator (added to the main one once per tile), so they are not rounded away against
// the running sum.  What changes against the f32 kernels is the ORDER and grouping of the roundings, not the operands'
// precision (PLANES = 3), so a hidden unit within a few ulp of zero can take the other LeakyReLU branch: DESIGN.md 5a has
// the measured flip rates and loss / gradient differences against the float64 path.
#pragma once

#include "dpll_mesh_kernels.hpp"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int PL> __device__ __forceinline__ void split_bf16(float x, __bf16 (&out)[PL]) {
  float r = x;
#pragma unroll
  for (int p = 0; p < PL; ++p) {
    out[p] = (__bf16)r;  // round to nearest even; the residual below is exact in f32
    r -= (float)out[p];
  }
}

// ---- two fp16 planes (dpll_solver_opts_t.mesh_gemm = 4).  fp16 keeps 11 significand bits against bf16's 8: x = h + l with
// h = fp16(x) and l = fp16(x - h) leaves |x - (h + l)| <= 2^-24 |x| -- the rounding of an f32 number itself -- so THREE products per
// k-step (h w_h, h w_l + l w_h; the dropped l w_l is 2^-24) give f32-grade products at the 2-plane cost.  The price is fp16's
// range: the low plane is stored scaled by 2^11 (so it is as far from the subnormals as the high plane; its products go to the
// `low` accumulator, which is scaled back once), and operands must stay below 65504 in magnitude -- the network's weights,
// activations of unit directions and |wout| are O(1); see DESIGN.md 5 for what is checked.  The planes travel in the same 16-bit
// containers as the bf16 ones (fragment order, LDS image): only the split and the MFMA instruction differ.
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kF16LowScale = 2048.f;  // 2^11
template <bool F16> __device__ __forceinline__ __bf16 enc16(float v) {
  if constexpr (F16) return __builtin_bit_cast(__bf16, (_Float16)v);
  else return (__bf16)v;
}
template <bool F16> __device__ __forceinline__ float dec16(__bf16 b) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, b);
  else return (float)b;
}
template <int PL, bool F16> __device__ __forceinline__ void split_planes(float x, __bf16 (&out)[PL]) {
  if constexpr (F16) {
    static_assert(PL == 2, "the fp16 form has two planes");
    out[0] = enc16<true>(x);
    out[1] = enc16<true>((x - dec16<true>(out[0])) * kF16LowScale);  // (the residual is exact in f32, the scaling a power of two)
  } else {
    split_bf16<PL>(x, out);
  }
}
template <bool F16> __device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> constexpr float kLowBack = F16 ? 1.f / kF16LowScale : 1.f;  // what the `low` accumulator is multiplied by at the end

// LDS image of one plane of a 32 x 256 A-operand tile for v_mfma_f32_32x32x16_bf16: lane (row r, half h) feeds k = 16 s +
// 8 h + j, j = 0..7, of step s -- one aligned 16-byte read; element (row, k) at bop(row, k).  The 8-element pad per k-block
// puts the two halves of a wave (blocks 2 s and 2 s + 1) on different banks.
constexpr int kBq = 32 * 8 + 8;
constexpr int kBopElems = 32 * kBq;
__device__ __forceinline__ int bop(int row, int k) { return (k >> 3) * kBq + row * 8 + (k & 7); }

// position of B-operand element (k, j) of a 256 x 256 matrix in the bf16 "fragment order": wave j / 32 keeps, per lane
// (half = (k >> 3) & 1, column j & 31) and step s = k / 16, the 8 values k & 7 as one 16-byte vector
__device__ __forceinline__ int frag_index_bf16(int k, int j) {
  const int lane = ((k >> 3) & 1) * 32 + (j & 31);
  return (((j >> 5) * 16 + (k >> 4)) * 64 + lane) * 8 + (k & 7);
}

// |Wh| and |Wh|^T as PL bf16 planes in fragment order (plane p at offset p * 65536)
template <int PL, bool F16 = false>
__global__ __launch_bounds__(256) void icnn_prep_bf16_kernel(IcnnWeights<float> w, __bf16* __restrict__ Af, __bf16* __restrict__ ATf,
                                                             float* __restrict__ a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  // fp16's range, guarded LOUDLY (F16): a weight at or beyond 2^14 (Wd0: 2^12) -- or not a number -- would send an operand plane to
  // infinity somewhere down the pipeline.  The thread that meets one writes NaN instead of its plane entries (|wout|: instead of
  // the entry of `a`): the second forward GEMM then returns NaN support points, every solve fails and is masked -- never a finite
  // wrong number.  (Activations of unit directions are bounded by 3 max |Wd0|, U1 by |wout|.)
  if (idx < kW) {
    float av = fabsf(w.wout[idx]);
    if constexpr (F16) av = av < 16384.f ? av : __builtin_nanf("");
    a[idx] = av;
  }
  // (fp16 planes: the 256 words where the workgroups of icnn_bwd1 leave the largest |r_bar| they met for icnn_bwd2 -- the place of
  // a third plane of Af, which the two-plane forms do not have -- start every forward at zero)
  if constexpr (F16) { if (idx < 256) reinterpret_cast<unsigned*>(Af + 2 * kW * kW)[idx] = 0u; }
  if (idx >= kW * kW) return;
  const int k = idx / kW, j = idx % kW;
  __bf16 parts[PL];
  float wv = fabsf(w.Wh[idx]);
  if constexpr (F16) {
    bool bad = !(wv < 16384.f);
    if (idx < 3 * kW) bad = bad || !(fabsf(w.Wd0[idx]) < 4096.f) || !(fabsf(w.Wd1[idx]) < 16384.f);
    if (bad) wv = __builtin_nanf("");
  }
  split_planes<PL, F16>(wv, parts);
#pragma unroll
  for (int p = 0; p < PL; ++p) {
    Af[p * kW * kW + frag_index_bf16(k, j)] = parts[p];
    ATf[p * kW * kW + frag_index_bf16(j, k)] = parts[p];
  }
}

template <int PL> struct WeightFragBf16 {
  bf16x8 v[PL][kW / 16];
};
template <int PL>
__device__ __forceinline__ void load_weight_fragment_bf16(const __bf16* __restrict__ F, int wv, int lane, WeightFragBf16<PL>& frag) {
#pragma unroll
  for (int p = 0; p < PL; ++p) {
    const bf16x8* f = (const bf16x8*)(F + p * kW * kW) + (wv * 16) * 64 + lane;
#pragma unroll
    for (int s = 0; s < kW / 16; ++s) frag.v[p][s] = f[s * 64];
  }
}

// C (32 x 32 of this wave) = X (32 x 256, LDS planes) * W (256 x 32, register planes): the products whose plane indices sum
// to <= PL - 1 (0-based), the leading one into `acc`, the others into `low`
template <int PL>
__device__ __forceinline__ f32x16 mfma_tile_bf16(const __bf16* __restrict__ Xs, const WeightFragBf16<PL>& w, int l31, int half) {
  f32x16 acc, low;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; low[r] = 0.f; }
#pragma unroll
  for (int s = 0; s < kW / 16; ++s) {
    bf16x8 a[PL];
#pragma unroll
    for (int p = 0; p < PL; ++p) a[p] = *(const bf16x8*)(Xs + p * kBopElems + (2 * s + half) * kBq + l31 * 8);
    // smallest terms first
#pragma unroll
    for (int order = PL - 1; order >= 1; --order)
#pragma unroll
      for (int i = 0; i <= order; ++i) low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], w.v[order - i][s], low, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w.v[0][s], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += low[r];
  return acc;
}

// fills the LDS planes of a tile: thread -> items (row = item & 31, k-group = item >> 5), item = t and t + 512; `value(row, c)`
template <int PL, class ValueOf>
__device__ __forceinline__ void fill_planes(__bf16* __restrict__ Xs, ValueOf value) {
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int item = (int)threadIdx.x + 512 * m, row = item & 31, kg = item >> 5;
    bf16x8 planes[PL];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 parts[PL];
      split_bf16<PL>(value(row, 8 * kg + e), parts);
#pragma unroll
      for (int p = 0; p < PL; ++p) planes[p][e] = parts[p];
    }
#pragma unroll
    for (int p = 0; p < PL; ++p) *(bf16x8*)(Xs + p * kBopElems + kg * kBq + row * 8) = planes[p];
  }
}

template <int PL>
__global__ __launch_bounds__(512) void icnn_fwd1_bf16(const float* __restrict__ x, long long ld, long long N, IcnnWeights<float> w,
                                                      const __bf16* __restrict__ Af, uint32_t* __restrict__ M1) {
  __shared__ float Qs[kMfmaRows][3];
  __shared__ __attribute__((aligned(16))) __bf16 Xs[PL * kBopElems];
  __shared__ f32x4 Wd0s[kW];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int col = 32 * wv + l31;
  WeightFragBf16<PL> frag;
  load_weight_fragment_bf16<PL>(Af, wv, lane, frag);
  const float wd1[3] = {w.Wd1[col], w.Wd1[kW + col], w.Wd1[2 * kW + col]};
  if (threadIdx.x < kW) Wd0s[threadIdx.x] = f32x4{w.Wd0[threadIdx.x], w.Wd0[kW + threadIdx.x], w.Wd0[2 * kW + threadIdx.x], 0.f};
  const long long tiles = (N + kMfmaRows - 1) / kMfmaRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kMfmaRows;
    __syncthreads();
    load_queries32(x, ld, w, n0, N, Qs);
    __syncthreads();
    fill_planes<PL>(Xs, [&](int row, int c) {
      const f32x4 d = Wd0s[c];
      return icnn_act(Qs[row][0] * d[0] + Qs[row][1] * d[1] + Qs[row][2] * d[2]);
    });
    __syncthreads();
    const f32x16 acc = mfma_tile_bf16<PL>(Xs, frag, l31, half);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mfma_row(reg, half);
      const float pre1 = acc[reg] + Qs[row][0] * wd1[0] + Qs[row][1] * wd1[1] + Qs[row][2] * wd1[2];
      const unsigned long long b = __ballot(pre1 > 0.f);
      if (lane == 0) {
        const int ra = mfma_row(reg, 0), rb = mfma_row(reg, 1);
        if (n0 + ra < N) M1[(n0 + ra) * kMaskWords + wv] = (uint32_t)(b & 0xffffffffull);
        if (n0 + rb < N) M1[(n0 + rb) * kMaskWords + wv] = (uint32_t)(b >> 32);
      }
    }
  }
}

// U1[row][c] = |wout[c]| * (mask bit ? 1 : 1/2): two values per column, so its bf16 planes are a per-column constant (the
// halving is exact) and nothing of U1 is stored -- the support-point product and the weight-gradient GEMM rebuild it from the
// mask words.
template <int PL> struct ColumnPlanes { __bf16 v[PL]; };

template <int PL>
__global__ __launch_bounds__(512) void icnn_fwd2_bf16(const float* __restrict__ x, long long ld, long long N, IcnnWeights<float> w,
                                                      const __bf16* __restrict__ ATf, const float* __restrict__ a,
                                                      const uint32_t* __restrict__ M1, float* __restrict__ U0, float* __restrict__ P) {
  __shared__ float Qs[kMfmaRows][3];
  __shared__ __attribute__((aligned(16))) __bf16 Xs[PL * kBopElems];
  __shared__ uint32_t Ms[kMfmaRows][kMaskWords];
  __shared__ float As[kW];
  __shared__ ColumnPlanes<PL> Ap[kW];
  __shared__ float Ys[kMfmaRows * kXs];  // U0 tile
  __shared__ f32x4 Wds[kW][2];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int col = 32 * wv + l31;
  WeightFragBf16<PL> frag;
  load_weight_fragment_bf16<PL>(ATf, wv, lane, frag);
  const float wd0[3] = {w.Wd0[col], w.Wd0[kW + col], w.Wd0[2 * kW + col]};
  if (threadIdx.x < kW) {
    const int c = threadIdx.x;
    Wds[c][0] = f32x4{w.Wd1[c], w.Wd1[kW + c], w.Wd1[2 * kW + c], 0.f};
    Wds[c][1] = f32x4{w.Wd0[c], w.Wd0[kW + c], w.Wd0[2 * kW + c], 0.f};
    As[c] = a[c];
    split_bf16<PL>(a[c], Ap[c].v);
  }
  const long long tiles = (N + kMfmaRows - 1) / kMfmaRows;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long n0 = tile * kMfmaRows;
    __syncthreads();
    load_queries32(x, ld, w, n0, N, Qs);
    if (threadIdx.x >= 256) {
      const int t = threadIdx.x - 256, row = t >> 3, word = t & 7;
      Ms[row][word] = (n0 + row < N) ? M1[(n0 + row) * kMaskWords + word] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m) {  // U1 tile planes: item -> (row, 8 columns)
      const int item = (int)threadIdx.x + 512 * m, row = item & 31, kg = item >> 5;
      const uint32_t bits = Ms[row][kg >> 2] >> (8 * (kg & 3));
      bf16x8 planes[PL];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool on = (bits >> e) & 1u;
#pragma unroll
        for (int p = 0; p < PL; ++p) {
          const __bf16 full = Ap[8 * kg + e].v[p];
          planes[p][e] = on ? full : (__bf16)((float)full * float(kIcnnSlope));  // (x 1/2: exact)
        }
      }
#pragma unroll
      for (int p = 0; p < PL; ++p) *(bf16x8*)(Xs + p * kBopElems + kg * kBq + row * 8) = planes[p];
    }
    __syncthreads();
    const f32x16 acc = mfma_tile_bf16<PL>(Xs, frag, l31, half);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mfma_row(reg, half);
      const float pre0 = Qs[row][0] * wd0[0] + Qs[row][1] * wd0[1] + Qs[row][2] * wd0[2];
      const float u0 = acc[reg] * icnn_mask(pre0);
      Ys[row * kXs + col] = u0;
      if (n0 + row < N) U0[(n0 + row) * kW + col] = u0;
    }
    __syncthreads();
    {  // P[row][i]: thread -> (row = t >> 4, part = t & 15), columns part + 16 m
      const int row = threadIdx.x >> 4, part = threadIdx.x & 15;
      float s[3] = {0.f, 0.f, 0.f};
#pragma unroll 2
      for (int m = 0; m < 16; ++m) {
        const int c = part + 16 * m;
        const float u1 = As[c] * mask_factor(Ms[row][c >> 5], c & 31), u0 = Ys[row * kXs + c];
        const f32x4 w1 = Wds[c][0], w0 = Wds[c][1];
#pragma unroll
        for (int i = 0; i < 3; ++i) s[i] += w1[i] * u1 + w0[i] * u0;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
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

// as icnn_bwd1_mfma; the Vb tile is split into planes as it is computed and written out row-major (N, 256) for icnn_bwd2_bf16
template <int PL>
__global__ __launch_bounds__(512) void icnn_bwd1_bf16(const float* __restrict__ x, long long ld, long long N, IcnnWeights<float> w,
                                                      const __bf16* __restrict__ Af, const float* __restrict__ a,
                                                      const uint32_t* __restrict__ M1, const float* __restrict__ U0,
                                                      const float* __restrict__ RB, double* __restrict__ partial, float* __restrict__ Vb) {
  __shared__ float Qs[kMfmaRows][3];
  __shared__ float Rs[kMfmaRows][3];
  __shared__ __attribute__((aligned(16))) __bf16 Xs[PL * kBopElems];
  __shared__ f32x4 Wd0s[kW];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int col = 32 * wv + l31;
  WeightFragBf16<PL> frag;
  load_weight_fragment_bf16<PL>(Af, wv, lane, frag);
  const float wd1[3] = {w.Wd1[col], w.Wd1[kW + col], w.Wd1[2 * kW + col]};
  const float acol = a[col];
  if (threadIdx.x < kW) Wd0s[threadIdx.x] = f32x4{w.Wd0[threadIdx.x], w.Wd0[kW + threadIdx.x], w.Wd0[2 * kW + threadIdx.x], 0.f};
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
#pragma unroll
    for (int m = 0; m < 2; ++m) {  // Vb tile: item -> (row = item >> 5, 8 columns of group item & 31): 1 KB rows written whole
      const int item = (int)threadIdx.x + 512 * m, row = item >> 5, kg = item & 31;
      bf16x8 planes[PL];
      f32x4 lo, hi;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f32x4 d = Wd0s[8 * kg + e];
        const float pre0 = Qs[row][0] * d[0] + Qs[row][1] * d[1] + Qs[row][2] * d[2];
        const float v = (Rs[row][0] * d[0] + Rs[row][1] * d[1] + Rs[row][2] * d[2]) * icnn_mask(pre0);  // rows past N: RB = 0
        if (e < 4) lo[e] = v; else hi[e - 4] = v;
        __bf16 parts[PL];
        split_bf16<PL>(v, parts);
#pragma unroll
        for (int p = 0; p < PL; ++p) planes[p][e] = parts[p];
      }
#pragma unroll
      for (int p = 0; p < PL; ++p) *(bf16x8*)(Xs + p * kBopElems + kg * kBq + row * 8) = planes[p];
      if (Vb && n0 + row < N) {
        f32x4* dst = (f32x4*)(Vb + (n0 + row) * kW + 8 * kg);
        dst[0] = lo;
        dst[1] = hi;
      }
    }
    float mfv[16], u0v[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mfma_row(reg, half);
      const bool ok = n0 + row < N;
      mfv[reg] = ok ? mask_factor(M1[(n0 + row) * kMaskWords + wv], l31) : 0.f;
      u0v[reg] = ok ? U0[(n0 + row) * kW + col] : 0.f;
    }
    __syncthreads();
    const f32x16 acc = mfma_tile_bf16<PL>(Xs, frag, l31, half);
    float t_abar = 0.f, t1[3] = {0.f, 0.f, 0.f}, t0[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = mfma_row(reg, half);
      const float u1b = acc[reg] + Rs[row][0] * wd1[0] + Rs[row][1] * wd1[1] + Rs[row][2] * wd1[2];
      t_abar += u1b * mfv[reg];
      const float u1 = acol * mfv[reg];
#pragma unroll
      for (int i = 0; i < 3; ++i) { t1[i] += Rs[row][i] * u1; t0[i] += Rs[row][i] * u0v[reg]; }
    }
    abar += double(t_abar);
#pragma unroll
    for (int i = 0; i < 3; ++i) { g1[i] += double(t1[i]); g0[i] += double(t0[i]); }
  }
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

// d|Wh| = Vb^T U1 as a split-K GEMM: grid (4 = k-half x j-half, n_slabs), a block owns a 128 x 128 piece over its slab of
// row tiles; wave v the k-tile v >> 1 and the two j-tiles 2 (v & 1), 2 (v & 1) + 1.  Vb comes row-major from icnn_bwd1_bf16
// (staged through LDS, double buffered); U1 is rebuilt from the mask words and |wout| -- per lane (one column) two constant
// sets of planes, selected by the mask bit.
template <int PL, bool F16 = false>
__global__ __launch_bounds__(512) void icnn_bwd2_bf16(long long N, const float* __restrict__ Vb, const uint32_t* __restrict__ M1,
                                                      const float* __restrict__ a, float* __restrict__ slabs,
                                                      const unsigned* __restrict__ rbmax) {
  __shared__ float Ls[2][kMfmaRows][128 + 1];
  __shared__ uint32_t Ms[2][kMfmaRows][4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int kq = blockIdx.x & 1, jq = blockIdx.x >> 1;
  const int ktl = wv >> 1, jtl = 2 * (wv & 1);
  const long long n_tiles = (N + kMfmaRows - 1) / kMfmaRows;
  const long long per = (n_tiles + gridDim.y - 1) / gridDim.y;
  const long long t_begin = (long long)blockIdx.y * per, t_end = (t_begin + per < n_tiles) ? t_begin + per : n_tiles;
  // this lane's columns of U1: j-tiles jtl and jtl + 1 of the piece
  // fp16 planes: Vb is an ADJOINT (its magnitude is the data's: loss scale, item weights) and the rows are this GEMM's contraction
  // index, so ONE power of two for the whole launch takes it into fp16's range: from the largest |r_bar| entry icnn_bwd1 met
  // (*rbmax, float bits; Vb = (r_bar . Wd0) . mask).  What matters for a sum over rows is the error relative to the largest rows.
  float vb_up = 1.f, vb_back = 1.f;
  if constexpr (F16) {
    // (256 words, one per workgroup of icnn_bwd1: lanes fold them -- bits of non-negative floats order like the numbers)
    unsigned top = 0u;
    if (rbmax) {
#pragma unroll
      for (int q = 0; q < 4; ++q) top = max(top, rbmax[64 * q + lane]);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) top = max(top, (unsigned)__shfl_xor((int)top, off));
    }
    const unsigned e = rbmax ? (top >> 23) & 0xffu : 127u;
    vb_up = __builtin_bit_cast(float, (254u - e) << 23);
    vb_back = __builtin_bit_cast(float, e << 23);  // (e = 0: no row had a gradient -- every product is zero)
  }
  __bf16 ua[2][PL];
  split_planes<PL, F16>(a[128 * jq + 32 * jtl + l31], ua[0]);
  split_planes<PL, F16>(a[128 * jq + 32 * (jtl + 1) + l31], ua[1]);
  f32x16 acc0, acc1, low0, low1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; low0[r] = 0.f; low1[r] = 0.f; }
  // copy assignment: float4 number t + 512 m of the tile's 32 x 128 floats (row = number >> 5, 4 columns 4 (number & 31))
  f32x4 stage[2];
  uint32_t mstage = 0u;
  auto fetch = [&](long long t) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int number = (int)threadIdx.x + 512 * m, row = number >> 5, c4 = number & 31;
      const long long n = t * kMfmaRows + row;
      stage[m] = n < N ? *(const f32x4*)(Vb + n * kW + 128 * kq + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (threadIdx.x < kMfmaRows * 4) {
      const long long n = t * kMfmaRows + (threadIdx.x >> 2);
      mstage = n < N ? M1[n * kMaskWords + 4 * jq + (threadIdx.x & 3)] : 0u;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int number = (int)threadIdx.x + 512 * m, row = number >> 5, c4 = number & 31;
#pragma unroll
      for (int e = 0; e < 4; ++e) Ls[buf][row][4 * c4 + e] = stage[m][e];
    }
    if (threadIdx.x < kMfmaRows * 4) Ms[buf][threadIdx.x >> 2][threadIdx.x & 3] = mstage;
  };
  if (t_begin < t_end) { fetch(t_begin); commit(0); }
  __syncthreads();
  for (long long t = t_begin; t < t_end; ++t) {
    const int cur = (int)((t - t_begin) & 1);
    const bool more = t + 1 < t_end;
    if (more) fetch(t + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 av[PL], b0[PL], b1[PL];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = 16 * s + 8 * half + e;
        __bf16 parts[PL];
        split_planes<PL, F16>(Ls[cur][row][32 * ktl + l31] * vb_up, parts);
        const bool on0 = (Ms[cur][row][jtl] >> l31) & 1u, on1 = (Ms[cur][row][jtl + 1] >> l31) & 1u;
#pragma unroll
        for (int p = 0; p < PL; ++p) {
          av[p][e] = parts[p];
          b0[p][e] = on0 ? ua[0][p] : enc16<F16>(dec16<F16>(ua[0][p]) * float(kIcnnSlope));
          b1[p][e] = on1 ? ua[1][p] : enc16<F16>(dec16<F16>(ua[1][p]) * float(kIcnnSlope));
        }
      }
#pragma unroll
      for (int order = PL - 1; order >= 1; --order)
#pragma unroll
        for (int i = 0; i <= order; ++i) {
          low0 = mfma16<F16>(av[i], b0[order - i], low0);
          low1 = mfma16<F16>(av[i], b1[order - i], low1);
        }
      acc0 = mfma16<F16>(av[0], b0[0], acc0);
      acc1 = mfma16<F16>(av[0], b1[0], acc1);
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
  }
  float* slab = slabs + (long long)blockIdx.y * kW * kW;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int k = 128 * kq + 32 * ktl + mfma_row(reg, half);
    slab[k * kW + 128 * jq + 32 * jtl + l31] = (acc0[reg] + low0[reg] * kLowBack<F16>) * vb_back;
    slab[k * kW + 128 * jq + 32 * (jtl + 1) + l31] = (acc1[reg] + low1[reg] * kLowBack<F16>) * vb_back;
  }
}

}  // namespace
