// dpll_icnn_pipe.hip -- the ICNN GEMMs of the mesh-geometry path (DeepSupportConvex / HomogeneousICNN, geometry.py:309-325,
// deep_support_function.py:213-266) as ONE-WAVE-PER-SIMD, SOFTWARE-PIPELINED kernels for gfx950 (round 5).
//
// Why a second form: the 8-wave kernels of dpll_mesh_kernels.hpp run two waves per SIMD and separate the
// phases of a 32-row tile (queries -> fill -> 128 MFMAs -> epilogue) by workgroup barriers, so a tile costs the SUM of its
// phases on the slower wave: 0.41 of the f32 matrix rate for four rounds.  Here a workgroup is FOUR waves, one per SIMD of
// its CU, with the whole 512-register file each:
//   * wave v owns output columns [64 v, 64 v + 64) of every tile: its 256 x 64 block of the weight matrix stays in 256
//     registers for the whole launch (the compiler places what does not fit the 256 architectural VGPRs in AGPRs and feeds
//     the MFMA from there: B operands may be AGPRs on gfx950);
//   * a v_mfma_f32_32x32x2_f32 occupies the matrix pipe for 64 cycles and the wave's issue port for a few, so the SAME wave
//     fills the NEXT tile's operand image, runs the PREVIOUS tile's epilogue and fetches the rows of the tile after next in
//     the shadow of this tile's 256 MFMAs.  The two 32-column blocks of a wave are TWO CHAINS run one after the other (a
//     dependent f32 MFMA chain issues at the pipe's full rate): while chain 0 of tile t runs, the epilogue of chain 1 of
//     tile t - 1 reads that chain's accumulator (which chain 1 of tile t only claims afterwards), and while chain 1 runs,
//     the epilogue of chain 0 of the SAME tile -- no accumulator is ever copied and 32 registers hold both.  A step = one
//     16-byte LDS read + 4 MFMAs + either one row of the next tile's fill or one accumulator register of an epilogue;
//   * ONE workgroup barrier per tile (the hand-over of the double-buffered operand image), no control flow inside a tile:
//     rows beyond N and tiles beyond the last are handled by clamped addresses, zeroed inputs and a spare "dump" tile.
// Operand maps of v_mfma_f32_32x32x2_f32:  A: lane l holds A[i = l & 31][k = l >> 5];  B: lane l holds B[k = l >> 5][j = l & 31];
// C/D: 16 registers, col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "dpll_icnn_pipe_api.hpp"

namespace {

using namespace dpll;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

constexpr int kW = kIcnnWidth;       // 256
constexpr int kRows = dpll_pipe::kTileRows;
constexpr int kMaskWords = kW / 32;  // 8
constexpr int kB1Cols = 7 * kW;      // partial row of bwd1: [d|wout| | dWd1 (3 W) | dWd0 (3 W)]
// LDS image of a 32 x 256 A-operand tile (the layout of dpll_mesh_kernels.hpp): the four values lane (row, half) feeds to
// steps 4 kq .. 4 kq + 3 (k = 2 (4 kq + e) + half) are one aligned float4; the 8-float pad per kq keeps the column-wise
// fills on distinct banks
constexpr int kXq = 32 * 8 + 8;
constexpr int kXopFloats = 32 * kXq;
__device__ __forceinline__ constexpr int xop(int row, int k) { return (k >> 3) * kXq + row * 8 + (k & 1) * 4 + ((k >> 1) & 3); }
__device__ __forceinline__ constexpr int mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
// fill order: step i of 32 handles row fill_row(i); four consecutive steps are the rows of ONE float4 of the operand-tile
// layout bwd2 reads (rows 2 (4 q + e) + h, e = 0..3)
__device__ __forceinline__ constexpr int fill_row(int i) { return 8 * (i >> 3) + 2 * (i & 3) + ((i >> 2) & 1); }

// per-row data of a tile in LDS: [0] = query (q0, q1, q2, -), [1] = r_bar (r0, r1, r2, -), [2..3] = the row's 8 mask words
struct RowRing { f32x4 v[4][kRows][4]; };

enum : int { kFwd1 = 0, kFwd2 = 1, kBwd1 = 2 };
#ifndef DPLL_PIPE_SCHED
#define DPLL_PIPE_SCHED 3
#endif
#ifndef DPLL_PIPE_VALU
#define DPLL_PIPE_VALU 4
#endif
constexpr int kSideValu = DPLL_PIPE_VALU;  // VALU instructions scheduled behind each MFMA of a step

struct PipeArgs {
  const float* x; long long ld; long long N;
  IcnnWeights<float> w;
  const float* F;          // |Wh| (fwd1, bwd1) or |Wh|^T (fwd2) in fragment order (frag_index of dpll_mesh_kernels.hpp)
  const float* a;          // |wout|
  const uint32_t* M1in;    // fwd2, bwd1
  uint32_t* M1out;         // fwd1
  const float* U0in;       // bwd1 (accumulator layout)
  float* U0out;            // fwd2
  unsigned* rbmax;         // bwd1, fp16 planes: kMaxBlocks words, workgroup b leaves the largest |r_bar| entry it met (float bits) in [b]
  const float* RB;         // bwd1
  float* P;                // fwd2
  double* partial;         // bwd1
  float* VbT;              // bwd1 (operand tiles for bwd2; one spare tile behind the last)
};

__device__ __forceinline__ float mask_factor(uint32_t word, int bit) { return ((word >> bit) & 1u) ? 1.0f : float(kIcnnSlope); }

// Four k-steps of a chain as ONE statement: v_mfma_f32_32x32x2_f32 runs on the vector ALU's own multipliers (its rate IS the f32
// vector rate; tools/diag/mfma_fill.hip: every VALU instruction between two of them costs its full issue time plus ~10 cycles
// for the switch), so the MFMAs of a step are kept back to back and the weights are read straight from accumulation registers
// ("a": no v_accvgpr_read in front of the MFMA).  FIRST: the chain's first step starts from C = 0; LAST: the wait states an
// XDL result needs before a VALU instruction may read it (hipcc pads nothing for an asm statement).
template <bool FIRST, bool LAST>
__device__ __forceinline__ void mfma_x4(f32x16& acc, const f32x4& x, float w0, float w1, float w2, float w3) {
  if (FIRST) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %5, 0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0"
                 : "=&v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "a"(w0), "a"(w1), "a"(w2), "a"(w3));
  } else if (LAST) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %5, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0\n\t"
                 "s_nop 15\n\ts_nop 3"
                 : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "a"(w0), "a"(w1), "a"(w2), "a"(w3));
  } else {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %5, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0"
                 : "+v"(acc) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "a"(w0), "a"(w1), "a"(w2), "a"(w3));
  }
}

// ---- the split-bf16 form (BF16 = true: what dpll_solver_opts_t.mesh_gemm = 2 runs).  An f32 operand is the sum of two bf16 planes
// to 16 significand bits, x = x1 + x2; per 16-deep k-step three products on the bf16 matrix cores -- x1 w2 and x2 w1 into a
// second accumulator `low`, x1 w1 into `acc`, added once per chain -- as dpll_mesh_bf16.hpp (same planes, same order, so the
// same numbers).  Unlike the f32 MFMA, v_mfma_f32_32x32x16_bf16 runs on the matrix pipe proper and up to five other
// instructions hide behind each (tools/diag/mfma_fill.hip): here the pipelined structure pays, and the MFMAs are issued ONE per
// statement with a share of the step's side work behind each.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
constexpr int kBq = 32 * 8 + 8;       // bf16 elements per 8-deep k-block of the LDS image (8 pad)
constexpr int kBopElems = 32 * kBq;   // one plane of a 32 x 256 tile: element (row, k) at (k >> 3) * kBq + row * 8 + (k & 7)
static_assert(2 * kBopElems * 2 == kXopFloats * 4, "two bf16 planes fill exactly the f32 image's bytes");

// acc (+)= a * w on the bf16 matrix cores, the weight vector read from accumulation registers.  FIRST: C = 0.  (hipcc pads
// nothing inside an asm statement; a dependent MFMA taking the previous result whole as C needs no wait state.)
// F16: the two planes are fp16 numbers in the same 16-bit containers (dpll_mesh_bf16.hpp: mesh_gemm = 4), the instruction is the
// fp16 one of the same shape and rate
template <bool FIRST, bool F16 = false>
__device__ __forceinline__ void mfma_bf16(f32x16& acc, const bf16x8& a, const bf16x8& w) {
  if constexpr (F16) {
    if (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(w));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w));
  } else {
    if (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(w));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w));
  }
}
// the two 16-bit planes of a value: bf16 (hi, residual) or fp16 (hi, residual x 2^11), as 16-bit patterns
constexpr float kF16LowScale = 2048.f;
template <bool F16> __device__ __forceinline__ void split2(float v, unsigned short& hi, unsigned short& lo) {
  if constexpr (F16) {
    const _Float16 h = (_Float16)v;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (_Float16)((v - (float)h) * kF16LowScale));
  } else {
    const __bf16 h = (__bf16)v;  // round to nearest even; the residual is exact in f32
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
  }
}
// fp16 planes, adjoint operands: a power of two that takes a row of magnitude m (its largest |entry|) to [1, 2), and its inverse.
// m = 0 (a row past N, a contact without gradient): the scale is 2^127 and the inverse 0 -- the row's products are zeros either way.
__device__ __forceinline__ float row_max3(const f32x4& r) { return fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fabsf(r[2])); }
__device__ __forceinline__ float pow2_up(float m) { return __builtin_bit_cast(float, (254u - ((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu)) << 23); }
__device__ __forceinline__ float pow2_back(float m) { return __builtin_bit_cast(float, ((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) << 23); }
template <bool F16> __device__ __forceinline__ float plane_value(unsigned short bits) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, bits);
  else return (float)__builtin_bit_cast(__bf16, bits);
}
template <bool F16> __device__ __forceinline__ unsigned short plane_bits(float v) {  // (v is exactly representable: a plane value halved)
  if constexpr (F16) return __builtin_bit_cast(unsigned short, (_Float16)v);
  else return __builtin_bit_cast(unsigned short, (__bf16)v);
}
// the wait states an 8-pass XDL result needs before a VALU instruction may read it
__device__ __forceinline__ void mfma_settle(f32x16& a, f32x16& b) { asm volatile("s_nop 7\n\ts_nop 4" : "+v"(a), "+v"(b)); }

// DPP: lane 15 of rows 0 / 2 into every lane of rows 1 / 3 (gfx9 row_bcast:15, row_mask 0xA); other rows read 0
__device__ __forceinline__ float row_bcast15(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ float dppf(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

#ifdef DPLL_PIPE_STAMPS
// diagnostic build (tools/diag/pipe_bench.hip): shader-clock stamps of wave 0 of every workgroup
__device__ unsigned long long g_pipe_stamps[256][16];
#define PIPE_STAMP(i) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_pipe_stamps[blockIdx.x][i] = t_; } } while (0)
#else
#define PIPE_STAMP(i) do { } while (0)
#endif

template <int KIND, bool BF16 = false, bool F16 = false>
__global__ __launch_bounds__(256) void icnn_pipe_kernel(PipeArgs g) {
  __shared__ __attribute__((aligned(16))) float Xs[2][kXopFloats];
  __shared__ RowRing ring;
  // fwd2 only: per-wave partial support points of a tile (double buffered: written by the epilogue of one iteration, summed
  // over the four waves in the next)
  __shared__ float Pp[KIND == kFwd2 ? 4 : 1][8][kRows][4];  // [tile & 3][wave, column block][row]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  PIPE_STAMP(0);
  const long long N = g.N;
  const long long n_tiles = (N + kRows - 1) / kRows;
  const long long my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;  // tiles blockIdx.x + it gridDim.x
  auto tile_of = [&](long long it) { return (long long)blockIdx.x + it * gridDim.x; };

  // ---- this wave's 256 x 64 block of the weight matrix: 2 column blocks x 128 k-steps (requested in the prologue, BEHIND
  // the row data of the first two tiles: memory returns in order, so waiting for the rows must not mean waiting for 256 KB)
  float b[2][kW / 2];
  bf16x8 wb[2][2][kW / 16];  // BF16: [plane][column block][k-step of 16]: 2 x 2 x 16 vectors of 4 registers
  // ---- per-thread constants --------------------------------------------------------------------------------------
  const int c = tid;  // fill role: column c of the operand tile
  float d[3] = {0.f, 0.f, 0.f}, ac = 0.f;
  if (!BF16 && (KIND == kFwd1 || KIND == kBwd1)) { d[0] = g.w.Wd0[c]; d[1] = g.w.Wd0[kW + c]; d[2] = g.w.Wd0[2 * kW + c]; }
  if (!BF16 && KIND == kFwd2) ac = g.a[c];
  // BF16 fill role: the column PAIR (c0, c0 + 1) -- one 32-bit LDS word per row and plane -- and the rows of parity rpar
  const int c0 = 2 * (tid & 127), rpar = tid >> 7;
  float d2[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, ahi[2] = {0.f, 0.f}, alo[2] = {0.f, 0.f};
  if (BF16) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (KIND == kFwd1 || KIND == kBwd1) {
#pragma unroll
        for (int i = 0; i < 3; ++i) d2[j][i] = g.w.Wd0[i * kW + c0 + j];
      }
      if (KIND == kFwd2) {  // U1 = |wout[c]| (mask ? 1 : 1/2): its planes are per-column constants (the halving is exact)
        const float av = g.a[c0 + j];
        unsigned short hb, lb;
        split2<F16>(av, hb, lb);
        ahi[j] = plane_value<F16>(hb);
        alo[j] = plane_value<F16>(lb);  // (F16: already scaled by 2^11)
      }
    }
  }
  int col[2];
  float wd0[2][3], wd1[2][3], acol[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    col[cb] = 64 * wv + 32 * cb + l31;
#pragma unroll
    for (int i = 0; i < 3; ++i) { wd0[cb][i] = g.w.Wd0[i * kW + col[cb]]; wd1[cb][i] = g.w.Wd1[i * kW + col[cb]]; }
    acol[cb] = (KIND == kFwd1) ? 0.f : g.a[col[cb]];
  }
  float pert3[3] = {0.f, 0.f, 0.f};
  if (!g.w.dirs && tid < kRows) {
#pragma unroll
    for (int i = 0; i < 3; ++i) pert3[i] = g.w.pert[3 * (tid & 3) + i];
  }

  // ---- row data of a tile: global -> registers (issue) -> LDS ring (commit) -----------------------------------------
  float rb_seen = 0.f;  // bwd1, fp16 planes: the largest |r_bar| entry of the rows this lane loaded
  struct RowRaw { float q[4]; float r3[3]; uint32_t m; bool ok; };
  auto rows_issue = [&](long long tile) {
    RowRaw raw;
    const long long n0 = tile * kRows;
    raw.ok = false; raw.m = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) raw.q[i] = 0.f;
    if (tid < kRows) {  // the row's query direction (raw: explicit direction or the item's quaternion)
      const long long n = n0 + tid;
      raw.ok = n < N;
      const long long nc = raw.ok ? n : N - 1;
      // (one path for both sources, no branch: a branch here made the wave wait for memory before it had asked for its weights)
      const float* src = g.w.dirs ? g.w.dirs + 3 * nc : g.x + (nc >> 2) * g.ld;
#pragma unroll
      for (int i = 0; i < 4; ++i) raw.q[i] = src[(i == 3 && g.w.dirs) ? 2 : i];
    }
    raw.r3[0] = raw.r3[1] = raw.r3[2] = 0.f;
    if (KIND == kBwd1 && tid >= 64 && tid < 64 + kRows) {  // lane r of wave 1: the three components of row r's adjoint
      const int r = tid - 64;
      const long long n = n0 + r;
      const uint32_t nc = (uint32_t)(n < N ? n : N - 1);  // (N < 2^31: the launchers check; a 64-bit division here kept waves 1-2 of
      const uint32_t item = nc / (uint32_t)g.w.qpi;         //  bwd1 ~3 k cycles behind at the prologue's first barrier)
      const int jq = (int)(nc - item * (uint32_t)g.w.qpi);
      int off = g.w.qoff[0];
#pragma unroll
      for (int cq = 1; cq < 8; ++cq) off = (jq == cq) ? g.w.qoff[cq] : off;
      const float* src = g.RB + (long long)item * g.w.point_stride + off;
#pragma unroll
      for (int i = 0; i < 3; ++i) { const float v = src[i]; raw.r3[i] = n < N ? v : 0.f; }
    }
    if (KIND != kFwd1) {
      const long long n = n0 + (tid >> 3);
      const long long nc = n < N ? n : N - 1;
      const uint32_t v = g.M1in[nc * kMaskWords + (tid & 7)];
      raw.m = n < N ? v : 0u;
    }
    return raw;
  };
  auto rows_commit = [&](const RowRaw& raw, int slot) {
    if (tid < kRows) {
      float q[3] = {0.f, 0.f, 1.f};
      if (raw.ok) {
        if (g.w.dirs) {
#pragma unroll
          for (int i = 0; i < 3; ++i) q[i] = raw.q[i];
        } else {
          icnn_query<float>(raw.q, pert3, q);
        }
      }
      ring.v[slot][tid][0] = f32x4{q[0], q[1], q[2], 0.f};
    }
    if (KIND == kBwd1 && tid >= 64 && tid < 64 + kRows) {
      // (fp16 planes: component 3 = the power of two that takes the row's results back from the scaled planes, pow2_back of the
      // row's largest |entry|; the fill forms its inverse from the same bits)
      const f32x4 rb = f32x4{raw.r3[0], raw.r3[1], raw.r3[2], 0.f};
      const float m = row_max3(rb);
      ring.v[slot][tid - 64][1] = f32x4{raw.r3[0], raw.r3[1], raw.r3[2], F16 ? pow2_back(m) : 0.f};
      if constexpr (F16) rb_seen = fmaxf(rb_seen, m);
    }
    if (KIND != kFwd1) ((uint32_t*)&ring.v[slot][tid >> 3][2])[tid & 7] = raw.m;
  };

  // ---- side work of the MFMA steps comes in two halves: the LDS reads of a step's inputs are issued ONE STEP AHEAD (side_in),
  // so that their latency passes under four MFMAs instead of stalling the wave between them -----------------------------
  struct SideIn { f32x4 a, b; uint32_t w; };

  // ---- fill: step i = row fill_row(i) of column c of a tile's operand image ----------------------------------------
  auto fill_load = [&](int i, int slot) {
    const int rr = fill_row(i);
    SideIn in;
    in.a = f32x4{0.f, 0.f, 0.f, 0.f}; in.b = in.a; in.w = 0u;
    if (KIND == kFwd1) in.a = ring.v[slot][rr][0];
    if (KIND == kFwd2) in.w = ((const uint32_t*)&ring.v[slot][rr][2])[c >> 5];
    if (KIND == kBwd1) { in.a = ring.v[slot][rr][0]; in.b = ring.v[slot][rr][1]; }
    return in;
  };
  f32x4 vb4 = {0.f, 0.f, 0.f, 0.f};  // bwd1: the four rows of one float4 of the operand tile for bwd2
  auto fill_step = [&](int i, const SideIn& in, float* __restrict__ X, float* __restrict__ vb_tile) {
    const int rr = fill_row(i);
    float val;
    if (KIND == kFwd1) {
      {
        const float pre = in.a[0] * d[0] + in.a[1] * d[1] + in.a[2] * d[2];
        val = fmaxf(pre, float(kIcnnSlope) * pre);  // = icnn_act for every finite pre (slope < 1), one instruction fewer
      }
    } else if (KIND == kFwd2) {
      val = ac * mask_factor(in.w, c & 31);
    } else {
      const float pre0 = in.a[0] * d[0] + in.a[1] * d[1] + in.a[2] * d[2];
      val = (in.b[0] * d[0] + in.b[1] * d[1] + in.b[2] * d[2]) * icnn_mask(pre0);  // rows past N: r_bar = 0
      vb4[i & 3] = val;
      if ((i & 3) == 3)  // rows 8 q + 2 e + h, e = 0..3 -> float4 ((q * 32 + (c & 31)) * 2 + h) of block (tile, c >> 5)
        ((f32x4*)(vb_tile + (c >> 5) * 1024))[((i >> 3) * 32 + (c & 31)) * 2 + ((i >> 2) & 1)] = vb4;
    }
    X[xop(rr, c)] = val;
  };

  // BF16 fill: step i of 16 = row 2 i + rpar of the column pair; Xb = the buffer's two planes; vb_rows = Vb row-major (n, 256)
  auto fill_load_b = [&](int i, int slot) {
    const int rr = 2 * i + rpar;
    SideIn in;
    in.a = f32x4{0.f, 0.f, 0.f, 0.f}; in.b = in.a; in.w = 0u;
    if (KIND == kFwd1) in.a = ring.v[slot][rr][0];
    if (KIND == kFwd2) in.w = ((const uint32_t*)&ring.v[slot][rr][2])[c0 >> 5];
    if (KIND == kBwd1) { in.a = ring.v[slot][rr][0]; in.b = ring.v[slot][rr][1]; }
    return in;
  };
  auto fill_step_b = [&](int i, const SideIn& in, float* __restrict__ X, float* __restrict__ vb_rows) {
    const int rr = 2 * i + rpar;
    unsigned short h0, h1, l0, l1;
    if (KIND == kFwd2) {
      const float f0 = mask_factor(in.w, c0 & 31), f1 = mask_factor(in.w, (c0 & 31) + 1);
      h0 = plane_bits<F16>(ahi[0] * f0); h1 = plane_bits<F16>(ahi[1] * f1);
      l0 = plane_bits<F16>(alo[0] * f0); l1 = plane_bits<F16>(alo[1] * f1);
    } else {
      float val[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float pre = in.a[0] * d2[j][0] + in.a[1] * d2[j][1] + in.a[2] * d2[j][2];
        if (KIND == kFwd1) val[j] = fmaxf(pre, float(kIcnnSlope) * pre);
        else val[j] = (in.b[0] * d2[j][0] + in.b[1] * d2[j][1] + in.b[2] * d2[j][2]) * icnn_mask(pre);  // rows past N: r_bar = 0
      }
      if (KIND == kBwd1) *(float2*)(vb_rows + rr * kW + c0) = float2{val[0], val[1]};
      // fp16 planes of an ADJOINT row: its magnitude is the data's (the loss scale 1 / batch, the item's weight), so the row goes
      // to the planes scaled to [1, 2) |Wd0| by a power of two taken from r_bar itself; the epilogue scales the row's results back
      const float up = (F16 && KIND == kBwd1) ? pow2_up(row_max3(in.b)) : 1.f;
      split2<F16>(val[0] * up, h0, l0);
      split2<F16>(val[1] * up, h1, l1);
    }
    uint32_t* plane0 = (uint32_t*)X;  // (two 16-bit values per word)
    const int word = ((c0 >> 3) * kBq + rr * 8 + (c0 & 7)) >> 1;
    plane0[word] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    plane0[kBopElems / 2 + word] = (uint32_t)l0 | ((uint32_t)l1 << 16);
  };

  // ---- epilogue state ------------------------------------------------------------------------------------------------
  // The two 32-column blocks of a wave are two MFMA chains run one after the other; acc[1] enters tile t still holding chain 1
  // of tile t - 1 (zero before the first tile), whose epilogue runs under chain 0 of tile t
  f32x16 low[2];  // BF16: the small products of a chain, added to acc once at its end
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) low[cb][r] = 0.f;
  f32x16 acc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  uint32_t mw = 0u;                        // fwd1: mask words of the chain's block, the word of row r in lane r
  f32x4 u0s = {0.f, 0.f, 0.f, 0.f};        // fwd2: U0 values staged for 16-byte stores
  float pv[KIND == kFwd2 ? 16 : 1][3];     // fwd2: the chain's support-point terms of this lane's column, one row per accumulator register
  // bwd1: U0 of the chain in accumulator layout (one float4 in use, the next in flight); column sums of the chain in float,
  // the running totals over this workgroup's tiles in double in LDS (14 per thread: registers are the scarce resource)
  f32x4 u0cur = {0.f, 0.f, 0.f, 0.f}, u0nxt = {0.f, 0.f, 0.f, 0.f};
  const f32x4* u0src = nullptr;
  float t_abar = 0.f, t_s1[3] = {0.f, 0.f, 0.f}, t_g0[3] = {0.f, 0.f, 0.f};
  __shared__ double totals[KIND == kBwd1 ? 14 : 1][256];
  if (KIND == kBwd1) {
#pragma unroll
    for (int j = 0; j < 14; ++j) totals[j][tid] = 0.0;
  }

  // epilogue of chain `cb` of a tile: begin (its U0 stream), 16 steps (one accumulator register each), end
  auto epi_begin = [&](int cb, long long tile_c /* clamped to a readable tile */) {
    if (KIND == kBwd1) {
      u0src = (const f32x4*)(g.U0in + (tile_c * 4 + wv) * 2048) + (4 * cb) * 64 + lane;
      u0nxt = u0src[0];
      t_abar = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) { t_s1[i] = 0.f; t_g0[i] = 0.f; }
    }
  };
  auto epi_load = [&](int cb, int reg, int slot) {
    const int row = mfma_row(reg, half);  // (half is a lane property: two rows per register)
    SideIn in;
    in.b = f32x4{0.f, 0.f, 0.f, 0.f};
    in.a = ring.v[slot][row][KIND == kBwd1 ? 1 : 0];  // r_bar (bwd1) or the query
    in.w = KIND == kFwd1 ? 0u : ((const uint32_t*)&ring.v[slot][row][2])[2 * wv + cb];
    return in;
  };
  auto epi_step = [&](int cb, int reg, const SideIn& in, float* __restrict__ u0_tile, int pbuf) {
    const int row = mfma_row(reg, half);
    const float av = acc[cb][reg];
    if (KIND == kFwd1) {
      const f32x4 q = in.a;
      const float pre1 = av + q[0] * wd1[cb][0] + q[1] * wd1[cb][1] + q[2] * wd1[cb][2];
      const unsigned long long bal = __ballot(pre1 > 0.f);
      // low word: rows of half 0, high word: rows of half 1; the word of row r goes to lane r
      mw = lane == mfma_row(reg, 0) ? (uint32_t)(bal & 0xffffffffull) : mw;
      mw = lane == mfma_row(reg, 1) ? (uint32_t)(bal >> 32) : mw;
    } else if (KIND == kFwd2) {
      const f32x4 q = in.a;
      const uint32_t word = in.w;
      const float pre0 = q[0] * wd0[cb][0] + q[1] * wd0[cb][1] + q[2] * wd0[cb][2];
      const float u0 = av * icnn_mask(pre0);
      const float u1 = acol[cb] * mask_factor(word, l31);
      u0s[reg & 3] = u0;
      if ((reg & 3) == 3) ((f32x4*)u0_tile)[(4 * cb + (reg >> 2)) * 64] = u0s;  // accumulator layout: float4 (4 cb + reg / 4) of this lane
      // this row's support-point terms of the lane's column; summed over the block's 32 columns at the end of the chain (epi_end)
#pragma unroll
      for (int i = 0; i < 3; ++i) pv[KIND == kFwd2 ? reg : 0][i] = wd1[cb][i] * u1 + wd0[cb][i] * u0;
      (void)pbuf;
    } else {
      const f32x4 r = in.a;
      const float mf = mask_factor(in.w, l31);
      const float avs = F16 ? av * r[3] : av;  // (fp16 planes: the row went through the GEMM scaled by 1 / r[3], rows_commit)
      const float u1b = avs + r[0] * wd1[cb][0] + r[1] * wd1[cb][1] + r[2] * wd1[cb][2];
      t_abar += u1b * mf;  // rows past N / tiles that do not exist: r_bar = 0 and a zero accumulator
      if ((reg & 3) == 0) {
        u0cur = u0nxt;
        if (reg + 4 < 16) u0nxt = u0src[((reg >> 2) + 1) * 64];
      }
      const float u0 = u0cur[reg & 3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { t_s1[i] += r[i] * mf; t_g0[i] += r[i] * u0; }
    }
  };
  auto epi_end = [&](int cb, long long tile, int pbuf = 0) {
    const long long n0 = tile * kRows;
    if constexpr (KIND == kFwd2) {
      // Sums over the 32 lanes of a half of 16 x 3 values per lane, as a TRANSPOSED butterfly: at every stage a lane keeps half of
      // its values (which half: one bit of its lane number) and adds its partner's copies of those -- 45 operations per component
      // instead of 80, and 15 of them cross-lane instead of 80 (round 5: the 15 DPP adds per accumulator register of the straight
      // row sums were ~4.8 k of fwd2's ~10.5 k non-MFMA cycles per tile).  Pairings: row_mirror (i <-> 15 - i, side = bit 3),
      // row_half_mirror (i <-> 7 - i, bit 2), quad xor 2 (bit 1), quad xor 1 (bit 0): partners always hold the same set, and lane
      // j of a 16-lane row ends with the row's sum of register j; the two rows of a half are added through ds_swizzle (i <-> i ^ 16).
      const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
      float tot[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float s8[8], s4[4], s2[2];
#pragma unroll
        for (int r = 0; r < 8; ++r) s8[r] = (b3 ? pv[r + 8][i] : pv[r][i]) + dppf<0x140>(b3 ? pv[r][i] : pv[r + 8][i]);
#pragma unroll
        for (int r = 0; r < 4; ++r) s4[r] = (b2 ? s8[r + 4] : s8[r]) + dppf<0x141>(b2 ? s8[r] : s8[r + 4]);
#pragma unroll
        for (int r = 0; r < 2; ++r) s2[r] = (b1 ? s4[r + 2] : s4[r]) + dppf<0x4E>(b1 ? s4[r] : s4[r + 2]);
        const float s1 = (b0 ? s2[1] : s2[0]) + dppf<0xB1>(b0 ? s2[0] : s2[1]);
        tot[i] = s1 + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, s1), 0x401F));  // + lane ^ 16
      }
      if ((lane & 16) == 0) {  // lanes 0..15 of each half: register (lane & 15) = row mfma_row(lane & 15, half)
        const int reg = lane & 15;
        f32x4* dst = (f32x4*)&Pp[pbuf][2 * wv + cb][(reg & 3) + 8 * (reg >> 2) + 4 * half][0];
        *dst = f32x4{tot[0], tot[1], tot[2], 0.f};
      }
    }
    if (KIND == kFwd1) {
      if (lane < kRows && n0 + lane < N) g.M1out[(n0 + lane) * kMaskWords + 2 * wv + cb] = mw;
    } else if (KIND == kBwd1) {
      totals[7 * cb][tid] += double(t_abar);
#pragma unroll
      for (int i = 0; i < 3; ++i) { totals[7 * cb + 1 + i][tid] += double(t_s1[i]); totals[7 * cb + 4 + i][tid] += double(t_g0[i]); }
    }
  };
  // fwd2: support points of a tile whose eight partials (wave, block) were written in EARLIER iterations (a barrier lies between)
  auto p_store = [&](long long it_of_tile) {
    if (KIND != kFwd2) return;
    const long long tile = tile_of(it_of_tile);
    const int pbuf = (int)(it_of_tile & 3);  // (by the workgroup's own count of tiles: tile numbers advance by the grid size)
    if (tid < 3 * kRows) {
      const int r = tid / 3, i = tid - 3 * r;
      const long long n = tile * kRows + r;
      if (n < N)
        g.P[icnn_point_index(n, g.w) + i] = ((Pp[pbuf][0][r][i] + Pp[pbuf][1][r][i]) + (Pp[pbuf][2][r][i] + Pp[pbuf][3][r][i])) +
                                            ((Pp[pbuf][4][r][i] + Pp[pbuf][5][r][i]) + (Pp[pbuf][6][r][i] + Pp[pbuf][7][r][i]));
    }
  };

  // ---- prologue: ring zeroed (slot 3 serves the epilogue of the tile "before the first"), rows of the first two tiles,
  // the first operand image ------------------------------------------------------------------------------------------
  {
    f32x4* z = &ring.v[0][0][0];
    for (int i = tid; i < 4 * kRows * 4; i += 256) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  {
    const RowRaw r0 = rows_issue(tile_of(0));
    const RowRaw r1 = rows_issue(tile_of(1));
    if (BF16) {  // chain 0's planes (fragment order of dpll_mesh_bf16.hpp: plane p at element offset p * 65536)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const bf16x8* f = (const bf16x8*)((const __bf16*)g.F + pl * kW * kW) + ((2 * wv) * 16) * 64 + lane;
#pragma unroll
        for (int st = 0; st < kW / 16; ++st) wb[pl][0][st] = f[st * 64];
      }
    } else {
       // (chain 0's half of the weights; chain 1's is requested between the rows of the first fill below: the CU's vector-memory
       // path takes 64 bytes a clock, so 256 KB of fragments occupy it for ~4 k cycles whatever the wave does meanwhile.  Requesting
       // them under the first chain 0 instead -- a conditional definition -- made the allocator keep them in VGPRs: spills)
      const f32x4* f = (const f32x4*)g.F + ((2 * wv) * 32) * 64 + lane;
#pragma unroll
      for (int q = 0; q < kW / 8; ++q) {
        const f32x4 v = f[q * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[0][4 * q + e] = v[e];
      }
    }
    PIPE_STAMP(10);
    __syncthreads();  // (the ring is zeroed)
    PIPE_STAMP(11);
    rows_commit(r0, 0);
    rows_commit(r1, 1);
    PIPE_STAMP(12);
  }
  __syncthreads();
  PIPE_STAMP(13);
  const long long dump_tile = n_tiles;  // the spare tile behind the last one (VbT, U0out)
  if (BF16) {
    float* vb_rows = KIND == kBwd1 ? g.VbT + tile_of(0) * kRows * kW : nullptr;
    const bf16x8* f0 = (const bf16x8*)g.F + ((2 * wv + 1) * 16) * 64 + lane;
    const bf16x8* f1 = (const bf16x8*)((const __bf16*)g.F + kW * kW) + ((2 * wv + 1) * 16) * 64 + lane;
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 8) {
      SideIn ins[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ins[j] = fill_load_b(i0 + j, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        wb[0][1][i0 + j] = f0[(i0 + j) * 64];
        wb[1][1][i0 + j] = f1[(i0 + j) * 64];
        fill_step_b(i0 + j, ins[j], Xs[0], vb_rows);
      }
    }
  } else {
    float* vb_tile = KIND == kBwd1 ? g.VbT + tile_of(0) * 8 * 1024 : nullptr;
    const f32x4* f = (const f32x4*)g.F + ((2 * wv + 1) * 32) * 64 + lane;
#pragma unroll
    for (int i0 = 0; i0 < kRows; i0 += 8) {  // (eight rows' inputs requested before the first is used)
      SideIn ins[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ins[j] = fill_load(i0 + j, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 v = f[(i0 + j) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[1][4 * (i0 + j) + e] = v[e];
        fill_step(i0 + j, ins[j], Xs[0], vb_tile);
      }
    }
  }

  // one chain: 32 steps of (16-byte LDS read, 4 MFMAs, `side(kq)`) over the operand image Xc
  auto run_chain = [&](auto cbc, const float* __restrict__ Xc, auto side_in, auto side) {
    constexpr int cb = decltype(cbc)::value;
    const f32x4* xq = (const f32x4*)(Xc + l31 * 8 + half * 4);
    f32x4 x4 = xq[0];
    SideIn in = side_in(0);
#pragma unroll
    for (int kq = 0; kq < kW / 8; ++kq) {
      const f32x4 xn = xq[(kq + 1 < kW / 8 ? kq + 1 : kq) * (kXq / 4)];
      const SideIn in_next = side_in(kq + 1 < kW / 8 ? kq + 1 : kq);
      if (kq == 0) mfma_x4<true, false>(acc[cb], x4, b[cb][0], b[cb][1], b[cb][2], b[cb][3]);
      else if (kq == kW / 8 - 1) mfma_x4<false, true>(acc[cb], x4, b[cb][4 * kq], b[cb][4 * kq + 1], b[cb][4 * kq + 2], b[cb][4 * kq + 3]);
      else mfma_x4<false, false>(acc[cb], x4, b[cb][4 * kq], b[cb][4 * kq + 1], b[cb][4 * kq + 2], b[cb][4 * kq + 3]);
#ifndef DPLL_PIPE_NOSIDE
      side(kq, in);
#endif
      x4 = xn;
      in = in_next;
      // the order the step is meant to issue in: behind every MFMA a share of the step's other work (the scheduler otherwise
      // clusters the MFMAs, and the matrix pipe idles through the side work behind them)
#ifdef DPLL_PIPE_SCHED_GROUPS
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, kSideValu, 0);  // VALU
#if DPLL_PIPE_SCHED >= 2
        __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);          // one LDS access
#endif
#if DPLL_PIPE_SCHED >= 3
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);          // one global access
#endif
      }
#endif
#ifdef DPLL_PIPE_FENCE
      __builtin_amdgcn_sched_barrier(0);  // (diagnostic: nothing moves across a step.  Measured worse: the groups above are then
                                          // solved per step and mostly come out as four MFMAs in a row)
#endif
    }
  };
  // BF16: one chain = 16 steps of (two 16-byte LDS reads, three MFMAs, one epilogue register behind the first, a fill step behind
  // every other second one)
  auto run_chain_b = [&](auto cbc, const float* __restrict__ Xc, auto epi_in, auto fill_in, auto epi, auto fill) {
    constexpr int cb = decltype(cbc)::value;
    const bf16x8* x0 = (const bf16x8*)((const __bf16*)Xc + half * kBq + l31 * 8);
    const bf16x8* x1 = (const bf16x8*)((const __bf16*)Xc + kBopElems + half * kBq + l31 * 8);
    bf16x8 a0 = x0[0], a1 = x1[0];
    SideIn ein = epi_in(0), fin = fill_in(0);
#pragma unroll
    for (int st = 0; st < kW / 16; ++st) {
      const int nx = st + 1 < kW / 16 ? st + 1 : st;
      const bf16x8 a0n = x0[nx * (2 * kBq / 8)], a1n = x1[nx * (2 * kBq / 8)];
      const SideIn ein_next = epi_in(nx);
      if (st == 0) mfma_bf16<true, F16>(low[cb], a0, wb[1][cb][st]);
      else mfma_bf16<false, F16>(low[cb], a0, wb[1][cb][st]);
      epi(st, ein);
      __builtin_amdgcn_sched_barrier(0);
      mfma_bf16<false, F16>(low[cb], a1, wb[0][cb][st]);
      SideIn fin_next = fin;
      if (st & 1) {
        fill(st >> 1, fin);
        fin_next = fill_in((st >> 1) + 1 < 8 ? (st >> 1) + 1 : (st >> 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (st == 0) mfma_bf16<true, F16>(acc[cb], a0, wb[0][cb][st]);
      else mfma_bf16<false, F16>(acc[cb], a0, wb[0][cb][st]);
      a0 = a0n; a1 = a1n;
      ein = ein_next; fin = fin_next;
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_settle(acc[cb], low[cb]);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] += low[cb][r] * (F16 ? 1.f / kF16LowScale : 1.f);  // (F16: the low plane travels scaled by 2^11)
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

  // ---- main loop, iteration it: chain 0 of tile it  | epilogue of chain 1 of tile it - 1 | rows 0..15 of the fill of tile it + 1
  //                               chain 1 of tile it  | epilogue of chain 0 of tile it     | rows 16..31 of that fill
  //                               and the row data of tile it + 2 (requested at the top, committed to the ring at the end)
  PIPE_STAMP(1);
  for (long long it = 0; it < my_tiles; ++it) {
    __syncthreads();
    if (it < 3) PIPE_STAMP(2 + 4 * (int)it);
    const int cur = (int)(it & 1);
    const long long t_cur = tile_of(it), t_next = tile_of(it + 1), t_prev = tile_of(it - 1);
    const bool has_next = it + 1 < my_tiles, has_prev = it > 0;
    const RowRaw raw = rows_issue(has_next ? tile_of(it + 2) : n_tiles);  // (beyond the last: every row masked)
    if (KIND == kFwd2 && it >= 2) p_store(it - 2);
    float* Xn = Xs[cur ^ 1];
    const float* Xc = Xs[cur];
    float* vb_tile = KIND == kBwd1 ? g.VbT + (has_next ? t_next : dump_tile) * 8 * 1024 : nullptr;  // (BF16: 32 rows of 256)
    const int slot_next = (int)((it + 1) & 3), slot_cur = (int)(it & 3), slot_prev = (int)((it + 3) & 3);
    {  // chain 0 | epilogue of chain 1 of the previous tile
      float* u0_tile = KIND == kFwd2 ? g.U0out + ((has_prev ? t_prev : dump_tile) * 4 + wv) * 2048 + lane * 4 : nullptr;
      const int pbuf = (int)((it + 3) & 3);
      epi_begin(1, has_prev ? t_prev : t_cur);
      if constexpr (BF16)
        run_chain_b(C0{}, Xc, [&](int st) { return epi_load(1, st, slot_prev); }, [&](int i) { return fill_load_b(i, slot_next); },
                    [&](int st, const SideIn& in) { epi_step(1, st, in, u0_tile, pbuf); },
                    [&](int i, const SideIn& in) { fill_step_b(i, in, Xn, vb_tile); });
      else
      run_chain(C0{}, Xc,
                [&](int kq) { return (kq & 1) == 0 ? fill_load(kq >> 1, slot_next) : epi_load(1, kq >> 1, slot_prev); },
                [&](int kq, const SideIn& in) {
                  if ((kq & 1) == 0) fill_step(kq >> 1, in, Xn, vb_tile);
                  else epi_step(1, kq >> 1, in, u0_tile, pbuf);
                });
      epi_end(1, has_prev ? t_prev : n_tiles, pbuf);
    }
    if (it < 3) PIPE_STAMP(3 + 4 * (int)it);
    {  // chain 1 | epilogue of chain 0 of this tile
      float* u0_tile = KIND == kFwd2 ? g.U0out + (t_cur * 4 + wv) * 2048 + lane * 4 : nullptr;
      const int pbuf = (int)(it & 3);
      epi_begin(0, t_cur);
      if constexpr (BF16)
        run_chain_b(C1{}, Xc, [&](int st) { return epi_load(0, st, slot_cur); }, [&](int i) { return fill_load_b(8 + i, slot_next); },
                    [&](int st, const SideIn& in) { epi_step(0, st, in, u0_tile, pbuf); },
                    [&](int i, const SideIn& in) { fill_step_b(8 + i, in, Xn, vb_tile); });
      else
      run_chain(C1{}, Xc,
                [&](int kq) { return (kq & 1) == 0 ? fill_load(16 + (kq >> 1), slot_next) : epi_load(0, kq >> 1, slot_cur); },
                [&](int kq, const SideIn& in) {
                  if ((kq & 1) == 0) fill_step(16 + (kq >> 1), in, Xn, vb_tile);
                  else epi_step(0, kq >> 1, in, u0_tile, pbuf);
                });
      epi_end(0, t_cur, pbuf);
    }
    if (it < 3) PIPE_STAMP(4 + 4 * (int)it);
    rows_commit(raw, (int)((it + 2) & 3));
    if (it < 3) PIPE_STAMP(5 + 4 * (int)it);
  }

  // ---- drain: epilogue of chain 1 of the last tile; fwd2: the support points of the last two tiles ---------------------
  {
    const long long it = my_tiles, t_prev = tile_of(it - 1);
    float* u0_tile = KIND == kFwd2 ? g.U0out + (t_prev * 4 + wv) * 2048 + lane * 4 : nullptr;
    epi_begin(1, t_prev);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) epi_step(1, reg, epi_load(1, reg, (int)((it + 3) & 3)), u0_tile, (int)((it + 3) & 3));
    epi_end(1, t_prev, (int)((it + 3) & 3));
    if (KIND == kFwd2) {
      __syncthreads();
      if (it >= 2) p_store(it - 2);
      p_store(it - 1);
    }
  }
  PIPE_STAMP(14);
  if constexpr (KIND == kBwd1 && F16) {
    // the largest |r_bar| entry this workgroup met (a non-negative float: its bits order like the number), for icnn_bwd2's global
    // scale of the adjoint operand
    // (the rows are loaded by lanes 0..31 of wave 1: that wave folds their maxima and leaves the workgroup's in its own slot --
    // 1024 atomics on one word cost the launch ~10 us)
    if (wv == 1) {
      float m = rb_seen;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      if (lane == 0 && g.rbmax) g.rbmax[blockIdx.x] = __builtin_bit_cast(unsigned, m);
    }
  }
  if (KIND == kBwd1) {
    // the two halves of the wave hold different rows of the same columns (threads tid and tid ^ 32)
    __syncthreads();
    if (half == 0) {
      double* row = g.partial + (long long)blockIdx.x * kB1Cols;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        row[col[cb]] = totals[7 * cb][tid] + totals[7 * cb][tid ^ 32];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          // dWd1[i][col] = sum_rows r_bar[i] |wout[col]| m1
          row[kW + i * kW + col[cb]] = double(acol[cb]) * (totals[7 * cb + 1 + i][tid] + totals[7 * cb + 1 + i][tid ^ 32]);
          row[4 * kW + i * kW + col[cb]] = totals[7 * cb + 4 + i][tid] + totals[7 * cb + 4 + i][tid ^ 32];
        }
      }
    }
  }
  PIPE_STAMP(15);
}

// (d|Wh| = Vb^T U1, the split-K GEMM, stays with the 8-wave icnn_bwd2_mfma: three pipelined versions were built and measured in
// round 5 -- 64 x 64 output blocks with scalar-loaded lane masks 42 us, the same on a compact mask array 29.1 us, 64 x 256 blocks
// with sixteen accumulators per wave in AGPRs 30.4 us, main loop at 0.81 of the matrix rate -- none ahead of its 28.7-29.3 us at
// 4096 pairs (2 row tiles per wave: the launch is prologue- and reduction-bound); NOTEBOOK.md R5.1 has the account.)

inline int check(const char* what) {
  const hipError_t e = hipGetLastError();
  (void)what;
  return e == hipSuccess ? 0 : -4;
}

}  // namespace

namespace dpll_pipe {

int fwd1(hipStream_t stream, const float* x, long long ld, long long N, const IcnnWeights<float>& w, const float* Af, uint32_t* M1) {
  if (N <= 0) return 0;
  if (N >= (1LL << 31)) return -1;  // (row numbers are 32-bit inside the kernels)
  PipeArgs g{};
  g.x = x; g.ld = ld; g.N = N; g.w = w; g.F = Af; g.M1out = M1;
  hipLaunchKernelGGL(icnn_pipe_kernel<kFwd1>, dim3(blocks(N)), dim3(256), 0, stream, g);
  return check("icnn_pipe_kernel<fwd1>");
}

int fwd2(hipStream_t stream, const float* x, long long ld, long long N, const IcnnWeights<float>& w, const float* ATf, const float* a,
         const uint32_t* M1, float* U0t, float* P) {
  if (N <= 0) return 0;
  if (N >= (1LL << 31)) return -1;  // (row numbers are 32-bit inside the kernels)
  PipeArgs g{};
  g.x = x; g.ld = ld; g.N = N; g.w = w; g.F = ATf; g.a = a; g.M1in = M1; g.U0out = U0t; g.P = P;
  hipLaunchKernelGGL(icnn_pipe_kernel<kFwd2>, dim3(blocks(N)), dim3(256), 0, stream, g);
  return check("icnn_pipe_kernel<fwd2>");
}

int bwd1(hipStream_t stream, const float* x, long long ld, long long N, const IcnnWeights<float>& w, const float* Af, const float* a,
         const uint32_t* M1, const float* U0t, const float* RB, double* partial, float* VbT) {
  if (N <= 0) return 0;
  if (N >= (1LL << 31)) return -1;  // (row numbers are 32-bit inside the kernels)
  PipeArgs g{};
  g.x = x; g.ld = ld; g.N = N; g.w = w; g.F = Af; g.a = a; g.M1in = M1; g.U0in = U0t; g.RB = RB; g.partial = partial; g.VbT = VbT;
  hipLaunchKernelGGL(icnn_pipe_kernel<kBwd1>, dim3(blocks(N)), dim3(256), 0, stream, g);
  return check("icnn_pipe_kernel<bwd1>");
}

// ---- the split-bf16 form (2 planes): the same three kernels on the bf16 matrix cores; Ab / ATb = the planes icnn_prep_bf16_kernel<2>
// writes (dpll_mesh_bf16.hpp), Vb leaves row-major (N rounded up to whole tiles, 256) for icnn_bwd2_bf16
int fwd1_bf16(hipStream_t stream, const float* x, long long ld, long long N, const IcnnWeights<float>& w, const void* Ab, uint32_t* M1, bool f16) {
  if (N <= 0) return 0;
  if (N >= (1LL << 31)) return -1;  // (row numbers are 32-bit inside the kernels)
  PipeArgs g{};
  g.x = x; g.ld = ld; g.N = N; g.w = w; g.F = (const float*)Ab; g.M1out = M1;
  if (f16) hipLaunchKernelGGL((icnn_pipe_kernel<kFwd1, true, true>), dim3(blocks(N)), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((icnn_pipe_kernel<kFwd1, true>), dim3(blocks(N)), dim3(256), 0, stream, g);
  return check("icnn_pipe_kernel<fwd1, bf16>");
}

int fwd2_bf16(hipStream_t stream, const float* x, long long ld, long long N, const IcnnWeights<float>& w, const void* ATb, const float* a,
              const uint32_t* M1, float* U0t, float* P, bool f16) {
  if (N <= 0) return 0;
  if (N >= (1LL << 31)) return -1;  // (row numbers are 32-bit inside the kernels)
  PipeArgs g{};
  g.x = x; g.ld = ld; g.N = N; g.w = w; g.F = (const float*)ATb; g.a = a; g.M1in = M1; g.U0out = U0t; g.P = P;
  if (f16) hipLaunchKernelGGL((icnn_pipe_kernel<kFwd2, true, true>), dim3(blocks(N)), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((icnn_pipe_kernel<kFwd2, true>), dim3(blocks(N)), dim3(256), 0, stream, g);
  return check("icnn_pipe_kernel<fwd2, bf16>");
}

int bwd1_bf16(hipStream_t stream, const float* x, long long ld, long long N, const IcnnWeights<float>& w, const void* Ab, const float* a,
              const uint32_t* M1, const float* U0t, const float* RB, double* partial, float* Vb, bool f16, unsigned* rbmax) {
  if (N <= 0) return 0;
  if (N >= (1LL << 31)) return -1;  // (row numbers are 32-bit inside the kernels)
  PipeArgs g{};
  g.x = x; g.ld = ld; g.N = N; g.w = w; g.F = (const float*)Ab; g.a = a; g.M1in = M1; g.U0in = U0t; g.RB = RB; g.partial = partial; g.VbT = Vb; g.rbmax = rbmax;
  if (f16) hipLaunchKernelGGL((icnn_pipe_kernel<kBwd1, true, true>), dim3(blocks(N)), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((icnn_pipe_kernel<kBwd1, true>), dim3(blocks(N)), dim3(256), 0, stream, g);
  return check("icnn_pipe_kernel<bwd1, bf16>");
}

}  // namespace dpll_pipe
