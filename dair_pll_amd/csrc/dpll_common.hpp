// dpll_common.hpp -- what the translation units of libdpll_hip.so share: launch constants, cross-lane primitives,
// kernel dimensions, the partial-row / chain-matrix layout of the gradient reduction and the host-side handle types.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dpll.h"
#include "dpll_core.hpp"
#include "dpll_allreduce.hpp"

namespace {

using namespace dpll;

static_assert(sizeof(ModelDesc) == sizeof(dpll_model_desc_t), "ModelDesc must mirror dpll_model_desc_t");
static_assert(sizeof(SolverOpts) == sizeof(dpll_solver_opts_t), "SolverOpts must mirror dpll_solver_opts_t");
static_assert(kMaxJoints == DPLL_MAX_JOINTS && kMaxBodies == DPLL_MAX_BODIES && kMaxGeoms == DPLL_MAX_GEOMS, "limits");
static_assert(kGeomBox == DPLL_GEOM_BOX && kGeomSphere == DPLL_GEOM_SPHERE && kGeomPolygon == DPLL_GEOM_POLYGON && kGeomMesh == DPLL_GEOM_MESH, "geometry kinds");

constexpr int kWave = 64;
constexpr int kMaxLossBlocks = 2048;  // partial-sum rows; 8 one-wave workgroups per CU
constexpr int kSimds = 1024;          // 256 CUs x 4

// ---- cross-lane primitives --------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
  // old = 0 + bound_ctrl lets the backend fold the move into the consuming add (v_add_f32_dpp); every
  // source lane of the controls used here is inside the wave, so the value of `old` never shows
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  int lo = (int)(unsigned)(u & 0xffffffffull), hi = (int)(unsigned)(u >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  const unsigned long long r = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
  return __builtin_bit_cast(double, r);
}
template <int CTRL> __device__ __forceinline__ int dpp_mov(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
constexpr int kQuadXor1 = 0xB1;       // quad_perm [1,0,3,2]
constexpr int kQuadXor2 = 0x4E;       // quad_perm [2,3,0,1]
constexpr int kRowHalfMirror = 0x141; // lane i <-> 7 - i inside each 8 lanes
constexpr int kRowMirror = 0x140;     // lane i <-> 15 - i inside each 16 lanes

// P: racing copies of every item (SolverOpts::portfolio): an item owns G * P consecutive lanes, copy v the lanes
// [v G, (v + 1) G) of them; sums and searches stay inside a copy's G lanes
template <int G, int P = 1> struct GpuLanes {
  static_assert(G == 1 || G == 4 || G == 8 || G == 16, "lanes per item: one per contact slot (4, 8 or 16), or 1 (wide build)");
  static_assert(P == 1 || ((P == 2 || P == 4) && G >= 4 && G * P <= 16), "racing copies share a 16-lane DPP row with their item");
  static constexpr int kGroup = G;
  static constexpr int kVariants = P;
  static __device__ __forceinline__ int variant() { return (int)((threadIdx.x / G) & (P - 1)); }
  // OR over the copies of an item of a value that is uniform inside each copy
  static __device__ __forceinline__ int item_or(int x) {
    if (P == 1) return x;
    if (G == 4) x |= dpp_mov<kRowHalfMirror>(x);          // lanes i <-> 7 - i: the other quad of the 8
    if (G * P == 16) x |= dpp_mov<kRowMirror>(x);          // the other half of the row
    return x;
  }
  // the value one copy of the item holds (`mine` true in exactly one copy, the value uniform inside a copy), in every copy:
  // a sum in which the other copies enter as zero
  template <typename T> static __device__ __forceinline__ T item_pick(bool mine, T x) {
    if (P == 1) return x;
    x = mine ? x : T(0);
    if (G == 4) x += dpp_mov<kRowHalfMirror>(x);
    if (G * P == 16) x += dpp_mov<kRowMirror>(x);
    return x;
  }
  template <typename T> static __device__ __forceinline__ T group_sum(T x) {
    if (G == 1) return x;
    x += dpp_mov<kQuadXor1>(x);
    x += dpp_mov<kQuadXor2>(x);
    if (G >= 8) x += dpp_mov<kRowHalfMirror>(x);  // both quads hold their own sum -> mirror pairs them
    if (G == 16) x += dpp_mov<kRowMirror>(x);     // both halves of the 16-lane row
    return x;
  }
  static __device__ __forceinline__ bool group_any(bool x) {
    if (G == 1) return x;
    const unsigned long long b = __ballot(x);
    const int base = (threadIdx.x & (kWave - 1)) & ~(G - 1);
    return ((b >> base) & ((1ull << G) - 1ull)) != 0ull;
  }
  static __device__ __forceinline__ bool wave_any(bool x) { return __any(x) != 0; }
  static __device__ __forceinline__ int lane_in_group() { return (int)(threadIdx.x & (G - 1)); }
  // where an item's shared terms live (ItemStore, dpll_core.hpp): the general build (16 lanes per item) keeps one copy per
  // item in LDS, the specialised builds the caller's registers.  The per-item stride is padded so that the four groups of
  // a wave, which read the same member in one instruction, fall on different LDS banks.
  template <class Store> static __device__ __forceinline__ Store& item_store(Store& local) {
    if constexpr (G == 16 && P == 1) {
      struct Padded { Store s; char pad[(sizeof(Store) % 256 < 64 ? 64 : 0) + 8]; };
      __shared__ Padded items[kWave / G];
      return items[(threadIdx.x & (kWave - 1)) / G].s;
    } else {
      return local;
    }
  }
  // the best (largest value; ties: smallest index) candidate over the lanes of the group, left in every lane
  template <int CTRL, typename S> static __device__ __forceinline__ void best_step(S& value, int& index, S (&d)[3]) {
    const S ov = dpp_mov<CTRL>(value);
    const int oi = dpp_mov<CTRL>(index);
    const S o0 = dpp_mov<CTRL>(d[0]), o1 = dpp_mov<CTRL>(d[1]), o2 = dpp_mov<CTRL>(d[2]);
    // (values within kPairTie of each other are a tie: the lower candidate number wins, as inside a lane's own sequence)
    const bool tie = !(ov > value + S(kPairTie)) && !(value > ov + S(kPairTie));
    const bool take = tie ? oi < index : ov > value;
    value = take ? ov : value;
    index = take ? oi : index;
    d[0] = take ? o0 : d[0]; d[1] = take ? o1 : d[1]; d[2] = take ? o2 : d[2];
  }
  template <typename S> static __device__ __forceinline__ void group_best(S& value, int& index, S (&d)[3]) {
    if (G == 1) return;
    best_step<kQuadXor1>(value, index, d);
    best_step<kQuadXor2>(value, index, d);
    if (G >= 8) best_step<kRowHalfMirror>(value, index, d);
    if (G == 16) best_step<kRowMirror>(value, index, d);
  }
  // where the group keeps a vertex set (0 / 1) of the direction search: LDS, one block per group of the wave
  template <typename S> static __device__ __forceinline__ S (*pair_storage(int which, S (*local)[3]))[3] {
    if (G == 1) return local;
    __shared__ S sets[kWave / G][2][kMaxPolyVerts][3];
    return sets[(threadIdx.x & (kWave - 1)) / G][which];
  }
};

// sum over the whole wave of a value that is already uniform inside each group of G lanes, counting every
// group once; the result is valid in every lane.  Rows of 16 lanes are closed with the mirror controls, the four
// rows with row_bcast:15 / row_bcast:31 (the total lands in lane 63) and one v_readlane: no LDS permutes.
constexpr int kRowBcast15 = 0x142;  // lane 15 of each row -> every lane of the next row (row_mask 0xA)
constexpr int kRowBcast31 = 0x143;  // lane 31 -> rows 2 and 3 (row_mask 0xC)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_rows(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_rows(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  int lo = (int)(unsigned)(u & 0xffffffffull), hi = (int)(unsigned)(u >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  const unsigned long long r = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
  return __builtin_bit_cast(double, r);  // +0.0 in the rows outside ROW_MASK
}
__device__ __forceinline__ float read_lane63(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ double read_lane63(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffull), 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
template <int G, typename T> __device__ __forceinline__ T wave_sum_of_groups(T x) {
  if (G == 1) {
    x += dpp_mov<kQuadXor1>(x);
    x += dpp_mov<kQuadXor2>(x);
  }
  if (G <= 4) x += dpp_mov<kRowHalfMirror>(x);
  if (G <= 8) x += dpp_mov<kRowMirror>(x);
  x += dpp_rows<kRowBcast15, 0xA>(x);
  x += dpp_rows<kRowBcast31, 0xC>(x);
  return read_lane63(x);
}

// Chain from the batch-summed row (iota space) to the learnable parameters [theta | friction | lengths]: linear in the
// row, with a matrix that depends on the parameters only.  One EXTRA one-wave workgroup of every gradient kernel (it
// owns no items and runs on a SIMD the launch leaves idle) writes that matrix behind the partial rows while the other
// workgroups solve; the finalize kernel then applies it with <= 10 multiply-adds per parameter -- the theta -> iota
// duals are on nobody's critical path.
template <typename S>
__device__ __forceinline__ void theta_jacobian_column(int inertia_mode, const S* theta_b, int c, S (&dio)[kIota],
                                                      const double (*body_rot)[3][3] = nullptr) {
  DualT<S> th[10], io[kIota];
#pragma unroll
  for (int i = 0; i < 10; ++i) th[i] = DualT<S>(theta_b[i], i == c ? S(1) : S(0));
  theta_to_iota<DualT<S>>(th, inertia_mode, io);
  if (body_rot) rotate_iota<DualT<S>>(*body_rot, io);
#pragma unroll
  for (int i = 0; i < kIota; ++i) dio[i] = io[i].d;
}
// GP = numbers per geometry in the `lengths` block (3: a box; the general build: 3 kMaxPolyVerts); gd (general build
// only): a polygon's vertices are signed parameters (chain factor 1), lengths and radii enter through their absolute value;
// the body-body candidates' coefficients (each of its two geometries' friction entries) get a block of their own, fr_pair
template <typename T, typename P, int NB, int NG = NB, int GP = 3>
__device__ __forceinline__ void write_chain_matrix(int inertia_mode, const P* __restrict__ theta, const P* __restrict__ friction,
                                                   const P* __restrict__ lengths, double* __restrict__ chain,
                                                   const ModelDesc* gd = nullptr) {
  const int lane = threadIdx.x;
  if (lane < 10 * NB) {  // lane = (body, theta component c): column c of that body's Jacobian
    T th[10], dio[kIota];
#pragma unroll
    for (int i = 0; i < 10; ++i) th[i] = T(theta[10 * (lane / 10) + i]);
    theta_jacobian_column<T>(inertia_mode, th, lane % 10, dio, (gd && (gd->rotated & 1)) ? &gd->body_rot[lane / 10] : nullptr);
#pragma unroll
    for (int i = 0; i < kIota; ++i) chain[lane * kIota + i] = double(dio[i]);
  }
  double* fr_fac = chain + 100 * NB;
  double* len_sign = fr_fac + (NG + 1) * NG;
  double* fr_pair = len_sign + GP * NG;  // (general build) (NG + 1 k, kMaxPairs p)
  // d (2 ma mb / (ma + mb)) / d friction_k for the coefficient that combines friction entries ia and ib, mu = |friction|
  auto factor = [&](int k, int ia, int ib) {
    const double m0 = fabs(double(friction[ia])), mb = fabs(double(friction[ib]));
    const double den = (m0 + mb) * (m0 + mb);
    const double pk = double(friction[k]);
    const double sign = pk > 0.0 ? 1.0 : (pk < 0.0 ? -1.0 : 0.0);
    double fac = 0.0;
    if (k == ia) fac += 2.0 * mb * mb / den;
    if (k == ib) fac += 2.0 * m0 * m0 / den;
    return fac * sign;
  };
  if (lane < (NG + 1) * NG) {  // lane = (friction entry k, geometry b): slot b combines the ground (entry 0) with geometry b
    const int k = lane / NG, b = lane % NG;
    fr_fac[lane] = (gd && b >= kMaxGeoms) ? 0.0 : factor(k, 0, 1 + b);  // (the group behind the geometries has no coefficient)
  }
  if (gd && lane < (NG + 1) * kMaxPairs) {  // lane = (friction entry k, body-body candidate p): its two geometries
    const int k = lane / kMaxPairs, p = lane % kMaxPairs;
    fr_pair[lane] = p < gd->n_pairs ? factor(k, 1 + gd->pair_a[p], 1 + gd->pair_b[p]) : 0.0;
  }
  for (int e = lane; e < GP * NG; e += kWave) {
    const double pl = lengths ? double(lengths[e]) : 0.0;
    const bool polygon = gd && e / GP < kMaxGeoms && gd->geom_kind[e / GP] == kGeomPolygon;
    // (general build: the block behind the geometries holds no parameters -- its row entries carry the pairs' d/d mu)
    len_sign[e] = (gd && e / GP >= kMaxGeoms) ? 0.0 : (polygon ? 1.0 : (pl > 0.0 ? 1.0 : (pl < 0.0 ? -1.0 : 0.0)));
  }
}
// learnable parameter k = sum_{j < count} tot[tot0 + j] * chain[coef0 + j] with the row sum `tot` ([loss | iota | mu_pair | length])
// (a friction entry of the general build has a second segment: the body-body candidates' d/d mu, which sit in the first
// kMaxPairs entries of the row's length block behind the geometries)
struct ChainRow { int coef0, tot0, count, coef1, tot1, count1; };
template <int NB, int NG = NB, int GP = 3> __device__ __forceinline__ ChainRow chain_row(int k) {
  if (k < 10 * NB) return ChainRow{k * kIota, 1 + kIota * (k / 10), kIota, 0, 0, 0};
  if (k < 10 * NB + NG + 1) {
    ChainRow cr{100 * NB + (k - 10 * NB) * NG, 1 + kIota * NB, NG, 0, 0, 0};
    if (GP > 3) {
      cr.coef1 = 100 * NB + (NG + 1) * NG + GP * NG + (k - 10 * NB) * kMaxPairs;
      cr.tot1 = 1 + kIota * NB + NG + GP * kMaxGeoms;
      cr.count1 = kMaxPairs;
    }
    return cr;
  }
  const int i = k - (10 * NB + NG + 1);
  return ChainRow{100 * NB + (NG + 1) * NG + i, 1 + kIota * NB + NG + i, 1, 0, 0, 0};
}
template <int NB, int NG = NB, int GP = 3> __device__ __forceinline__ double apply_chain(const double* tot, const double* __restrict__ chain, int k) {
  const ChainRow cr = chain_row<NB, NG, GP>(k);
  double v = 0.0;
  for (int j = 0; j < cr.count; ++j) v += tot[cr.tot0 + j] * chain[cr.coef0 + j];
  for (int j = 0; j < cr.count1; ++j) v += tot[cr.tot1 + j] * chain[cr.coef1 + j];
  return v;
}


// NG = collision geometries (the two fast builds: one per body; the general build: always kMaxGeoms slots)
template <typename T, int NJ, int NG_ = NJ + 1, int GP_ = 3> struct Dims {
  static constexpr int NB = NJ + 1, NG = NG_, GP = GP_, NV = 6 + NJ, NQ = 7 + NJ, NX = 13 + 2 * NJ, K = kQuery * NG, G = K;
  static constexpr int IPW = kWave / G;                       // items per wave (lane-per-contact builds: G = 4 or 8)
  static constexpr int P = NB * 10 + (NG + 1) + NG * GP;      // learnable parameters [theta | friction | lengths]
  static constexpr int PI = 1 + P;                            // row stride of the partial sums; the output row [loss | d/d params]
  static constexpr int PIOTA = 1 + 10 * NB + (1 + GP) * NG;   // a partial row: [loss | d/d iota | d/d mu_pair | d/d |length|]
  // the chain matrix behind the rows: [d iota_b,i / d theta_b,c (NB, 10 c, 10 i) | d mu_pair,g / d friction_k (NG + 1 k, NG g) |
  // sign(length_params) (GP NG)], doubles
  // (general build, GP > 3: + d mu_pair,p / d friction_k (NG + 1 k, kMaxPairs p) of the body-body candidates)
  static constexpr int CHAIN = 100 * NB + (NG + 1) * NG + GP * NG + (GP > 3 ? (NG + 1) * kMaxPairs : 0);
};

template <typename T> struct Acc { using type = double; };  // cone residual / y accumulate in double


// Wave reduction of the items' d/d(iota, mu_pair, |length|): one row [loss | d/d iota (10 NB) | d/d mu_pair (NB) |
// d/d |length| (3 NB)] of double partial sums per wave, written by lane 63 (where the DPP row reduction lands).  The
// chain to the learnable parameters (theta, friction_params, length_params) is linear in the row, so it runs ONCE on
// the row sum in the finalize kernel instead of in every wave's prologue (forward-mode duals of theta -> iota cost
// ~3.5 k cycles per wave there).  Row stride D::PI (>= 1 + 14 NB).
template <int G, typename T> __device__ __forceinline__ T wave_sum_to_lane63(T x) {
  if (G == 1) {
    x += dpp_mov<kQuadXor1>(x);
    x += dpp_mov<kQuadXor2>(x);
  }
  if (G <= 4) x += dpp_mov<kRowHalfMirror>(x);
  if (G <= 8) x += dpp_mov<kRowMirror>(x);
  x += dpp_rows<kRowBcast15, 0xA>(x);
  x += dpp_rows<kRowBcast31, 0xC>(x);
  return x;  // the total over the groups in lane 63; other lanes hold partial sums
}

// WAVES > 1: a workgroup of several waves still writes ONE row -- every wave leaves its sum in LDS, the first PIOTA threads
// add them up in wave order (fixed order: bitwise reproducible)
template <typename T, int NJ, int G = Dims<T, NJ>::G, int NG = NJ + 1, int GP = 3, int WAVES = 1>
__device__ __forceinline__ void store_iota_row(const LossGrad<T, NJ, NG, GP>& acc, double loss_acc, double* __restrict__ partials) {
  using D = Dims<T, NJ, NG, GP>;
  using Lanes = GpuLanes<G>;
  double* dst = partials + (long long)blockIdx.x * D::PI;
  const bool writer = (threadIdx.x & (kWave - 1)) == kWave - 1;
  static_assert(WAVES == 1 || GP == 3, "multi-wave workgroups: the specialised builds");
  if constexpr (GP > 3) {
    // wide rows (the general build): element by element, nothing kept live
    const double l = wave_sum_to_lane63<G>(Lanes::group_sum(loss_acc));
    if (writer) dst[0] = l;
#pragma unroll
    for (int b = 0; b < D::NB; ++b)
#pragma unroll
      for (int i = 0; i < kIota; ++i) {
        const double v = double(wave_sum_to_lane63<G>(acc.g_iota[b][i]));
        if (writer) dst[1 + kIota * b + i] = v;
      }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const double m = double(wave_sum_to_lane63<G>(Lanes::group_sum(acc.g_mu[g])));
      if (writer) dst[1 + kIota * D::NB + g] = m;
#pragma unroll
      for (int i = 0; i < GP; ++i) {
        const double v = double(wave_sum_to_lane63<G>(Lanes::group_sum(acc.g_len[g][i])));
        if (writer) dst[1 + kIota * D::NB + NG + GP * g + i] = v;
      }
    }
    return;
  }
  double row[D::PIOTA];
  row[0] = wave_sum_to_lane63<G>(Lanes::group_sum(loss_acc));
#pragma unroll
  for (int b = 0; b < D::NB; ++b)
#pragma unroll
    for (int i = 0; i < kIota; ++i)  // g_iota is replicated inside the group: no group_sum
      row[1 + kIota * b + i] = double(wave_sum_to_lane63<G>(acc.g_iota[b][i]));
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    row[1 + kIota * D::NB + g] = double(wave_sum_to_lane63<G>(Lanes::group_sum(acc.g_mu[g])));
#pragma unroll
    for (int i = 0; i < GP; ++i)
      row[1 + kIota * D::NB + NG + GP * g + i] = double(wave_sum_to_lane63<G>(Lanes::group_sum(acc.g_len[g][i])));
  }
  if constexpr (WAVES > 1) {
    __shared__ double wave_rows[WAVES][D::PIOTA];
    if (writer) {
#pragma unroll
      for (int i = 0; i < D::PIOTA; ++i) wave_rows[threadIdx.x / kWave][i] = row[i];
    }
    __syncthreads();
    if (threadIdx.x < D::PIOTA) {
      double v = wave_rows[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) v += wave_rows[w][threadIdx.x];
      dst[threadIdx.x] = v;
    }
  } else if (writer) {
#pragma unroll
    for (int i = 0; i < D::PIOTA; ++i) dst[i] = row[i];
  }
}


}  // namespace

// Adam on the learnable parameters, done by the threads that have just written their gradient entry
// (experiment.py:213-228: torch.optim.Adam; the update of torch's _single_tensor_adam, no amsgrad): a training step is then the
// loss launch and the kernel that ends it -- the optimizer costs no launch of its own.  params == nullptr: no update.
struct AdamArgs {
  void* params;      // (P,) the flat buffer the kernels read their parameters from: updated in place
  void* exp_avg;     // (P,)
  void* exp_avg_sq;  // (P,)
  double* state;     // (3,) [steps taken, beta1^steps, beta2^steps]: advanced once per step (no pow on the device)
  double lr, beta1, beta2, eps, weight_decay;
};
// parameter k with gradient g; pow1 / pow2 = beta^steps of THIS step; returns the updated value
template <typename T> __device__ __forceinline__ T adam_apply(const AdamArgs& adam, long long k, double g_in, double pow1, double pow2) {
  T* p = (T*)adam.params + k;
  T* m = (T*)adam.exp_avg + k;
  T* v = (T*)adam.exp_avg_sq + k;
  const double g = g_in + adam.weight_decay * double(*p);
  const double m1 = double(*m) + (g - double(*m)) * (1.0 - adam.beta1);
  const double v1 = adam.beta2 * double(*v) + (1.0 - adam.beta2) * g * g;
  const double denom = sqrt(v1) / sqrt(1.0 - pow2) + adam.eps;
  *m = T(m1);
  *v = T(v1);
  const T updated = T(double(*p) - (adam.lr / (1.0 - pow1)) * m1 / denom);
  *p = updated;
  return updated;
}
// the running products of a step, read by every thread before one of them advances the state
__device__ __forceinline__ void adam_powers(const AdamArgs& adam, double& steps, double& pow1, double& pow2) {
  steps = adam.state[0] + 1.0;
  pow1 = adam.state[1] * adam.beta1;
  pow2 = adam.state[2] * adam.beta2;
}
// which entries of the general build's flat parameter buffer [theta | friction (1 + slots) | lengths (slots, stride)] belong to a
// parameter of the model (the rest is padding, which an optimizer must leave alone: weight decay would move it)
__host__ __device__ inline bool general_param_is_real(const dpll::ModelDesc& md, int k) {
  const int nb = md.n_joints + 1, slots = dpll::kGenSlots, stride = dpll::GeneralDesc::kGeoStride;
  if (k < 10 * nb) return true;
  k -= 10 * nb;
  if (k < 1 + slots) return k <= md.n_geoms;
  k -= 1 + slots;
  const int g = k / stride, e = k % stride;
  if (g >= md.n_geoms) return false;
  const int kind = md.geom_kind[g];
  return kind == dpll::kGeomBox ? e < 3 : (kind == dpll::kGeomSphere ? e < 1 : (kind == dpll::kGeomPolygon ? e < 3 * md.geom_nverts[g] : false));
}

struct dpll_ar {
  int rank, world;
  void* local;             // this rank's receive buffer (uncached device memory)
  uint32_t* state;         // [0] call counter, [1] error word (device)
  dpll_arx::Peers peers;
  void* opened[dpll_arx::kMaxWorld];
};

namespace dpll_forest { struct ForestDesc; }
struct dpll_model {
  dpll::ModelDesc desc;
  dpll::SolverOpts opts[2];
  // a model of the forest build (csrc/dpll_forest.hip): its description on the host and, from the first launch on, on the device
  dpll_forest::ForestDesc* forest = nullptr;
  // the forest description's device copies, one per device the handle has launched on (made at the first launch there, under
  // a mutex, never inside a stream capture: dpll_forest.hip device_desc)
  static constexpr int kMaxDevices = 16;
  void* forest_dev[kMaxDevices] = {};
};

// error reporting shared by the translation units (defined in dpll_kernels.hip)
int dpll_fail(int code, const char* fmt, const char* detail = "");
int dpll_check_launch(const char* what);
