// dpll_forest.hip -- the FOREST build on the device: one wave per item, an item's blocks in LDS with run-time sizes.
//
// For systems the register-resident builds do not take (dpll_kernels.hip: cube / elbow; dpll_general.hip: one model of up to
// 3 joints, 3 geometries, 4 candidates): several models in one system (the reference's init_urdfs: Dict[str, str],
// multibody_learnable_system.py:51-54, drake_utils.py:309-335), up to 16 bodies, 12 geometries, 16 body-body candidates, 64
// contacts, 32 velocities.  The per-item program is csrc/dpll_forest.hpp -- the same source the host checker runs with a team
// of one lane; here the team is a wavefront: lanes take bodies / contacts / matrix entries in strides, phases end at a
// workgroup barrier (one wave per workgroup), sums over the team are DPP wave reductions, the direction search of a body-body
// candidate spreads its candidate directions over the 64 lanes.  One kernel per dtype and entry point: no template ranges.
// Gradients: every workgroup keeps ONE partial row in LDS across the items it loops over and writes it at the end; rows are
// folded 64 at a time and the finalize kernel chains the sum to [theta | friction | lengths] (one thread per parameter).
#include <hip/hip_runtime.h>

#include <mutex>

#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>

#include "dpll_common.hpp"

// Diagnostic build only (-DDPLL_FOREST_STAMPS, never shipped; tools/diag/forest_stamps.py): shader-clock ticks per phase of the
// per-item program, summed over the items of workgroup 0
#ifdef DPLL_FOREST_STAMPS
__device__ unsigned long long g_fstamps[32][2];
__device__ unsigned long long g_flast;
#define DPLL_FSTAMP(slot)                                                                        \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    unsigned long long t_;                                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                                                   \
      if ((slot) != 0) { g_fstamps[slot][0] += t_ - g_flast; g_fstamps[slot][1] += 1; }          \
      g_flast = t_;                                                                              \
    }                                                                                            \
  } while (0)
#endif
#include "dpll_forest.hpp"
#include "dpll_forest_api.hpp"

namespace {

using namespace dpll_forest;

static_assert(sizeof(ForestDesc) == sizeof(dpll_forest_desc_t), "ForestDesc must mirror dpll_forest_desc_t");
static_assert(dpll_forest::kMaxBodies == DPLL_FOREST_MAX_BODIES && dpll_forest::kMaxGeoms == DPLL_FOREST_MAX_GEOMS &&
                  dpll_forest::kMaxPairs == DPLL_FOREST_MAX_PAIRS && kMaxContacts == DPLL_FOREST_MAX_CONTACTS && kMaxV == DPLL_FOREST_MAX_V,
              "limits");

// the lane-group policy of dpll_core.hpp's direction search for a team of W = 32 or 64 lanes (butterflies through the LDS
// crossbar: the partner of lane l at distance off < W is in the same team)
template <int W> struct ShflLanes {
  static constexpr int kGroup = W;
  static constexpr int kVariants = 1;
  static __device__ __forceinline__ int lane_in_group() { return (int)(threadIdx.x & (W - 1)); }
  template <typename S> static __device__ __forceinline__ void group_best(S& value, int& index, S (&d)[3]) {
#pragma unroll
    for (int off = 1; off < W; off <<= 1) {
      const S ov = __shfl_xor(value, off);
      const int oi = __shfl_xor(index, off);
      const S o0 = __shfl_xor(d[0], off), o1 = __shfl_xor(d[1], off), o2 = __shfl_xor(d[2], off);
      const bool tie = !(ov > value + S(kPairTie)) && !(value > ov + S(kPairTie));  // (dpll_common.hpp best_step)
      const bool take = tie ? oi < index : ov > value;
      value = take ? ov : value;
      index = take ? oi : index;
      d[0] = take ? o0 : d[0]; d[1] = take ? o1 : d[1]; d[2] = take ? o2 : d[2];
    }
  }
};
using WaveLanes = ShflLanes<kWave>;
// The team that works on an item: the whole wavefront (G = 64: one item per wave), half of it (G = 32: two items per wave --
// an instruction costs a lone wave its four cycles whether 9 lanes have work or 64, so systems whose phases are narrower
// than 32 lanes lose nothing per item and a wave finishes two) or a 16-lane DPP row (G = 16: four items per wave, for
// systems small enough that most phases would leave three quarters of a wave idle).  Barriers are workgroup barriers
// either way (one wave per workgroup): teams of one wave walk through every phase together.
// f(integral_constant<J>) for J = FROM, FROM + 1, ... while J < n (n uniform over the wave: ONE branch leaves the unrolled chain)
template <int J, int JMAX> struct RowsUp {
  template <class F> static __device__ __forceinline__ void run(int n, F&& f) {
    if constexpr (J < JMAX) {
      if (J < n) {
        f(std::integral_constant<int, J>{});
        RowsUp<J + 1, JMAX>::run(n, f);
      }
    }
  }
};
// ... and for J = JMAX - 1 down to 0, those with J < n
template <int J> struct RowsDown {
  template <class F> static __device__ __forceinline__ void run(int n, F&& f) {
    if constexpr (J >= 0) {
      if (J < n) f(std::integral_constant<int, J>{});
      RowsDown<J - 1>::run(n, f);
    }
  }
};

template <int G> struct GroupTeam {
  static_assert(G == 16 || G == 32 || G == kWave, "a DPP row, half the wave or the whole wave");
  static constexpr int kSize = G;
  static constexpr int kTeams = kWave / G;
  static __device__ __forceinline__ int rank() { return (int)(threadIdx.x & (G - 1)); }
  static __device__ __forceinline__ int team() { return (int)(threadIdx.x / G); }
  static __device__ __forceinline__ void sync() { __syncthreads(); }
  template <typename T> static __device__ __forceinline__ T sum(T x) {
    if constexpr (G == kWave) return wave_sum_of_groups<1>(x);
    else if constexpr (G == 32) {
      x = GpuLanes<16>::group_sum(x);  // both rows of the half hold their own sum
      return x + __shfl_xor(x, 16);
    } else return GpuLanes<16>::group_sum(x);
  }
  static __device__ __forceinline__ bool any(bool x) {
    if constexpr (G == kWave) return __any(x) != 0;
    else if constexpr (G == 32) return (unsigned)(__ballot(x) >> (32 * team())) != 0u;
    else return GpuLanes<16>::group_any(x);
  }
  static __device__ __forceinline__ bool wave_any(bool x) { return __any(x) != 0; }
  using Lanes = typename std::conditional<G == 16, GpuLanes<16>, ShflLanes<G>>::type;

  // ---- dense factorisations with ONE ROW PER LANE in registers (dpll_forest.hpp cholesky / chol_solve hand over to these) ------
  // The column-by-column factorisation in LDS pays two barriers and four dependent LDS round trips per column, on a wave that
  // has nothing else to run meanwhile.  Here lane i keeps row i in registers and column j reaches the others as a lane
  // broadcast (v_readlane: a scalar; teams of 16: DPP row_newbcast): n^2 / 2 multiply-adds per lane and no memory in between.
  static constexpr bool kLaneRows = true;
  static constexpr int kRowsMax = G == 16 ? 8 : dpll_forest::kMaxV;  // (teams of 16 are given systems of <= 8 velocities)
  template <int J> static __device__ __forceinline__ int bcast_word(int x) {
    if constexpr (G == kWave) return __builtin_amdgcn_readlane(x, J);
    else if constexpr (G == 32) {  // lane J of either half, each team keeps its own
      const int lo = __builtin_amdgcn_readlane(x, J), hi = __builtin_amdgcn_readlane(x, 32 + J);
      return threadIdx.x & 32 ? hi : lo;
    } else return __builtin_amdgcn_update_dpp(x, x, 0x150 + J, 0xf, 0xf, false);  // row_newbcast:J
  }
  template <int J> static __device__ __forceinline__ float bcast(float x) { return __int_as_float(bcast_word<J>(__float_as_int(x))); }
  template <int J> static __device__ __forceinline__ double bcast(double x) {
    const int lo = bcast_word<J>(__double2loint(x)), hi = bcast_word<J>(__double2hiint(x));
    return __hiloint2double(hi, lo);
  }
  // Aio (n x n, row-major, lower triangle used) -> L in its strict lower triangle, invd = 1 / diag(L); same operations in the
  // same order as the LDS version (right-looking: entry (i, k) loses L_ij L_kj for j ascending)
  // (the entry points are real functions -- one copy per kernel, several call sites -- so they say themselves what inlining
  // would have told the compiler: the pointers are LDS, n is the same in every lane; and they pick the unrolled length)
  template <typename S> using Lds = __attribute__((address_space(3))) S*;
  template <typename S> using LdsConst = const __attribute__((address_space(3))) S*;
  template <typename S> static __device__ __forceinline__ void factor_rows(S* Aio_, S* invd_, int n_, bool fast_) {
    const int n = __builtin_amdgcn_readfirstlane(n_);
    const bool fast = __builtin_amdgcn_readfirstlane(fast_ ? 1 : 0) != 0;
    Lds<S> Aio = (Lds<S>)Aio_;
    Lds<S> invd = (Lds<S>)invd_;
    if (n <= 8) factor_rows_n<S, 8>(Aio, invd, n, fast);
    else if constexpr (kRowsMax > 8) {
      if (n <= 16) factor_rows_n<S, 16>(Aio, invd, n, fast);
      else factor_rows_n<S, kRowsMax>(Aio, invd, n, fast);
    }
  }
  template <typename S> static __device__ __forceinline__ void solve_rows(const S* L_, const S* invd_, const S* b_, S* x_, int n_) {
    const int n = __builtin_amdgcn_readfirstlane(n_);
    LdsConst<S> L = (LdsConst<S>)L_;
    LdsConst<S> invd = (LdsConst<S>)invd_;
    LdsConst<S> b = (LdsConst<S>)b_;
    Lds<S> x = (Lds<S>)x_;
    if (n <= 8) solve_rows_n<S, 8>(L, invd, b, x, n);
    else if constexpr (kRowsMax > 8) {
      if (n <= 16) solve_rows_n<S, 16>(L, invd, b, x, n);
      else solve_rows_n<S, kRowsMax>(L, invd, b, x, n);
    }
  }
  // Aio (n x n, row-major, lower triangle used) -> L in its strict lower triangle (the upper one is scratch afterwards),
  // invd = 1 / diag(L); same operations in the same order as the LDS version (right-looking: entry (i, k) loses L_ij L_kj
  // for j ascending)
  template <typename S, int NMAX> static __device__ __forceinline__ void factor_rows_n(Lds<S> Aio, Lds<S> invd, int n, bool fast) {
    const int i = rank();
    const bool mine = i < n;
    const int row = mine ? i : 0;  // (lanes without a row follow along on a copy of row 0: no branches around the loads)
    S r[NMAX];
    sync();
    RowsUp<0, NMAX>::run(n, [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      r[j] = Aio[row * n + j];
    });
    S my_id = S(1);
    RowsUp<0, NMAX>::run(n, [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const S djj = bcast<j>(r[j]);
      S id;
      if constexpr (sizeof(S) == 4) id = fast ? fast_rsqrt(djj) : S(1) / tsqrt(djj);
      else id = S(1) / tsqrt(djj);
      r[j] = r[j] * id;
      my_id = i == j ? id : my_id;
      RowsUp<j + 1, NMAX>::run(n, [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        r[k] -= r[j] * bcast<k>(r[j]);
      });
    });
    if (mine) {
      RowsUp<0, NMAX>::run(n, [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        Aio[i * n + j] = r[j];
      });
      invd[i] = my_id;
    }
    sync();
  }
  // x = (L L^T)^-1 b by substitution: lane i holds row i of L (forward) and column i (backward)
  template <typename S, int NMAX>
  static __device__ __forceinline__ void solve_rows_n(LdsConst<S> L, LdsConst<S> invd, LdsConst<S> b, Lds<S> x, int n) {
    const int i = rank();
    const bool mine = i < n;
    const int me = mine ? i : 0;
    S row[NMAX], col[NMAX];
    RowsUp<0, NMAX>::run(n, [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const S lower = L[me * n + j], upper = L[j * n + me];
      row[j] = (mine && j < i) ? lower : S(0);
      col[j] = (mine && j > i) ? upper : S(0);
    });
    const S my_id = mine ? S(invd[me]) : S(0);
    S t = mine ? S(b[me]) : S(0);
    RowsUp<0, NMAX>::run(n, [&](auto jc) {  // y = L^-1 b: y_j is final once rows 0 .. j - 1 have been taken out
      constexpr int j = decltype(jc)::value;
      const S yj = bcast<j>(t * my_id);
      t -= row[j] * yj;
    });
    t = t * my_id;
    RowsDown<NMAX - 1>::run(n, [&](auto jc) {  // x = L^-T y
      constexpr int j = decltype(jc)::value;
      const S xj = bcast<j>(t * my_id);
      t -= col[j] * xj;
    });
    if (mine) x[i] = t * my_id;
    sync();
  }
};
using WaveTeam = GroupTeam<kWave>;

// the description from device memory into LDS, by the whole wave
__device__ __forceinline__ void fetch_desc(const ForestDesc* __restrict__ src, ForestDesc& dst) {
  const int* s = reinterpret_cast<const int*>(src);
  int* d = reinterpret_cast<int*>(&dst);
  for (int i = threadIdx.x; i < (int)(sizeof(ForestDesc) / 4); i += kWave) d[i] = s[i];
  __syncthreads();
}

extern __shared__ __align__(16) char forest_smem[];

// waves per SIMD the item kernels are compiled for (register budget 512 / this): an item's wave spends most of its time waiting
// on LDS round trips and barriers, so resident waves -- not registers per wave -- are what fills a SIMD
#ifndef DPLL_FOREST_OCC
#define DPLL_FOREST_OCC 2
#endif
// partial rows (= one-wave workgroups) of a gradient launch at most: DPLL_FOREST_OCC per SIMD on every CU.  Round 5, 4096 items,
// f32 loss + gradients: OCC 3 with 3072 rows / OCC 4 with 4096 rows (every item resident at once) against the shipped 2 / 2048:
// cube 168 / 188 vs 141 us, two_cubes 513 / 522 vs 364, gripper 516 / 549 vs 350, chain6 2173 / 2230 vs 1926 -- the register
// budget of 170 / 128 costs more than the extra resident waves return
#ifndef DPLL_FOREST_MAX_ROWS
#define DPLL_FOREST_MAX_ROWS kMaxLossBlocks
#endif
constexpr int kForestMaxRows = DPLL_FOREST_MAX_ROWS;

// ---- ContactNets loss, forward + backward ------------------------------------------------------------------------------------
// G: lanes per item (16: four items per wave; 64: one).  LDS: [the teams' partial rows | the teams' arenas]
template <typename T, int G>
__global__ __launch_bounds__(kWave, DPLL_FOREST_OCC) void forest_loss_kernel(const ForestDesc* __restrict__ fdp, SolverOpts opt, const T* __restrict__ theta,
                                                            const T* __restrict__ friction, const T* __restrict__ lengths,
                                                            const T* __restrict__ x, long long ld_x, const T* __restrict__ xp, long long ld_xp,
                                                            long long batch, const T* __restrict__ weights, double scale, T* __restrict__ loss,
                                                            T* __restrict__ force, int* __restrict__ iters, double* __restrict__ partials,
                                                            int want_grad, int row_stride, unsigned arena_stride,
                                                            const T* __restrict__ u, long long ld_u) {
  using Team = GroupTeam<G>;
  __shared__ ForestDesc fd;
  fetch_desc(fdp, fd);
  const int width = row_width(fd), team = Team::team(), rank = Team::rank();
  const size_t row_bytes = ((size_t)width * sizeof(double) + 15) & ~(size_t)15;
  double* row = reinterpret_cast<double*>(forest_smem + (size_t)team * row_bytes);
  Arena<T, double> A;
  A.carve(forest_smem + Team::kTeams * row_bytes + (size_t)team * arena_stride, fd.n_bodies, fd.n_v, fd.n_q, fd.n_contacts, fd.n_geoms, fd.n_pairs);
  for (int e = rank; e < width; e += G) row[e] = 0.0;
  Forest<T, double, Team> prog(fd, A);
  prog.derive(theta, friction, lengths);
  prog.load_actuation((const T*)nullptr);  // (no inputs: B u = 0 until an item brings its own)
  const int K = fd.n_contacts;
  for (long long base = (long long)blockIdx.x * Team::kTeams; base < batch; base += (long long)gridDim.x * Team::kTeams) {
    const long long mine = base + team;
    const bool valid = mine < batch;
    const long long item = valid ? mine : batch - 1;  // (an idle team shadows the last item with weight zero: it meets every barrier)
    const T w = valid ? T(scale) * (weights ? weights[item] : T(1)) : T(0);
    int n_it = 0;
    if (u) prog.load_actuation(u + item * ld_u);
    const T L = prog.loss(x + item * ld_x, xp + item * ld_xp, lengths, opt, w, want_grad != 0, row, n_it);
    if (valid && rank == 0) {
      if (loss) loss[item] = L;
      if (iters) iters[item] = n_it;
    }
    if (force && valid) {  // reference ordering: normals, then (t_x, t_y) per contact (multibody_terms.py:415-426)
      T* dst = force + item * (3 * K);
      for (int c = rank; c < K; c += G) {
        dst[c] = A.force[3 * c + 2];
        dst[K + 2 * c] = A.force[3 * c];
        dst[K + 2 * c + 1] = A.force[3 * c + 1];
      }
    }
    __syncthreads();
  }
  if (!want_grad) return;
  __syncthreads();
  for (int e = threadIdx.x; e < width; e += kWave) {  // the teams' rows in team order: one row per workgroup
    double v = 0.0;
    for (int t = 0; t < Team::kTeams; ++t) v += reinterpret_cast<const double*>(forest_smem + (size_t)t * row_bytes)[e];
    partials[(long long)blockIdx.x * row_stride + e] = v;
  }
}

// ---- rows -> parameters: blocks of kFold rows are summed first, the finalize kernel sums those and applies the chain ------------
constexpr int kFold = 64;
constexpr int kRowThreads = 512;  // one thread per column (a row has at most 1 + 160 + 12 + 16 + 288 = 477 entries)
__global__ __launch_bounds__(kRowThreads) void forest_fold_kernel(const double* __restrict__ partials, int n_rows, int width, int row_stride,
                                                                  double* __restrict__ folded) {
  const int col = threadIdx.x;
  if (col >= width) return;
  const int r0 = (int)blockIdx.x * kFold, r1 = r0 + kFold < n_rows ? r0 + kFold : n_rows;
  double s = 0.0;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partials[(long long)(r + u) * row_stride + col];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; r < r1; ++r) s += partials[(long long)r * row_stride + col];
  folded[(long long)blockIdx.x * row_stride + col] = s;
}
template <typename T>
__global__ __launch_bounds__(kRowThreads) void forest_finalize_kernel(const ForestDesc* __restrict__ fdp, const T* __restrict__ theta,
                                                                      const T* __restrict__ friction, const T* __restrict__ lengths,
                                                                      const double* __restrict__ folded, int n_rows, int row_stride,
                                                                      T* __restrict__ grad, T* __restrict__ loss_total, AdamArgs adam) {
  __shared__ ForestDesc fd;
  __shared__ double tot[kRowThreads];
  {
    const int* s = reinterpret_cast<const int*>(fdp);
    int* d = reinterpret_cast<int*>(&fd);
    for (int i = threadIdx.x; i < (int)(sizeof(ForestDesc) / 4); i += kRowThreads) d[i] = s[i];
  }
  __syncthreads();
  const int width = row_width(fd), col = threadIdx.x;
  double s = 0.0;
  if (col < width)
    for (int r = 0; r < n_rows; ++r) s += folded[(long long)r * row_stride + col];
  tot[col] = s;
  double steps = 0.0, pow1 = 0.0, pow2 = 0.0;
  if (adam.params) adam_powers(adam, steps, pow1, pow2);
  __syncthreads();  // (column totals in place; every thread has read the optimizer state before thread 0 advances it)
  const int n_params = param_count(fd);
  // the chain reads the parameters the gradient was taken at: every thread finishes its chains before any parameter moves
  double mine[2] = {0.0, 0.0};  // (at most 10 * 16 + 13 + 288 = 461 parameters: one per thread; two keeps the loop general)
  int count = 0;
  for (int k = threadIdx.x; k < n_params; k += kRowThreads) mine[count++ & 1] = chain_param(fd, theta, friction, lengths, tot, k);
  __syncthreads();
  count = 0;
  for (int k = threadIdx.x; k < n_params; k += kRowThreads) {
    const T g = T(mine[count++ & 1]);
    grad[k] = g;
    // fused training step: Adam on parameter k by the thread that wrote its gradient (padding entries are left alone)
    if (adam.params && param_is_real(fd, k)) adam_apply<T>(adam, k, double(g), pow1, pow2);
  }
  if (threadIdx.x == 0) {
    if (loss_total) *loss_total = T(tot[0]);
    if (adam.params) { adam.state[0] = steps; adam.state[1] = pow1; adam.state[2] = pow2; }
  }
}

// ---- simulation: `steps` VelocityIntegrator steps per item, the current state in LDS ---------------------------------------------
template <typename T, int G>
__global__ __launch_bounds__(kWave, DPLL_FOREST_OCC) void forest_simulate_kernel(const ForestDesc* __restrict__ fdp, SolverOpts opt, const T* __restrict__ theta,
                                                                const T* __restrict__ friction, const T* __restrict__ lengths,
                                                                const T* __restrict__ x0, long long ld_x, long long batch, long long steps,
                                                                T* __restrict__ out, long long ld_item, long long ld_step, int write_x0,
                                                                int* __restrict__ iters, unsigned arena_stride, const T* __restrict__ u,
                                                                long long ld_u) {
  using Team = GroupTeam<G>;
  __shared__ ForestDesc fd;
  fetch_desc(fdp, fd);
  const int nx = fd.n_q + fd.n_v, team = Team::team(), rank = Team::rank();
  const size_t state_bytes = ((size_t)2 * nx * sizeof(T) + 15) & ~(size_t)15;
  T* cur = reinterpret_cast<T*>(forest_smem + (size_t)team * state_bytes);
  T* nxt = cur + nx;
  Arena<T, double> A;
  A.carve(forest_smem + Team::kTeams * state_bytes + (size_t)team * arena_stride, fd.n_bodies, fd.n_v, fd.n_q, fd.n_contacts, fd.n_geoms, fd.n_pairs);
  Forest<T, double, Team> prog(fd, A);
  prog.derive(theta, friction, lengths);
  prog.load_actuation((const T*)nullptr);  // (no inputs: B u = 0 until an item brings its own)
  for (long long base = (long long)blockIdx.x * Team::kTeams; base < batch; base += (long long)gridDim.x * Team::kTeams) {
    const long long mine = base + team;
    const bool valid = mine < batch;
    const long long item = valid ? mine : batch - 1;
    for (int i = rank; i < nx; i += G) cur[i] = x0[item * ld_x + i];
    __syncthreads();
    T* dst = out + item * ld_item;
    if (write_x0) {
      if (valid)
        for (int i = rank; i < nx; i += G) dst[i] = cur[i];
      dst += ld_step;
    }
    int total = 0;
    if (u) prog.load_actuation(u + item * ld_u);  // (the launcher passes inputs for one step only: dpll_step)
    for (long long s = 0; s < steps; ++s) {
      total += prog.step(cur, lengths, opt, nxt);
      for (int i = rank; i < nx; i += G) {
        const T v = nxt[i];
        cur[i] = v;
        if (valid) dst[i] = v;
      }
      __syncthreads();
      dst += ld_step;
    }
    if (iters && valid && rank == 0) iters[item] = total;
  }
}

// ---- backward of one step: parameter rows and (STATE) the state adjoint; double arithmetic ------------------------------------------
template <typename T, bool STATE>
__global__ __launch_bounds__(kWave) void forest_step_backward_kernel(const ForestDesc* __restrict__ fdp, SolverOpts opt, const T* __restrict__ theta,
                                                                     const T* __restrict__ friction, const T* __restrict__ lengths,
                                                                     const T* __restrict__ x, long long ld_x, const T* __restrict__ gx,
                                                                     long long ld_g, long long batch, double* __restrict__ partials,
                                                                     int row_stride, T* __restrict__ xbar_out, long long ld_xb,
                                                                     const T* __restrict__ u, long long ld_u) {
  __shared__ ForestDesc fd;
  fetch_desc(fdp, fd);
  const int width = row_width(fd);
  double* row = reinterpret_cast<double*>(forest_smem);
  Arena<double, double> A;
  Arena<DualT<double>, DualT<double>> B;
  size_t off = ((size_t)width * sizeof(double) + 15) & ~(size_t)15;
  off += A.carve(forest_smem + off, fd.n_bodies, fd.n_v, fd.n_q, fd.n_contacts, fd.n_geoms, fd.n_pairs);
  if (STATE) B.carve(forest_smem + off, fd.n_bodies, fd.n_v, fd.n_q, fd.n_contacts, fd.n_geoms, fd.n_pairs, true);
  else B.carve(nullptr, fd.n_bodies, fd.n_v, fd.n_q, fd.n_contacts, fd.n_geoms, fd.n_pairs, true);
  for (int e = threadIdx.x; e < width; e += kWave) row[e] = 0.0;
  Forest<double, double, WaveTeam> prog(fd, A);
  prog.derive(theta, friction, lengths);
  prog.load_actuation((const T*)nullptr);  // (no inputs: B u = 0 until an item brings its own)
  ForestBackward<WaveTeam> back(fd, A, B);
  for (long long item = blockIdx.x; item < batch; item += gridDim.x) {
    if (u) prog.load_actuation(u + item * ld_u);
    back.run(x + item * ld_x, gx + item * ld_g, theta, friction, lengths, opt, row, STATE ? xbar_out + item * ld_xb : (T*)nullptr);
    __syncthreads();
  }
  __syncthreads();
  for (int e = threadIdx.x; e < width; e += kWave) partials[(long long)blockIdx.x * row_stride + e] = row[e];
}

// ---- MultibodyTerms.forward (multibody_terms.py:584-609): D, M, J, phi, a --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kWave) void forest_terms_kernel(const ForestDesc* __restrict__ fdp, const T* __restrict__ theta,
                                                             const T* __restrict__ friction, const T* __restrict__ lengths,
                                                             const T* __restrict__ x, long long ld_x, long long batch, T* __restrict__ Dout,
                                                             T* __restrict__ Mout, T* __restrict__ Jout, T* __restrict__ phiout,
                                                             T* __restrict__ aout, const T* __restrict__ u, long long ld_u) {
  __shared__ ForestDesc fd;
  fetch_desc(fdp, fd);
  Arena<T, double> A;
  A.carve(forest_smem, fd.n_bodies, fd.n_v, fd.n_q, fd.n_contacts, fd.n_geoms, fd.n_pairs);
  Forest<T, double, WaveTeam> prog(fd, A);
  prog.derive(theta, friction, lengths);
  prog.load_actuation((const T*)nullptr);  // (no inputs: B u = 0 until an item brings its own)
  const int nv = fd.n_v, K = fd.n_contacts;
  for (long long item = blockIdx.x; item < batch; item += gridDim.x) {
    if (u) prog.load_actuation(u + item * ld_u);
    prog.load_state(x + item * ld_x);
    prog.terms();
    prog.contacts(lengths);
    if (Mout)
      for (int e = threadIdx.x; e < nv * nv; e += kWave) Mout[item * nv * nv + e] = A.M[e];
    if (aout)
      for (int e = threadIdx.x; e < nv; e += kWave) aout[item * nv + e] = A.a[e];
    if (phiout)
      for (int c = threadIdx.x; c < K; c += kWave) phiout[item * K + c] = A.ct[c].phi;
    // rows of J in the reference order [normals | mu (t_x, t_y) per contact]: row r of contact c, scaled
    auto jrow = [&](int r, int i) -> T {
      const int c = r < K ? r : (r - K) / 2, comp = r < K ? 2 : (r - K) % 2;
      const T scale = r < K ? T(1) : A.ct[c].mu;
      return scale * A.J[((size_t)c * 3 + comp) * nv + i];
    };
    if (Jout)
      for (int e = threadIdx.x; e < 3 * K * nv; e += kWave) Jout[item * 3 * K * nv + e] = jrow(e / nv, e % nv);
    if (Dout) {
      // D = J M^-1 J^T = Z Z^T with Z = (L^-1 J^T)^T: one forward substitution per row of J (a lane each), into the CJ block
      T* Z = A.CJ;
      for (int r = threadIdx.x; r < 3 * K; r += kWave)
        for (int i = 0; i < nv; ++i) {
          T s = jrow(r, i);
          for (int p = 0; p < i; ++p) s -= A.LM[i * nv + p] * Z[(size_t)r * nv + p];
          Z[(size_t)r * nv + i] = s * A.invdM[i];
        }
      __syncthreads();
      for (int e = threadIdx.x; e < 9 * K * K; e += kWave) {
        const int r = e / (3 * K), c = e % (3 * K);
        T s = T(0);
        for (int i = 0; i < nv; ++i) s += Z[(size_t)r * nv + i] * Z[(size_t)c * nv + i];
        Dout[item * 9 * K * K + e] = s;
      }
    }
    __syncthreads();
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
const ForestDesc& host_desc(const dpll_model* m) { return *m->forest; }

// The description's device copy for the CURRENT device, made at the handle's first launch there: under a mutex (two
// first calls used to race and leak one allocation), keyed by hipGetDevice() (a handle used with another GPU's stream used to
// dereference the first GPU's memory), and refused with a clear error while `stream` is being captured (the allocation and the
// synchronous copy would invalidate the capture: warm up with one eager call first, as include/dpll.h says).
const ForestDesc* device_desc(const dpll_model* m, hipStream_t stream) {
  static std::mutex guard;
  std::lock_guard<std::mutex> lock(guard);
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= dpll_model::kMaxDevices) return nullptr;
  dpll_model* mm = const_cast<dpll_model*>(m);
  if (!mm->forest_dev[device]) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &status) == hipSuccess && status != hipStreamCaptureStatusNone) {
      (void)dpll_fail(-5, "%s: the forest description is not on this device yet and the stream is being captured: make one eager call first", "dpll (forest build)");
      return nullptr;
    }
    void* dev = nullptr;
    if (hipMalloc(&dev, sizeof(ForestDesc)) != hipSuccess) return nullptr;
    if (hipMemcpy(dev, m->forest, sizeof(ForestDesc), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(dev);
      return nullptr;
    }
    mm->forest_dev[device] = dev;
  }
  return static_cast<const ForestDesc*>(mm->forest_dev[device]);
}

int stride_of(const ForestDesc& fd) { return (row_width(fd) + 1) & ~1; }
// Lanes per item of the loss and rollout launches: four items per wave (16 lanes each) for the smallest systems (cube- and
// elbow-sized: 4096 cube pairs 261 -> 161 us per loss + gradients); else the whole wave on one item -- four items per wave pay for
// the slowest of the four at every iteration and take four passes where 64 lanes take one (two cubes 580 -> 860 us, gripper
// 482 -> 598 us: measured)
// Half a wave per item (GroupTeam<32>, built with -DDPLL_FOREST_HALF=1 only): measured in round 4 after the factorisations moved
// to registers and NOT adopted -- 4096 items, f32 loss + gradients, one item per wave vs two: two_cubes 365 -> 430 us, gripper
// 350 -> 395, pendulum + cube 385 -> 403, rake 442 -> 429: the phases wider than 32 lanes (271 direction candidates, 78
// Hessian entries) take their extra rounds and the wave pays for the slower of its two items at every iteration
#ifndef DPLL_FOREST_HALF
#define DPLL_FOREST_HALF 0
#endif
template <typename T> int lanes_per_item(const ForestDesc& fd) {
  const size_t arena = arena_bytes<T, double>(fd);
  if (4 * arena <= 96 * 1024 && fd.n_contacts <= 8 && fd.n_v <= 8) return 16;
  if (DPLL_FOREST_HALF && 2 * arena <= 64 * 1024 && fd.n_contacts <= 16 && fd.n_v <= 16 && fd.n_pairs <= 2) return 32;
  return kWave;
}
int grid_for(long long batch, size_t lds_bytes, int items_per_wave = 1) {
  batch = (batch + items_per_wave - 1) / items_per_wave;
  // workgroups that can be resident at once (160 KB of LDS per CU, DPLL_FOREST_OCC one-wave workgroups per SIMD), capped at the batch
  long long per_cu = (long long)((160 * 1024) / (lds_bytes + sizeof(ForestDesc) + 256));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4 * DPLL_FOREST_OCC) per_cu = 4 * DPLL_FOREST_OCC;
  long long blocks = 256 * per_cu;
  if (blocks > kForestMaxRows) blocks = kForestMaxRows;
  if (blocks > batch) blocks = batch;
  return (int)(blocks < 1 ? 1 : blocks);
}
long long folded_rows(long long rows) { return (rows + kFold - 1) / kFold; }
// rows a gradient launch may write, whatever the batch, dtype and lanes per item: what the workspace is laid out for
// ([rows (max_rows) | folded rows])
int max_rows(const ForestDesc&) { return kForestMaxRows; }

template <typename K> int allow_lds(K kernel, size_t bytes, const char* who) {
  if (bytes + sizeof(ForestDesc) + 4096 > 160 * 1024) return dpll_fail(-2, "%s: the model needs more LDS per item than a CU has", who);
  // (once per kernel, device and size: the attribute call used to be made at every launch)
  static thread_local size_t allowed[dpll_model::kMaxDevices] = {};
  int device = 0;
  (void)hipGetDevice(&device);
  if (device >= 0 && device < dpll_model::kMaxDevices && bytes <= allowed[device]) return 0;
  if (device >= 0 && device < dpll_model::kMaxDevices) allowed[device] = bytes > 48 * 1024 ? bytes : allowed[device];
  if (bytes > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return dpll_fail(-5, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", who);
  return 0;
}
size_t round16(size_t n) { return (n + 15) & ~(size_t)15; }

template <typename T>
int launch_loss(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp, long long ld_xp,
                long long batch, const void* weights, double scale, void* loss, void* grad, void* loss_total, void* force, int32_t* iters,
                void* workspace, long long ws_bytes, hipStream_t stream, const AdamArgs* adam) {
  const ForestDesc& fd = host_desc(m);
  const ForestDesc* dev = device_desc(m, stream);
  if (!dev) return dpll_fail(-5, "dpll_contactnets_loss (forest build): could not place the model description on the device (a first call inside a stream capture? make one eager call before capturing)%s");
  const int want_grad = grad != nullptr;
  if (want_grad) {
    if (!workspace || ws_bytes < dpll_forest_api::workspace_bytes(m, batch)) return dpll_fail(-3, "dpll_contactnets_loss: workspace too small%s");
  } else if (loss_total) {
    return dpll_fail(-3, "dpll_contactnets_loss: loss_total requires grad%s");
  }
  const int lanes = lanes_per_item<T>(fd), teams = kWave / lanes;
  const size_t arena = arena_bytes<T, double>(fd);
  const size_t lds = teams * (round16((size_t)row_width(fd) * sizeof(double)) + arena);
  if (lanes == 16) {
    if (int rc = allow_lds(forest_loss_kernel<T, 16>, lds, "dpll_contactnets_loss")) return rc;
#if DPLL_FOREST_HALF
  } else if (lanes == 32) {
    if (int rc = allow_lds(forest_loss_kernel<T, 32>, lds, "dpll_contactnets_loss")) return rc;
#endif
  } else {
    if (int rc = allow_lds(forest_loss_kernel<T, kWave>, lds, "dpll_contactnets_loss")) return rc;
  }
  const int rows = batch > 0 ? grid_for(batch, lds, teams) : 0;
  const int stride = stride_of(fd);
  if (rows > 0) {
#define DPLL_FOREST_LOSS(G_)                                                                                                                    \
    hipLaunchKernelGGL((forest_loss_kernel<T, G_>), dim3(rows), dim3(kWave), lds, stream, dev, m->opts[dtype], (const T*)p->theta,                \
                       (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)xp, ld_xp, batch, (const T*)weights, scale,   \
                       (T*)loss, (T*)force, (int*)iters, (double*)workspace, want_grad, stride, (unsigned)arena, (const T*)p->u, (long long)p->ld_u)
    if (lanes == 16) DPLL_FOREST_LOSS(16);
#if DPLL_FOREST_HALF
    else if (lanes == 32) DPLL_FOREST_LOSS(32);
#endif
    else DPLL_FOREST_LOSS(kWave);
#undef DPLL_FOREST_LOSS
    if (int rc = dpll_check_launch("forest_loss_kernel")) return rc;
  }
  if (!want_grad) return 0;
  double* folded = (double*)workspace + (long long)max_rows(fd) * stride;
  const int n_folded = (int)folded_rows(rows);
  if (n_folded > 0) {
    hipLaunchKernelGGL(forest_fold_kernel, dim3(n_folded), dim3(kRowThreads), 0, stream, (const double*)workspace, rows, row_width(fd), stride, folded);
    if (int rc = dpll_check_launch("forest_fold_kernel")) return rc;
  }
  hipLaunchKernelGGL((forest_finalize_kernel<T>), dim3(1), dim3(kRowThreads), 0, stream, dev, (const T*)p->theta, (const T*)p->friction,
                     (const T*)p->lengths, (const double*)folded, n_folded, stride, (T*)grad, (T*)loss_total, adam ? *adam : AdamArgs{});
  return dpll_check_launch("forest_finalize_kernel");
}

template <typename T>
int launch_simulate(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x0, long long ld_x, long long batch, long long steps,
                    void* out, long long ld_item, long long ld_step, int write_x0, int32_t* iters, hipStream_t stream) {
  const ForestDesc& fd = host_desc(m);
  const ForestDesc* dev = device_desc(m, stream);
  if (!dev) return dpll_fail(-5, "dpll_simulate (forest build): could not place the model description on the device (a first call inside a stream capture? make one eager call before capturing)%s");
  const int lanes = lanes_per_item<T>(fd), teams = kWave / lanes;
  const size_t arena = arena_bytes<T, double>(fd);
  const size_t lds = teams * (round16((size_t)2 * (fd.n_q + fd.n_v) * sizeof(T)) + arena);
  // (actuation inputs belong to ONE step: dpll_step; a rollout runs unactuated, as the reference's sim_step passes a u of width 0)
  const T* sim_u = steps == 1 ? (const T*)p->u : (const T*)nullptr;
#define DPLL_FOREST_SIM(G_)                                                                                                                     \
  do {                                                                                                                                          \
    if (int rc = allow_lds(forest_simulate_kernel<T, G_>, lds, "dpll_simulate")) return rc;                                                      \
    hipLaunchKernelGGL((forest_simulate_kernel<T, G_>), dim3(grid_for(batch, lds, teams)), dim3(kWave), lds, stream, dev, m->opts[dtype],         \
                       (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x, batch, steps, (T*)out, ld_item,      \
                       ld_step, write_x0, (int*)iters, (unsigned)arena, sim_u, (long long)p->ld_u);                    \
  } while (0)
  if (lanes == 16) DPLL_FOREST_SIM(16);
#if DPLL_FOREST_HALF
  else if (lanes == 32) DPLL_FOREST_SIM(32);
#endif
  else DPLL_FOREST_SIM(kWave);
#undef DPLL_FOREST_SIM
  return dpll_check_launch("forest_simulate_kernel");
}

template <typename T>
int launch_step_backward(const dpll_model* m, const dpll_params_t* p, const void* x, long long ld_x, const void* gx, long long ld_g, long long batch,
                         void* grad, void* grad_x, long long ld_gx, void* workspace, long long ws_bytes, hipStream_t stream) {
  const ForestDesc& fd = host_desc(m);
  const ForestDesc* dev = device_desc(m, stream);
  if (!dev) return dpll_fail(-5, "dpll_step_backward (forest build): could not place the model description on the device (a first call inside a stream capture? make one eager call before capturing)%s");
  if (!workspace || ws_bytes < dpll_forest_api::workspace_bytes(m, batch)) return dpll_fail(-3, "dpll_step_backward: workspace too small%s");
  const size_t head = round16((size_t)row_width(fd) * sizeof(double));
  const size_t lds = head + arena_bytes<double, double>(fd) + (grad_x ? arena_bytes<DualT<double>, DualT<double>>(fd, true) : 0);
  if (grad_x) {
    if (int rc = allow_lds(forest_step_backward_kernel<T, true>, lds, "dpll_step_backward")) return rc;
  } else {
    if (int rc = allow_lds(forest_step_backward_kernel<T, false>, lds, "dpll_step_backward")) return rc;
  }
  // (rows: never more than the workspace is sized for)
  int rows = grid_for(batch, lds);
  const int cap = max_rows(fd);
  if (rows > cap) rows = cap;
  const int stride = stride_of(fd);
  if (grad_x)
    hipLaunchKernelGGL((forest_step_backward_kernel<T, true>), dim3(rows), dim3(kWave), lds, stream, dev, m->opts[DPLL_F64], (const T*)p->theta,
                       (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)gx, ld_g, batch, (double*)workspace, stride,
                       (T*)grad_x, ld_gx, (const T*)p->u, (long long)p->ld_u);
  else
    hipLaunchKernelGGL((forest_step_backward_kernel<T, false>), dim3(rows), dim3(kWave), lds, stream, dev, m->opts[DPLL_F64], (const T*)p->theta,
                       (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)gx, ld_g, batch, (double*)workspace, stride,
                       (T*)nullptr, 0LL, (const T*)p->u, (long long)p->ld_u);
  if (int rc = dpll_check_launch("forest_step_backward_kernel")) return rc;
  double* folded = (double*)workspace + (long long)cap * stride;
  const int n_folded = (int)folded_rows(rows);
  hipLaunchKernelGGL(forest_fold_kernel, dim3(n_folded), dim3(kRowThreads), 0, stream, (const double*)workspace, rows, row_width(fd), stride, folded);
  if (int rc = dpll_check_launch("forest_fold_kernel")) return rc;
  hipLaunchKernelGGL((forest_finalize_kernel<T>), dim3(1), dim3(kRowThreads), 0, stream, dev, (const T*)p->theta, (const T*)p->friction,
                     (const T*)p->lengths, (const double*)folded, n_folded, stride, (T*)grad, (T*)nullptr, AdamArgs{});
  return dpll_check_launch("forest_finalize_kernel");
}

template <typename T>
int launch_terms(const dpll_model* m, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm, void* M, void* J, void* phi,
                 void* a, hipStream_t stream) {
  const ForestDesc& fd = host_desc(m);
  const ForestDesc* dev = device_desc(m, stream);
  if (!dev) return dpll_fail(-5, "dpll_terms (forest build): could not place the model description on the device (a first call inside a stream capture? make one eager call before capturing)%s");
  const size_t lds = arena_bytes<T, double>(fd);
  if (int rc = allow_lds(forest_terms_kernel<T>, lds, "dpll_terms")) return rc;
  hipLaunchKernelGGL((forest_terms_kernel<T>), dim3(grid_for(batch, lds)), dim3(kWave), lds, stream, dev, (const T*)p->theta, (const T*)p->friction,
                     (const T*)p->lengths, (const T*)x, ld_x, batch, (T*)Dm, (T*)M, (T*)J, (T*)phi, (T*)a, (const T*)p->u, (long long)p->ld_u);
  return dpll_check_launch("forest_terms_kernel");
}

}  // namespace

#ifdef DPLL_FOREST_STAMPS
extern "C" int dpll_debug_forest_stamps(unsigned long long* host_out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fstamps), sizeof(unsigned long long) * 64);
  if (reset) {
    unsigned long long zero[64] = {};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fstamps), zero, sizeof(zero));
  }
  return rc;
}
#endif

namespace dpll_forest_api {

int check_desc(const dpll_forest_desc_t* d) {
  if (d->n_bodies < 1 || d->n_bodies > dpll_forest::kMaxBodies) return dpll_fail(-2, "dpll_forest_model_create: 1 to 16 bodies%s");
  if (d->n_geoms < 1 || d->n_geoms > dpll_forest::kMaxGeoms) return dpll_fail(-2, "dpll_forest_model_create: 1 to 12 collision geometries%s");
  if (d->n_pairs < 0 || d->n_pairs > dpll_forest::kMaxPairs) return dpll_fail(-2, "dpll_forest_model_create: at most 16 body-body collision candidates%s");
  if (d->n_contacts < 1 || d->n_contacts > kMaxContacts) return dpll_fail(-2, "dpll_forest_model_create: 1 to 64 contacts%s");
  if (d->n_v < 1 || d->n_v > kMaxV) return dpll_fail(-2, "dpll_forest_model_create: 1 to 32 generalized velocities%s");
  if (!(d->dt > 0.0)) return dpll_fail(-1, "dpll_forest_model_create: dt must be positive%s");
  if (d->inertia_mode != DPLL_INERTIA_REFERENCE_LITERAL && d->inertia_mode != DPLL_INERTIA_PHYSICAL &&
      d->inertia_mode != DPLL_INERTIA_COMPOSED)
    return dpll_fail(-1, "dpll_forest_model_create: unknown inertia_mode%s");
  if (d->rotated & ~3) return dpll_fail(-1, "dpll_forest_model_create: rotated holds bits 0 and 1 only%s");
  if (d->n_u < 0 || d->n_u > dpll_forest::kMaxV) return dpll_fail(-1, "dpll_forest_model_create: n_u must be between 0 and 32%s");
  int n_q = 0, n_v = 0, max_depth = 0;
  for (int b = 0; b < d->n_bodies; ++b) {
    const int kind = d->joint_kind[b], parent = d->parent[b];
    if (kind != kJointRevolute && kind != kJointPrismatic && kind != kJointFloating && kind != kJointFixed)
      return dpll_fail(-1, "dpll_forest_model_create: unknown joint kind%s");
    const bool root = kind == kJointFloating || kind == kJointFixed;
    if (root ? parent != -1 : (parent < 0 || parent >= b)) return dpll_fail(-1, "dpll_forest_model_create: a root has parent -1, any other body a parent listed before it%s");
    if (d->depth[b] != (root ? 0 : d->depth[parent] + 1)) return dpll_fail(-1, "dpll_forest_model_create: depth[] must be the distance to the root%s");
    max_depth = d->depth[b] > max_depth ? d->depth[b] : max_depth;
    const int nq = kind == kJointFloating ? 7 : (kind == kJointFixed ? 0 : 1), nv = dofs_of(kind);
    if (d->q_index[b] < 0 || d->q_index[b] + nq > d->n_q || d->v_index[b] < 0 || d->v_index[b] + nv > d->n_v)
      return dpll_fail(-1, "dpll_forest_model_create: q_index / v_index out of range%s");
    if (!root && d->v_index[b] < d->v_index[parent] + dofs_of(d->joint_kind[parent]))
      return dpll_fail(-1, "dpll_forest_model_create: a body's velocities come after its parent's%s");
    for (int i = 0; i < nv; ++i)
      if (d->dof_body[d->v_index[b] + i] != b) return dpll_fail(-1, "dpll_forest_model_create: dof_body does not match v_index%s");
    n_q += nq;
    n_v += nv;
  }
  if (n_q != d->n_q || n_v != d->n_v || max_depth != d->max_depth) return dpll_fail(-1, "dpll_forest_model_create: n_q / n_v / max_depth do not match the bodies%s");
  // actuators: each on a revolute or prismatic joint, no joint twice
  for (int k = 0; k < d->n_u; ++k) {
    const int b = d->act_body[k];
    if (b < 0 || b >= d->n_bodies || (d->joint_kind[b] != kJointRevolute && d->joint_kind[b] != kJointPrismatic))
      return dpll_fail(-1, "dpll_forest_model_create: act_body must name bodies on revolute or prismatic joints%s");
    for (int j = 0; j < k; ++j)
      if (d->act_body[j] == b) return dpll_fail(-1, "dpll_forest_model_create: two actuators on one joint%s");
  }
  //  what a C caller could get silently wrong: the `rotated` bits against the matrices they announce, hinge axes that
  // are not unit vectors, coordinate ranges of two bodies that overlap
  auto is_identity = [](const double (&r)[3][3]) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        if (fabs(r[i][j] - (i == j ? 1.0 : 0.0)) > 1e-12) return false;
    return true;
  };
  bool body_turned = false, geom_turned = false;
  for (int b = 0; b < d->n_bodies; ++b) body_turned = body_turned || !is_identity(d->body_rot[b]);
  for (int g = 0; g < d->n_geoms && g < dpll_forest::kMaxGeoms; ++g) geom_turned = geom_turned || !is_identity(d->geom_rot[g]);
  if (body_turned != ((d->rotated & 1) != 0) || geom_turned != ((d->rotated & 2) != 0))
    return dpll_fail(-1, "dpll_forest_model_create: rotated must say which of body_rot / geom_rot hold a rotation other than the identity%s");
  for (int b = 0; b < d->n_bodies; ++b) {
    const int kind = d->joint_kind[b];
    if (kind == kJointRevolute || kind == kJointPrismatic) {
      const double n2 = d->joint_axis[b][0] * d->joint_axis[b][0] + d->joint_axis[b][1] * d->joint_axis[b][1] + d->joint_axis[b][2] * d->joint_axis[b][2];
      if (fabs(n2 - 1.0) > 1e-9) return dpll_fail(-1, "dpll_forest_model_create: joint_axis must be a unit vector%s");
    }
    const int nq_b = kind == kJointFloating ? 7 : (kind == kJointFixed ? 0 : 1);
    for (int c = 0; c < b; ++c) {
      const int kc = d->joint_kind[c], nq_c = kc == kJointFloating ? 7 : (kc == kJointFixed ? 0 : 1);
      if (nq_b > 0 && nq_c > 0 && d->q_index[b] < d->q_index[c] + nq_c && d->q_index[c] < d->q_index[b] + nq_b)
        return dpll_fail(-1, "dpll_forest_model_create: the coordinate ranges of two bodies overlap%s");
    }
  }
  int contacts = 0;
  for (int g = 0; g < d->n_geoms; ++g) {
    if (d->geom_body[g] < 0 || d->geom_body[g] >= d->n_bodies) return dpll_fail(-1, "dpll_forest_model_create: geom_body out of range%s");
    const int kind = d->geom_kind[g];
    if (kind != DPLL_GEOM_BOX && kind != DPLL_GEOM_SPHERE && kind != DPLL_GEOM_POLYGON)
      return dpll_fail(-2, "dpll_forest_model_create: boxes, spheres and polygons (learned shapes run on the general build)%s");
    if (kind == DPLL_GEOM_POLYGON && (d->geom_nverts[g] < 4 || d->geom_nverts[g] > DPLL_MAX_POLYGON_VERTICES))
      return dpll_fail(-2, "dpll_forest_model_create: a polygon has 4 to 8 vertices%s");
    // a geometry on a body welded to the world (the root of a fixed-base model) is ANCHORED: it does not meet the ground, which
    // sits on the world body too (Drake filters anchored-anchored candidates, drake_utils.py:178-184) -- it lists no witnesses
    const bool anchored = d->joint_kind[d->geom_body[g]] == kJointFixed;
    for (int s = 0; s < (anchored ? 0 : (kind == DPLL_GEOM_SPHERE ? 1 : 4)); ++s, ++contacts)
      if (contacts >= d->n_contacts || d->contact_geom[contacts] != g || d->contact_slot[contacts] != s)
        return dpll_fail(-1, "dpll_forest_model_create: contacts must list the witnesses of every geometry that can move, in order, then the candidates%s");
  }
  for (int p = 0; p < d->n_pairs; ++p, ++contacts) {
    const int a = d->pair_a[p], b = d->pair_b[p];
    if (a < 0 || b < 0 || a >= d->n_geoms || b >= d->n_geoms || a == b || d->geom_body[a] == d->geom_body[b])
      return dpll_fail(-1, "dpll_forest_model_create: a candidate joins geometries of two different bodies%s");
    if (d->joint_kind[d->geom_body[a]] == kJointFixed && d->joint_kind[d->geom_body[b]] == kJointFixed)
      return dpll_fail(-1, "dpll_forest_model_create: a candidate between two geometries welded to the world%s");
    for (int o = 0; o < p; ++o)
      if ((d->pair_a[o] == a && d->pair_b[o] == b) || (d->pair_a[o] == b && d->pair_b[o] == a))
        return dpll_fail(-1, "dpll_forest_model_create: a candidate listed twice%s");
    if (contacts >= d->n_contacts || d->contact_geom[contacts] >= 0 || d->contact_slot[contacts] != p)
      return dpll_fail(-1, "dpll_forest_model_create: contacts must list every geometry's witnesses in order, then the candidates%s");
  }
  if (contacts != d->n_contacts) return dpll_fail(-1, "dpll_forest_model_create: n_contacts does not match the geometries and candidates%s");
  return 0;
}

int n_x(const dpll_model* m) { return m->forest->n_q + m->forest->n_v; }
int n_contacts(const dpll_model* m) { return m->forest->n_contacts; }
int n_u(const dpll_model* m) { return host_desc(m).n_u; }
int param_count(const dpll_model* m) { return dpll_forest::param_count(*m->forest); }
long long workspace_bytes(const dpll_model* m, long long batch) {
  const ForestDesc& fd = *m->forest;
  (void)batch;
  const long long rows = max_rows(fd);
  return (rows + folded_rows(rows)) * stride_of(fd) * (long long)sizeof(double);
}
void release(dpll_model* m) {
  for (int d = 0; d < dpll_model::kMaxDevices; ++d) {
    if (m->forest_dev[d]) (void)hipFree(m->forest_dev[d]);  // (hipFree finds the owning device from the pointer)
    m->forest_dev[d] = nullptr;
  }
  delete m->forest;
  m->forest = nullptr;
}

#define DPLL_FOREST_DISPATCH(FN, ...)                          \
  do {                                                         \
    if (dtype == DPLL_F32) return FN<float>(__VA_ARGS__);      \
    return FN<double>(__VA_ARGS__);                            \
  } while (0)

int loss(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp, long long ld_xp, long long batch,
         const void* weights, double scale, void* loss_out, void* grad, void* loss_total, void* force, int32_t* iters, void* workspace,
         long long ws_bytes, hipStream_t stream, const AdamArgs* adam) {
  DPLL_FOREST_DISPATCH(launch_loss, m, dtype, p, x, ld_x, xp, ld_xp, batch, weights, scale, loss_out, grad, loss_total, force, iters, workspace,
                       ws_bytes, stream, adam);
}
int simulate(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x0, long long ld_x, long long batch, long long steps, void* out,
             long long ld_item, long long ld_step, int write_x0, int32_t* iters, hipStream_t stream) {
  DPLL_FOREST_DISPATCH(launch_simulate, m, dtype, p, x0, ld_x, batch, steps, out, ld_item, ld_step, write_x0, iters, stream);
}
int step_backward(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* gx, long long ld_g,
                  long long batch, void* grad, void* grad_x, long long ld_gx, void* workspace, long long ws_bytes, hipStream_t stream) {
  DPLL_FOREST_DISPATCH(launch_step_backward, m, p, x, ld_x, gx, ld_g, batch, grad, grad_x, ld_gx, workspace, ws_bytes, stream);
}
int terms(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm, void* M, void* J,
          void* phi, void* a, hipStream_t stream) {
  DPLL_FOREST_DISPATCH(launch_terms, m, p, x, ld_x, batch, Dm, M, J, phi, a, stream);
}

}  // namespace dpll_forest_api
