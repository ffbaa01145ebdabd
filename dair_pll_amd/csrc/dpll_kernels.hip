// dpll_kernels.hip -- gfx950 kernels and the C ABI (include/dpll.h) of the contact-dynamics hot path.
//
// Mapping: ONE LANE PER CONTACT.  A batch item (one (x, x+) transition or one trajectory) is owned by a
// group of G = 4 n_bodies adjacent lanes of a wavefront (cube: 4 lanes, 16 items per wave; elbow: 8 lanes,
// 8 items per wave).  Each lane keeps its contact's witness point, 3 x n_v Jacobian rows and cone state in
// registers; the n_v x n_v blocks (M, Newton Hessian, Cholesky factors) are replicated over the group and
// sums over contacts are DPP butterflies (quad_perm / row_half_mirror), so the inner solver loop touches
// neither LDS nor memory.  One wave per workgroup: at the benchmark batch (4096 items = 256 waves) every
// wave gets a CU of its own, which is what a latency-bound Newton iteration wants; bigger batches loop
// items inside the wave (grid capped) and fill the chip with more waves per SIMD.
//
// Gradients never leave the chip per item: each wave reduces its items' d/d(iota, mu, |length|) with
// cross-lane adds, writes one row of double partial sums, and a one-block finalize kernel sums the rows in
// a fixed order (bitwise reproducible, no float atomics) and chains them to (theta, friction, lengths).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <cstdlib>
#include <type_traits>

#include "../../include/dpll.h"

// Diagnostic build only (-DDPLL_STAMPS, never shipped): shader-clock stamps per wave at phase boundaries of
// the loss kernel, written to a buffer of their own that no kernel reads.
#ifdef DPLL_STAMPS
__device__ unsigned long long g_stamps[2048][8];
#define DPLL_STAMP(slot)                                                                                   \
  do {                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    unsigned long long t_;                                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (threadIdx.x == 0) g_stamps[blockIdx.x][slot] = t_;                                                 \
  } while (0)
#else
#define DPLL_STAMP(slot) do {} while (0)
#endif
#ifdef DPLL_STAMPS
#define DPLL_CORE_STAMP(slot) DPLL_STAMP(slot)
// cycles per phase of the Newton iteration, summed over the iterations of a wave (rows 1024.. of g_stamps)
__device__ __forceinline__ unsigned long long dpll_clock_() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t_;
}
#define DPLL_PHASE_BEGIN() unsigned long long ph_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_dur_ = 0; unsigned long long ph_last_ = dpll_clock_()
#define DPLL_PHASE(slot) do { const unsigned long long t_ = dpll_clock_(); ph_dur_ = t_ - ph_last_; ph_acc_[slot] += ph_dur_; ph_last_ = t_; } while (0)
#define DPLL_PHASE_COUNT(slot) do { ph_acc_[slot] += 1; } while (0)
// one fallback-search event of this wave: slot 7 = count | duration of the first event << 8 | shortest event << 32
#define DPLL_PHASE_EVENT(slot, happened)                                                                   \
  do {                                                                                                     \
    if (happened) {                                                                                        \
      const unsigned long long d_ = ph_dur_ & 0xffffffull, n_ = ph_acc_[7] & 0xffull;                      \
      const unsigned long long f_ = n_ ? ((ph_acc_[7] >> 8) & 0xffffffull) : d_;                           \
      const unsigned long long m0_ = (ph_acc_[7] >> 32) & 0xffffffull, m_ = (n_ && m0_ < d_) ? m0_ : d_;   \
      ph_acc_[7] = (n_ + 1) | (f_ << 8) | (m_ << 32);                                                      \
    }                                                                                                      \
  } while (0)
#define DPLL_PHASE_END()                                                                                   \
  do {                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 1024)                                                             \
      for (int k_ = 0; k_ < 8; ++k_) g_stamps[1024 + blockIdx.x][k_] = ph_acc_[k_];                        \
  } while (0)
#endif
#include "dpll_core.hpp"
#include "dpll_allreduce.hpp"

#include "dpll_common.hpp"
#include "dpll_general.hpp"
#include "dpll_genmesh.hpp"
#include "dpll_forest.hpp"
#include "dpll_forest_api.hpp"
#include "dpll_gjk.hpp"
#include "dpll_mesh_kernels.hpp"
#include "dpll_mesh_bf16.hpp"
#include "dpll_icnn_pipe_api.hpp"
namespace {

// ---- ContactNets loss, forward + backward -----------------------------------------------------
// MESH: the contact's support point is read from `witness` (ICNN kernels) and its adjoint written to `rbar_out`;
// a template flag so that the box kernels carry none of it (a run-time choice of array put both in scratch).
// DENSE: built for two waves per SIMD (<= 256 registers, a few spills) -- for launches with more waves than SIMDs, where
// a second resident wave fills the issue slots a lone wave leaves empty (65,536 pairs: 800 -> 944 M steps/s); the
// headline launch (256 waves) keeps the roomier one-wave build (it is 1 % faster there).
constexpr int kRaceWaves = 4;  // waves per workgroup of the racing build
// A racing launch has a wave for EVERY SIMD, and the dispatcher does not spread them: a float kernel of <= 256 registers fits
// twice on a SIMD, waves get doubled up while other SIMDs idle, and a doubled wave runs at ~2/3 speed (measured with the
// rollout kernel, 188 VGPRs: 256 / 512 / 1024 / 2048 one-wave workgroups of identical work take 13.1 / 13.4 / 20.1 / 20.2 us
// per step; the double build, 334 VGPRs, 19.4 / 19.8 / 19.7 / 37.8; an LDS reservation that caps a CU at four workgroups
// does not help: the doubling is inside the CU).  Touching accumulation register a95 makes the wave's allocation (vector +
// accumulation registers share one file of 512) exceed half the file for every float kernel here (>= 183 vector registers),
// so a SIMD holds exactly one such wave; it leaves 416 vector registers to the kernel.  The double kernels allocate more than
// half the file by themselves and do not claim (a claim of 128 accumulation registers cost the double elbow kernel, 512
// vector registers, 6 %: 48.7 -> 51.6 us).
template <typename T> __device__ __forceinline__ void claim_whole_simd() {
  if constexpr (sizeof(T) == 4) asm volatile("v_accvgpr_write_b32 a95, 0" ::: "a95");
}
// RACE: copies of every item's lane group that run other continuation schedules of the cone solve in lock step
// (SolverOpts::portfolio, dpll_core.hpp sap_newton): for launches that would leave SIMDs idle.  An item owns G * RACE
// lanes; the copy that converged first supplies the item's loss, forces, iteration count and gradient terms.
// The racing build runs workgroups of four waves (one per SIMD of a CU) that share one partial row, so that the finalize
// kernel sums as many rows as without the copies.
// WAVES: waves per workgroup; the waves of a workgroup share ONE partial row (the racing build always; the plain builds from
// 512 waves per launch, so that the finalize kernel never sums more than 512 rows)
// KPL: contacts per lane (1; 2 for the elbow's four-copy racing build: 8 contacts on 4 lanes leave its 16-lane row room for
// four copies)
template <typename T, int NJ, bool MESH = false, bool DENSE = false, int RACE = 1, int WAVES = (RACE > 1 ? kRaceWaves : 1), int KPL = 1>
__global__ __launch_bounds__(WAVES * kWave, DENSE ? 2 : 1) void loss_kernel(const T* __restrict__ x, long long ld_x,
                                                     const T* __restrict__ xp, long long ld_xp, long long batch,
                                                     const T* __restrict__ theta, const T* __restrict__ friction,
                                                     const T* __restrict__ lengths,
                                                     const T* __restrict__ weights, double scale, T* __restrict__ loss,
                                                     T* __restrict__ force, int* __restrict__ iters,
                                                     double* __restrict__ partials, int want_grad,
                                                     const T* __restrict__ witness, T* __restrict__ rbar_out,
                                                     ModelDesc md, SolverOpts opt) {
  // (argument order: what the first loads of a wave need -- the state pointers, strides and the batch size -- comes first so that
  // the kernel-argument preload of gfx950 (-amdgpu-kernarg-preload-count, csrc/Makefile) hands it over in SGPRs at wave start:
  // the state rows are requested without waiting for a fetch of the argument segment; the two structs are fetched behind them)
  using D = Dims<T, NJ>;
  constexpr int G = D::K / KPL;  // lanes of one copy of an item
  using Lanes = GpuLanes<G, RACE>;
  static_assert(RACE == 1 || !MESH || KPL == 1, "racing copies with learned shapes: one contact per lane");
  static_assert(KPL == 1 || !MESH, "several contacts per lane: box geometry");
  constexpr int kItems = WAVES * (kWave / G) / RACE;  // items per workgroup
  if constexpr (!DENSE) claim_whole_simd<T>();  // (this build serves the launches of at most one wave per SIMD: the launchers
                                                // send anything bigger to the DENSE build -- box and mesh geometry alike)
  const int lane = threadIdx.x;
  const int cidx = lane % G;
  const int slot = lane / (G * RACE);
  const int item_blocks = (int)gridDim.x - 1;  // the last workgroup owns no items: it writes the chain matrix
  if ((int)blockIdx.x == item_blocks) {
    if (want_grad && lane < kWave) write_chain_matrix<T, T, D::NB>(md.inertia_mode & 1, theta, friction, lengths, partials + (long long)item_blocks * D::PI);
    return;
  }
  DPLL_STAMP(0);
  // the first item's state rows are requested before the parameter math so that their memory latency hides behind it
  const long long stride = (long long)item_blocks * kItems;
  long long base = (long long)blockIdx.x * kItems;
  long long item = base + slot;
  bool valid = item < batch;
  long long it = valid ? item : batch - 1;  // idle groups shadow the last item: keeps every lane live for DPP
  T xr[D::NX], xpr[D::NX];
#pragma unroll
  for (int i = 0; i < D::NX; ++i) { xr[i] = x[it * ld_x + i]; xpr[i] = xp[it * ld_xp + i]; }
  Derived<T, NJ> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  LossGrad<T, NJ> acc;
  zero_grad(acc);
  double loss_acc = 0.0;
  while (true) {
    const T w = valid ? T(scale) * (weights ? weights[it] : T(1)) : T(0);
    T f[KPL][3];
    int n_it = 0;
    DPLL_STAMP(1);
    // mesh geometry: this contact's support point comes from the ICNN kernels; its adjoint goes back to them
    T wit[1][3] = {{T(0), T(0), T(0)}}, rb[1][3] = {{T(0), T(0), T(0)}};
    T L;
    if constexpr (MESH) {
#pragma unroll
      for (int i = 0; i < 3; ++i) wit[0][i] = witness[(it * D::K + cidx) * 3 + i];
      bool winner = true;
      if constexpr (KPL == 1)
        L = loss_item<T, typename Acc<T>::type, NJ, 1, Lanes>(md, dp, opt, xr, xpr, cidx, w, want_grad != 0, acc, f, n_it, wit, rb, nullptr,
                                                              nullptr, &winner);
      if constexpr (RACE > 1) valid = valid && winner;  // (the copies read the same witness; the winner alone writes its adjoint)
      if (rbar_out && valid) {
#pragma unroll
        for (int i = 0; i < 3; ++i) rbar_out[(it * D::K + cidx) * 3 + i] = rb[0][i];
      }
    } else {
      bool winner = true;
      L = loss_item<T, typename Acc<T>::type, NJ, KPL, Lanes>(md, dp, opt, xr, xpr, cidx * KPL, w, want_grad != 0, acc, f, n_it, nullptr,
                                                              nullptr, nullptr, nullptr, &winner);
      if constexpr (RACE > 1) valid = valid && winner;  // the other copies write nothing and add nothing to the row
    }
    if (valid) {
      if (cidx == 0) {
        if (loss) loss[it] = L;
        if (iters) iters[it] = n_it;
      }
      if (force) {
        T* row = force + it * (3 * D::K);
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
          const int contact = cidx * KPL + c;
          row[contact] = f[c][2];
          row[D::K + 2 * contact] = f[c][0];
          row[D::K + 2 * contact + 1] = f[c][1];
        }
      }
    }
    loss_acc += (cidx == 0 && (RACE == 1 || valid)) ? double(w) * double(L) : 0.0;
#ifdef DPLL_STAMPS
    {
      const int mx = __builtin_amdgcn_readfirstlane(n_it);  // not the max, just a sample; max below
      int m = n_it;
      for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
      if (threadIdx.x == 0) g_stamps[blockIdx.x][6] = (unsigned long long)m;
      (void)mx;
    }
#endif
    DPLL_STAMP(2);
    base += stride;
    if (base >= batch) break;
    item = base + slot;
    valid = item < batch;
    it = valid ? item : batch - 1;
#pragma unroll
    for (int i = 0; i < D::NX; ++i) { xr[i] = x[it * ld_x + i]; xpr[i] = xp[it * ld_xp + i]; }
  }
  if (!want_grad) return;
  store_iota_row<T, NJ, G, NJ + 1, 3, WAVES>(acc, loss_acc, partials);
  DPLL_STAMP(3);
}

// Wide build: ONE lane per item, all of its contacts in that lane (the core templates with KPL = K, as the host build
// runs them).  No cross-lane sums inside the solver (a DPP add costs 9.5 cycles against 4 for plain arithmetic, DESIGN
// section 4) and 64 items per wave instead of 16, at the price of a ~2x longer serial chain per iteration: for launches
// far beyond one wave per SIMD only (launch_loss_kernel picks it beyond 32,768 pairs).
template <typename T, int NJ, int WAVES = 1>
__global__ __launch_bounds__(WAVES * kWave) void loss_kernel_wide(const T* __restrict__ x, long long ld_x,
                                                          const T* __restrict__ xp, long long ld_xp, long long batch,
                                                          const T* __restrict__ theta, const T* __restrict__ friction,
                                                          const T* __restrict__ lengths,
                                                          const T* __restrict__ weights, double scale, T* __restrict__ loss,
                                                          T* __restrict__ force, int* __restrict__ iters,
                                                          double* __restrict__ partials, int want_grad, ModelDesc md, SolverOpts opt) {
  using D = Dims<T, NJ>;
  using Lanes = GpuLanes<1>;
  const int lane = threadIdx.x;
  const int item_blocks = (int)gridDim.x - 1;  // the last workgroup owns no items: it writes the chain matrix
  if ((int)blockIdx.x == item_blocks) {
    if (want_grad && lane < kWave) write_chain_matrix<T, T, D::NB>(md.inertia_mode & 1, theta, friction, lengths, partials + (long long)item_blocks * D::PI);
    return;
  }
  Derived<T, NJ> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  LossGrad<T, NJ> acc;
  zero_grad(acc);
  double loss_acc = 0.0;
  const long long stride = (long long)item_blocks * (WAVES * kWave);
  for (long long base = (long long)blockIdx.x * (WAVES * kWave); base < batch; base += stride) {
    const long long item = base + lane;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    T xr[D::NX], xpr[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) { xr[i] = x[it * ld_x + i]; xpr[i] = xp[it * ld_xp + i]; }
    const T w = valid ? T(scale) * (weights ? weights[it] : T(1)) : T(0);
    T f[D::K][3];
    int n_it = 0;
    const T L = loss_item<T, typename Acc<T>::type, NJ, D::K, Lanes>(md, dp, opt, xr, xpr, 0, w, want_grad != 0, acc, f, n_it);
    if (valid) {
      if (loss) loss[it] = L;
      if (iters) iters[it] = n_it;
      if (force) {
        T* row = force + it * (3 * D::K);
#pragma unroll
        for (int c = 0; c < D::K; ++c) {
          row[c] = f[c][2];
          row[D::K + 2 * c] = f[c][0];
          row[D::K + 2 * c + 1] = f[c][1];
        }
      }
    }
    loss_acc += double(w) * double(L);
  }
  if (!want_grad) return;
  store_iota_row<T, NJ, 1, NJ + 1, 3, WAVES>(acc, loss_acc, partials);
}

// sums the per-wave rows in a fixed order (bitwise reproducible) and converts to the parameter dtype.
// 32 row groups x 32 columns = 1024 threads; every thread first issues all of its (independent) loads,
// so the kernel costs about one memory round trip instead of one per row.
constexpr int kFinalizeThreads = 1024;
static_assert(kMaxLossBlocks <= 2048, "finalize_kernel sums at most two passes of 32 rows per thread");

// rows rowg, rowg + 32, ... of one column, all loads issued before the first add (fixed summation order).  The loads
// are unconditional (clamped addresses, the value masked afterwards): no branch per load, all of them in flight at once.
template <int LOADS, int GROUPS = 32>
__device__ __forceinline__ double finalize_column(const double* __restrict__ partials, int n_rows, int stride, int width, int col, int rowg,
                                                  int row0 = 0) {
  double v[LOADS];
  const int c = col < width ? col : width - 1;
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int r = row0 + rowg + GROUPS * i;
    const int rc = r < n_rows ? r : (n_rows > 0 ? n_rows - 1 : 0);  // n_rows = 0 (empty shard): reads the chain area, masked below
    v[i] = partials[(long long)rc * stride + c];
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < LOADS; ++i) s += (col < width && row0 + rowg + GROUPS * i < n_rows) ? v[i] : 0.0;
  return s;
}

// FUSED: the row [loss | gradients] is summed over the ranks of a data-parallel job before it is written, with the
// one-shot peer-memory exchange of dpll_allreduce.hpp (same protocol and call counter as dpll_ar_allreduce): the
// gradient exchange costs no launch of its own.
// GROUPS: row groups = waves / 2 of the one workgroup (32: 1024 threads; 8: 256 threads, for launches of <= 256 rows)
template <typename T, int NJ, bool FUSED = false, int GROUPS = 32>
__global__ __launch_bounds__(GROUPS * 32) void finalize_kernel(const double* __restrict__ partials, int n_rows,
                                                                    T* __restrict__ grad, T* __restrict__ loss_total,
                                                                    dpll_arx::Peers peers, int rank, int world,
                                                                    uint32_t* __restrict__ seq_ptr, uint32_t* __restrict__ err,
                                                                    AdamArgs adam) {
  using D = Dims<T, NJ>;
  static_assert(D::PI <= 32, "partial row must fit 32 columns");
  __shared__ double red[GROUPS][33];
  __shared__ double tot[32];
  const int col = threadIdx.x & 31, rowg = threadIdx.x >> 5;
  // uniform branch (n_rows is a kernel argument): the headline grid has 256 rows = 8 per thread
  const double s = GROUPS == 8 ? finalize_column<32, 8>(partials, n_rows, D::PI, D::PIOTA, col, rowg)
                   : n_rows <= 256 ? finalize_column<8>(partials, n_rows, D::PI, D::PIOTA, col, rowg)
                   : n_rows <= 512 ? finalize_column<16>(partials, n_rows, D::PI, D::PIOTA, col, rowg)
                   : n_rows <= 1024 ? finalize_column<32>(partials, n_rows, D::PI, D::PIOTA, col, rowg)
                                    : finalize_column<32>(partials, n_rows, D::PI, D::PIOTA, col, rowg) +
                                          finalize_column<32>(partials, n_rows, D::PI, D::PIOTA, col, rowg, 1024);
  // thread 1 + k writes learnable parameter k: the chain matrix (written behind the rows by the producing kernel's extra
  // workgroup) applied to the row sum.  (Requesting the <= 10 coefficients per parameter together with the partial rows
  // was measured slower: 3.9 vs 2.9 us -- 1024 threads asking for the same few lines.)
  const int k = (int)threadIdx.x - 1;
  const double* chain = partials + (long long)n_rows * D::PI;
  red[rowg][col] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < GROUPS; ++r) t += red[r][col];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  T value = T(0);
  if (threadIdx.x < D::PI) value = T(k < 0 ? tot[0] : apply_chain<D::NB>(tot, chain, k));
  if constexpr (FUSED) {
    constexpr int kWordsPer = sizeof(T) / 4;
    __shared__ uint32_t words[D::PI * kWordsPer];
    __shared__ uint32_t gathered[dpll_arx::kMaxWorld][dpll_arx::kMaxWords];
    const uint32_t seq = *seq_ptr + 1u;
    if (threadIdx.x < D::PI) __builtin_memcpy(&words[threadIdx.x * kWordsPer], &value, sizeof(T));
    __syncthreads();
    dpll_arx::exchange_words(words, gathered, D::PI * kWordsPer, peers, rank, world, seq, err);
    if (threadIdx.x < D::PI) value = dpll_arx::sum_over_ranks<T>(gathered, world, threadIdx.x);
    if (threadIdx.x == 0) *seq_ptr = seq;
  }
  if (threadIdx.x < D::PI) {
    if (threadIdx.x == 0) {
      if (loss_total) *loss_total = value;
    } else {
      grad[threadIdx.x - 1] = value;
    }
  }
  // A timed-out exchange hands back NaN rows by design (dpll_allreduce.hpp) and sets the error word: the update is then
  // skipped altogether -- parameters, both moments and the step count keep the values of the previous step, so the abort the
  // trainer raises when it next looks at the word (every 16 steps) finds nothing that has to be undone.
  bool exchange_ok = true;
  if constexpr (FUSED) exchange_ok = err == nullptr || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
  if (adam.params && exchange_ok) {  // (uniform: kernel arguments and one word every thread reads alike)
    double steps, pow1, pow2;
    adam_powers(adam, steps, pow1, pow2);
    __syncthreads();  // every thread has read the state before thread 0 advances it
    if (threadIdx.x == 0) { adam.state[0] = steps; adam.state[1] = pow1; adam.state[2] = pow2; }
    if (threadIdx.x >= 1 && threadIdx.x < D::PI) adam_apply<T>(adam, (int)threadIdx.x - 1, double(value), pow1, pow2);
  }
}

// ---- simulation: `steps` VelocityIntegrator steps per item, trajectory written as it goes ---------
// RACE: racing copies of every item (as in the loss kernel): a rollout's wave pays, at every step, for its slowest item --
// with four copies it holds 4 items instead of 16 and an item needs the fewest iterations of its copies
// SOLO: for launches of at most one wave per SIMD (claim_whole_simd)
template <typename T, int NJ, bool MESH = false, int RACE = 1, bool SOLO = (RACE > 1)>
__global__ __launch_bounds__(kWave) void simulate_kernel(ModelDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                         const T* __restrict__ friction, const T* __restrict__ lengths,
                                                         const T* __restrict__ x0, long long ld_x, long long batch,
                                                         long long steps, T* __restrict__ out, long long ld_item,
                                                         long long ld_step, int write_x0, int* __restrict__ iters,
                                                         const T* __restrict__ witness) {
  using D = Dims<T, NJ>;
  using Lanes = GpuLanes<D::G, RACE>;
  static_assert(RACE == 1 || !MESH, "racing copies: box geometry");
  constexpr int kItems = D::IPW / RACE;  // items per wave
  if constexpr (SOLO) claim_whole_simd<T>();
  const int lane = threadIdx.x;
  const int cidx = lane % (D::G * RACE);  // (0: the lane that writes the item's rows)
  const int slot = lane / (D::G * RACE);
  Derived<T, NJ> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  const long long stride = (long long)gridDim.x * kItems;
  for (long long base = (long long)blockIdx.x * kItems; base < batch; base += stride) {
    const long long item = base + slot;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    T xr[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) xr[i] = x0[it * ld_x + i];
    T* dst = out + it * ld_item;
    if (write_x0) {
      if (valid && cidx == 0) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) dst[i] = xr[i];
      }
      dst += ld_step;
    }
    int total = 0;
    for (long long s = 0; s < steps; ++s) {
      T xn[D::NX], imp[1][3];
      int n_it = 0;
      if constexpr (MESH) {  // mesh geometry: support points of the CURRENT state, one step per launch
        T wit[1][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) wit[0][i] = witness[(it * D::K + cidx) * 3 + i];
        step_item<T, typename Acc<T>::type, NJ, 1, Lanes>(md, dp, opt, xr, cidx, xn, imp, n_it, wit);
      } else {
        step_item<T, typename Acc<T>::type, NJ, 1, Lanes>(md, dp, opt, xr, cidx % D::G, xn, imp, n_it);
      }
      total += n_it;
#pragma unroll
      for (int i = 0; i < D::NX; ++i) xr[i] = xn[i];
      if (valid && cidx == 0) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) dst[i] = xr[i];
      }
      dst += ld_step;
    }
    if (iters && valid && cidx == 0) iters[it] = total;
  }
}

// Wide build of the rollout kernel: ONE lane per trajectory, all of its contacts in that lane (as loss_kernel_wide): for
// launches far beyond one wave per SIMD, where the lane-per-contact build needs several rounds of waves.
template <typename T, int NJ, bool SOLO = false>
__global__ __launch_bounds__(kWave) void simulate_kernel_wide(ModelDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                              const T* __restrict__ friction, const T* __restrict__ lengths,
                                                              const T* __restrict__ x0, long long ld_x, long long batch,
                                                              long long steps, T* __restrict__ out, long long ld_item,
                                                              long long ld_step, int write_x0, int* __restrict__ iters) {
  using D = Dims<T, NJ>;
  using Lanes = GpuLanes<1>;
  if constexpr (SOLO) claim_whole_simd<T>();
  Derived<T, NJ> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  const long long stride = (long long)gridDim.x * kWave;
  for (long long base = (long long)blockIdx.x * kWave; base < batch; base += stride) {
    const long long item = base + threadIdx.x;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    T xr[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) xr[i] = x0[it * ld_x + i];
    T* dst = out + it * ld_item;
    if (write_x0) {
      if (valid) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) dst[i] = xr[i];
      }
      dst += ld_step;
    }
    int total = 0;
    for (long long s = 0; s < steps; ++s) {
      T xn[D::NX], imp[D::K][3];
      int n_it = 0;
      step_item<T, typename Acc<T>::type, NJ, D::K, Lanes>(md, dp, opt, xr, 0, xn, imp, n_it);
      total += n_it;
#pragma unroll
      for (int i = 0; i < D::NX; ++i) xr[i] = xn[i];
      if (valid) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) dst[i] = xr[i];
      }
      dst += ld_step;
    }
    if (iters && valid) iters[it] = total;
  }
}

// ---- adjoint of one simulation step with respect to the parameters (state = data) -----------------------
// STATE: also the adjoint of the input state (n_x forward-mode passes of the terms, core step_state_adjoint)
// MESH: witnesses from the ICNN kernels in, their adjoint r_bar out (as in the loss kernel)
// The arithmetic is double for float storage too: lambda = H^-1 s at the dynamics' eps = 1e-4 amplifies float rounding of
// H by its condition number (float32 state adjoints were off by 1e-3 .. 0.2 on items in contact), and this kernel is
// not on the hot path.  The float loss kernel keeps its float arithmetic; only the step's backward pays for double.
template <typename T, int NJ, bool STATE = false, bool MESH = false>
__global__ __launch_bounds__(kWave) void step_backward_kernel(ModelDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                              const T* __restrict__ friction,
                                                              const T* __restrict__ lengths, const T* __restrict__ x,
                                                              long long ld_x, const T* __restrict__ gx,
                                                              long long ld_g, long long batch,
                                                              double* __restrict__ partials, T* __restrict__ xbar_out,
                                                              long long ld_xb, const T* __restrict__ witness,
                                                              T* __restrict__ rbar_out) {
  using D = Dims<T, NJ>;
  using C = double;
  using Lanes = GpuLanes<D::G>;
  const int lane = threadIdx.x;
  const int cidx = lane % D::G;
  const int slot = lane / D::G;
  const int item_blocks = (int)gridDim.x - 1;  // the last workgroup owns no items: it writes the chain matrix
  if ((int)blockIdx.x == item_blocks) {
    write_chain_matrix<C, T, D::NB>(md.inertia_mode & 1, theta, friction, lengths, partials + (long long)item_blocks * D::PI);
    return;
  }
  C theta_c[D::NB * 10], friction_c[D::NB + 1], lengths_c[D::NB * 3];
#pragma unroll
  for (int i = 0; i < D::NB * 10; ++i) theta_c[i] = C(theta[i]);
#pragma unroll
  for (int i = 0; i < D::NB + 1; ++i) friction_c[i] = C(friction[i]);
#pragma unroll
  for (int i = 0; i < D::NB * 3; ++i) lengths_c[i] = lengths ? C(lengths[i]) : C(0);
  Derived<C, NJ> dp;
  derive_params<C, NJ>(md, theta_c, friction_c, lengths ? lengths_c : nullptr, dp);
  LossGrad<C, NJ> acc;
  zero_grad(acc);
  const long long stride = (long long)item_blocks * D::IPW;
  for (long long base = (long long)blockIdx.x * D::IPW; base < batch; base += stride) {
    const long long item = base + slot;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    C xr[D::NX], gr[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) { xr[i] = C(x[it * ld_x + i]); gr[i] = valid ? C(gx[it * ld_g + i]) : C(0); }
    LossGrad<C, NJ> g;
    zero_grad(g);
    C wit[1][3] = {{C(0), C(0), C(0)}}, rb[1][3] = {{C(0), C(0), C(0)}};
    if constexpr (MESH) {
#pragma unroll
      for (int i = 0; i < 3; ++i) wit[0][i] = C(witness[(it * D::K + cidx) * 3 + i]);
    }
    C xb[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) xb[i] = C(0);
    if constexpr (STATE && MESH)
      step_item_backward<C, C, NJ, 1, Lanes>(md, dp, opt, xr, cidx, gr, g, wit, rb, &xb);
    else if constexpr (STATE)
      step_item_backward<C, C, NJ, 1, Lanes>(md, dp, opt, xr, cidx, gr, g, nullptr, nullptr, &xb);
    else if constexpr (MESH)
      step_item_backward<C, C, NJ, 1, Lanes>(md, dp, opt, xr, cidx, gr, g, wit, rb);
    else
      step_item_backward<C, C, NJ, 1, Lanes>(md, dp, opt, xr, cidx, gr, g);
    if constexpr (STATE) {
      if (valid && cidx == 0) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) xbar_out[it * ld_xb + i] = T(xb[i]);
      }
    }
    if constexpr (MESH) {
      if (valid) {
#pragma unroll
        for (int i = 0; i < 3; ++i) rbar_out[(it * D::K + cidx) * 3 + i] = T(rb[0][i]);
      }
    }
    // every lane of the group holds the item's d/d iota; d/d mu and d/d length are per contact
#pragma unroll
    for (int b = 0; b < D::NB; ++b) {
#pragma unroll
      for (int i = 0; i < kIota; ++i) acc.g_iota[b][i] += (cidx == 0) ? g.g_iota[b][i] : C(0);
      acc.g_mu[b] += g.g_mu[b];
#pragma unroll
      for (int i = 0; i < 3; ++i) acc.g_len[b][i] += g.g_len[b][i];
    }
  }
  // g_iota here is NOT replicated (only the group leader kept it): spread it so the shared tail applies
#pragma unroll
  for (int b = 0; b < D::NB; ++b)
#pragma unroll
    for (int i = 0; i < kIota; ++i) acc.g_iota[b][i] = Lanes::group_sum(acc.g_iota[b][i]);
  store_iota_row<C, NJ>(acc, 0.0, partials);
}

// ---- MultibodyTerms.forward for API parity (off the hot path: the loss / step kernels never form D) ----
template <typename T, int NJ>
__global__ __launch_bounds__(kWave) void terms_kernel(ModelDesc md, const T* __restrict__ theta,
                                                      const T* __restrict__ friction, const T* __restrict__ lengths,
                                                      const T* __restrict__ x, long long ld_x, long long batch,
                                                      T* __restrict__ Dout, T* __restrict__ Mout, T* __restrict__ Jout,
                                                      T* __restrict__ phiout, T* __restrict__ aout,
                                                      const T* __restrict__ witness) {
  using D = Dims<T, NJ>;
  constexpr int NV = D::NV, K = D::K;
  __shared__ T Jrows[D::IPW][3 * K][NV];
  const int lane = threadIdx.x;
  const int cidx = lane % D::G;
  const int slot = lane / D::G;
  Derived<T, NJ> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  const long long stride = (long long)gridDim.x * D::IPW;
  for (long long base = (long long)blockIdx.x * D::IPW; base < batch; base += stride) {
    const long long item = base + slot;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    T xr[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) xr[i] = x[it * ld_x + i];
    Terms<T, NJ> t;
    Kin<typename Acc<T>::type, NJ> kinA;
    compute_terms<T, typename Acc<T>::type, NJ>(md, dp, xr, xr + D::NQ, t, kinA);
    ContactGeom<T, NJ> cg;
    T wit[3] = {T(0), T(0), T(0)};  // mesh geometry: the support point from the ICNN kernels (off the hot path)
    if (witness) {
#pragma unroll
      for (int i = 0; i < 3; ++i) wit[i] = witness[(it * K + cidx) * 3 + i];
    }
    compute_contact<T, typename Acc<T>::type, NJ>(md, dp, t.kin, kinA, cidx, cg, witness ? wit : nullptr);
    // rows of J in the reference order [normals | mu (t_x, t_y) per contact] (multibody_terms.py:415-426)
    T mine[3][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      T col[3];
      cjac_column<T, NJ>(cg.J, i, col);
      mine[0][i] = col[2]; mine[1][i] = cg.mu * col[0]; mine[2][i] = cg.mu * col[1];
    }
    const int rows[3] = {cidx, K + 2 * cidx, K + 2 * cidx + 1};
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NV; ++i) Jrows[slot][rows[r]][i] = mine[r][i];
    __syncthreads();
    if (valid) {
      if (phiout) phiout[it * K + cidx] = cg.phi;
      if (Jout) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int i = 0; i < NV; ++i) Jout[(it * 3 * K + rows[r]) * NV + i] = mine[r][i];
      }
      if (cidx == 0) {
        if (Mout) {
#pragma unroll
          for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < NV; ++j) Mout[(it * NV + i) * NV + j] = t.M[i][j];
        }
        if (aout) {
#pragma unroll
          for (int i = 0; i < NV; ++i) aout[it * NV + i] = t.a[i];
        }
      }
      if (Dout) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          T w[NV];
          chol_solve<T, NV>(t.LM, t.invdM, mine[r], w);
          for (int c = 0; c < 3 * K; ++c) {
            T s = T(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) s += w[i] * Jrows[slot][c][i];
            Dout[(it * 3 * K + rows[r]) * (3 * K) + c] = s;
          }
        }
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------
thread_local char g_error[512] = "";

int fail(int code, const char* fmt, const char* detail = "") {
  std::snprintf(g_error, sizeof(g_error), fmt, detail);
  return code;
}

int check_launch(const char* what) {
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    std::snprintf(g_error, sizeof(g_error), "%s: %s", what, hipGetErrorString(err));
    return -5;
  }
  return 0;
}

SolverOpts default_opts(int dtype, int n_joints = 0, bool general = false) {
  SolverOpts o;
  if (dtype == DPLL_F64) {
    o.max_iter = 100; o.max_ls = 50; o.tol = 1e-13; o.stall_tol = 1e-10; o.ls_tol = 0.9;
    o.n_stages = 6; o.stage_max_iter = 3; o.stage_factor = 3.0; o.stage_tol = 0.3; o.stage_ls_tol = 0.9; o.stage_max_ls = 50; o.fast_ls = 1;
    o.warm_start = 0; o.wide = -1;
  } else {
    o.max_iter = 60; o.max_ls = 30; o.tol = 1e-6; o.stall_tol = 1e-5; o.ls_tol = 0.9;
    o.n_stages = 6; o.stage_max_iter = 3; o.stage_factor = 3.0; o.stage_tol = 0.3; o.stage_ls_tol = 0.9; o.stage_max_ls = 50; o.fast_ls = 1;
    o.warm_start = 0; o.wide = -1;
  }
  // the elbow's loss solve: 5 stages of 2.5 (worst case 21 -> 18 iterations in f32, 23 -> 20 in f64 on the 4096-pair batch;
  // the cube keeps 6 x 3.0: 14 against 15; the general build is not tuned)
  o.loss_n_stages = (!general && n_joints == 1) ? 5 : 0;
  o.loss_stage_factor = 2.5;
  o.f64_refine = 1;
  // the float32 ICNN GEMMs: two fp16 planes (dpll_mesh_bf16.hpp) -- f32-grade products (against float64 on the benchmark batch: no
  // support point on another vertex, loss error 4e-8, worst gradient error 1.4e-6; the f32 MFMA form: 0, 5e-8, 1.1e-6) at 98 us per
  // step instead of 140; 0 selects the f32 MFMA kernels
  o.mesh_gemm = 4;
  // racing copies of the loss solve (launches of <= 4096 cube pairs: 4 copies): the schedules that, together with the one
  // above, had the lowest worst case over EIGHT 4096-pair samples of the reference's 57,812 toss pairs, the benchmark batch
  // among them (tools/diag/race_schedules.py and its multi-sample search; all cold starts on full Newton steps): slowest
  // item 14 / 16 / 15 / 14 / 15 / 16 / 13 / 14 -> 11 / 11 / 11 / 11 / 11 / 12 / 11 / 12 iterations, mean 7.0 -> 4.7.  (The
  // first table -- picked on the benchmark batch alone: 1, 2 x 30, 5 x 2 -- left 12 / 13 / 16 on three of the others.)
  // The elbow: four copies on its two-contacts-per-lane build (float: 18 -> 13 iterations on the 4096-pair batch; the copies are
  // the best pure-Newton triple of tools/diag/race_schedules.py elbow_box_4096 4)
  o.portfolio = 0;
  if (n_joints == 1 && !general) {
    o.race_stages[0] = 2; o.race_factor[0] = 5.0; o.race_flags[0] = 2;
    o.race_stages[1] = 5; o.race_factor[1] = 2.0; o.race_flags[1] = 2;
    o.race_stages[2] = 8; o.race_factor[2] = 2.0; o.race_flags[2] = 2;
  } else {
    o.race_stages[0] = 1; o.race_factor[0] = 1.0; o.race_flags[0] = 2;
    o.race_stages[1] = 2; o.race_factor[1] = 100.0; o.race_flags[1] = 2;
    o.race_stages[2] = 6; o.race_factor[2] = 2.0; o.race_flags[2] = 2;
  }
  return o;
}

// item workgroups (= partial rows) of a launch whose waves hold `ipw` items each
inline int blocks_for(long long batch, int ipw) {
  long long blocks = (batch + ipw - 1) / ipw;
  if (blocks > kMaxLossBlocks) blocks = kMaxLossBlocks;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}
// lanes of one copy of an item in the loss launch: one per contact, except the elbow's four-copy racing build (two contacts
// per lane, so that four copies fit the item's 16-lane row)
// (measured, elbow 4096 pairs, loss kernel us with 1 / 2 / 4 copies: f32 33.2 / 31.5 / 30.0 -- slowest item 18 / 17 / 13 iterations at
// +20 % per iteration for the second contact in the lane --, f64 52.5 / 57.1 / 63.3: the default races the float elbow only)
inline int race_lanes_of(int n_joints, bool is_float, int copies_asked) {  // (the one rule: launchers and dpll_racing_copies)
  return (n_joints == 1 && (copies_asked == 4 || (copies_asked == 0 && is_float))) ? 4 : kQuery * (n_joints + 1);
}
template <typename T, int NJ> int race_lanes(int copies_asked) { return race_lanes_of(NJ, sizeof(T) == 4, copies_asked); }
template <typename T, int NJ> int loss_blocks(long long batch, int copies = 1, int lanes = Dims<T, NJ>::G) {
  return blocks_for(batch, copies > 1 ? kRaceWaves * kWave / (lanes * copies) : Dims<T, NJ>::IPW);
}

// Racing copies per item (dpll_solver_opts_t::portfolio): what was asked for -- 0: `preferred`, the number measured best for
// the launch in question -- reduced until an item stays inside a 16-lane row and the launch within one wave per SIMD (the
// copies use SIMDs that would idle, they never make a wave wait for one).  A default that cannot have its preferred
// number runs without copies: fewer copies buy too little (cube loss: two copies save one iteration of fourteen).
inline int race_copies(int requested, int lanes_per_item, long long batch, int preferred = 4, bool fewer_is_fine = false) {
  int p = requested == 0 ? preferred : requested;
  if (batch < 1) return 1;
  while (p > 1 && (lanes_per_item * p > 16 || (long long)kRaceWaves * blocks_for(batch, kRaceWaves * kWave / (lanes_per_item * p)) > kSimds))
    p >>= 1;
  if (requested == 0 && p < preferred && !fewer_is_fine) p = 1;
  return p;
}

// The shape of a loss launch with racing copies: lanes of one copy of an item, copies per item.  The two-contacts-per-lane
// build (the elbow: 4 lanes per copy) exists with FOUR copies only, so when the launch cannot have four (a batch beyond one
// wave per SIMD) the shape is decided again on the one-contact-per-lane build -- lanes first and copies second used to leave
// a two-copy launch sized for workgroups of twice its items (correct through the grid-stride loop, half the workgroups).
struct RaceShape { int lanes, copies; };
inline RaceShape race_shape(int n_joints, bool is_float, int asked, long long batch) {
  const int full = kQuery * (n_joints + 1);
  RaceShape r{race_lanes_of(n_joints, is_float, asked), 1};
  r.copies = race_copies(asked, r.lanes, batch);
  if (r.lanes < full && r.copies != 4) {
    r.lanes = full;
    r.copies = race_copies(asked, full, batch);
  }
  return r;
}
template <typename T, int NJ> RaceShape race_shape(const dpll_model* m, int dtype, long long batch) {
  return race_shape(NJ, sizeof(T) == 4, m->opts[dtype].portfolio, batch);
}

}  // namespace

int dpll_fail(int code, const char* fmt, const char* detail) { return fail(code, fmt, detail); }
int dpll_check_launch(const char* what) { return check_launch(what); }

namespace {

// Which build of the loss kernel a launch of `batch` pairs runs and with what grid: decided in ONE place, for the launcher,
// the workspace check, the profiling utility and dpll_racing_copies (what a caller / a test may ask about a launch).
enum LossBuild { kLossEmpty, kLossWide, kLossWideShared, kLossDense, kLossPlainShared, kLossPlain, kLossRace2, kLossRace4, kLossRace4Kpl2 };
struct LossPlan {
  LossBuild build;
  int lanes, copies;  // lanes of one copy of an item, racing copies per item
  int blocks;         // item workgroups of the one-wave plain launch (the workspace is sized for these rows)
  int rows;           // item workgroups of THIS launch = partial rows written (the grid is rows + 1: the chain-matrix workgroup)
  int threads;        // per workgroup
};
template <typename T, int NJ> LossPlan plan_loss(const dpll_model* m, int dtype, long long batch) {
  const RaceShape shape = race_shape<T, NJ>(m, dtype, batch);
  LossPlan pl{kLossPlain, shape.lanes, shape.copies, loss_blocks<T, NJ>(batch, shape.copies, shape.lanes), 0, kWave};
  pl.rows = pl.blocks;
  // wide build (one lane per item): beyond 32,768 pairs.  Measured (round 3, waves that claim their SIMD, shared partial rows),
  // loss + finalize in us, lane-per-contact builds vs wide: cube f32 16,384 pairs 22.2 vs 34.8, 32,768 32.0 vs 34.9, 49,152
  // 41.3 vs 34.8, 65,536 55.6 vs 35.2; cube f64 32,768 63.5 vs 70.7, 49,152 81.6 vs 73.2; elbow f32 32,768 112.6 vs 122.1,
  // 49,152 163.7 vs 132.1; elbow f64 stays on the lane-per-contact builds (65,536: 118 vs 85 M steps/s, 3.2 KB of spills)
  // dpll_solver_opts_t::wide overrides the choice (tests compare the builds on the same inputs)
  const int wide_opt = m->opts[dtype].wide;
  const bool wide = wide_opt >= 0 ? wide_opt == 1 : (batch > 32768 && !(std::is_same<T, double>::value && NJ == 1));
  // beyond 512 waves per launch the workgroups are four waves that share a partial row: the finalize kernel's time follows the
  // number of rows (256 rows 3.0 us, 1024 rows 5.5, 2048 rows 8.7)
  constexpr int kShare = 4;
  if (batch == 0) {  // an empty shard: no item workgroups, only the one that writes the chain matrix; zero partial rows
    pl.build = kLossEmpty; pl.rows = 0;
  } else if (wide) {
    long long wb = (batch + kWave - 1) / kWave;
    if (wb > kMaxLossBlocks) wb = kMaxLossBlocks;
    if (wb >= 512) { pl.build = kLossWideShared; pl.rows = (int)((wb + kShare - 1) / kShare); pl.threads = kShare * kWave; }
    else { pl.build = kLossWide; pl.rows = (int)wb; }
  } else if (pl.blocks > kSimds) {  // more waves than SIMDs: the two-waves-per-SIMD build
    pl.build = kLossDense; pl.rows = (pl.blocks + kShare - 1) / kShare; pl.threads = kShare * kWave;
  } else if (pl.copies == 1 && pl.blocks > 512) {  // (at 512 waves the one-wave workgroups spread over all CUs, two each: the double
                                                   // elbow kernel, 512 registers and spills, lost 6 % when packed four to a CU)
    pl.build = kLossPlainShared; pl.rows = (pl.blocks + kShare - 1) / kShare; pl.threads = kShare * kWave;
  } else if (pl.copies == 1) {
    pl.build = kLossPlain;
  } else {
    pl.threads = kRaceWaves * kWave;
    pl.build = pl.copies == 2 ? kLossRace2 : (pl.lanes * 4 <= 16 && pl.lanes == Dims<T, NJ>::G ? kLossRace4 : kLossRace4Kpl2);
  }
  return pl;
}

// launches the build plan_loss picked; returns the number of partial rows written
template <typename T, int NJ>
int launch_loss_kernel(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp,
                       long long ld_xp, long long batch, const void* weights, double scale, void* loss, void* force,
                       int32_t* iters, void* workspace, int want_grad, hipStream_t stream) {
  const LossPlan pl = plan_loss<T, NJ>(m, dtype, batch);
  constexpr int kShare = 4;
#define DPLL_LOSS_ARGS (const T*)x, ld_x, (const T*)xp, ld_xp, batch, (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, \
                       (const T*)weights, scale, (T*)loss, (T*)force, (int*)iters, (double*)workspace, want_grad
  const dim3 grid(pl.rows + 1), block(pl.threads);
  switch (pl.build) {
    case kLossEmpty:
    case kLossPlain:
      hipLaunchKernelGGL((loss_kernel<T, NJ, false, false>), grid, block, 0, stream, DPLL_LOSS_ARGS, (const T*)nullptr, (T*)nullptr, m->desc, m->opts[dtype]);
      break;
    case kLossWideShared:
      hipLaunchKernelGGL((loss_kernel_wide<T, NJ, kShare>), grid, block, 0, stream, DPLL_LOSS_ARGS, m->desc, m->opts[dtype]);
      break;
    case kLossWide:
      hipLaunchKernelGGL((loss_kernel_wide<T, NJ>), grid, block, 0, stream, DPLL_LOSS_ARGS, m->desc, m->opts[dtype]);
      break;
    case kLossDense:
      hipLaunchKernelGGL((loss_kernel<T, NJ, false, true, 1, kShare>), grid, block, 0, stream, DPLL_LOSS_ARGS, (const T*)nullptr, (T*)nullptr, m->desc, m->opts[dtype]);
      break;
    case kLossPlainShared:
      hipLaunchKernelGGL((loss_kernel<T, NJ, false, false, 1, kShare>), grid, block, 0, stream, DPLL_LOSS_ARGS, (const T*)nullptr, (T*)nullptr, m->desc, m->opts[dtype]);
      break;
    case kLossRace2:
      hipLaunchKernelGGL((loss_kernel<T, NJ, false, false, 2>), grid, block, 0, stream, DPLL_LOSS_ARGS, (const T*)nullptr, (T*)nullptr, m->desc, m->opts[dtype]);
      break;
    case kLossRace4:
      if constexpr (Dims<T, NJ>::G * 4 <= 16)
        hipLaunchKernelGGL((loss_kernel<T, NJ, false, false, 4>), grid, block, 0, stream, DPLL_LOSS_ARGS, (const T*)nullptr, (T*)nullptr, m->desc, m->opts[dtype]);
      break;
    case kLossRace4Kpl2:  // the elbow: two contacts per lane
      if constexpr (Dims<T, NJ>::G == 8)
        hipLaunchKernelGGL((loss_kernel<T, NJ, false, false, 4, kRaceWaves, 2>), grid, block, 0, stream, DPLL_LOSS_ARGS, (const T*)nullptr,
                           (T*)nullptr, m->desc, m->opts[dtype]);
      break;
  }
#undef DPLL_LOSS_ARGS
  return pl.rows;
}

// the finalize kernel behind a loss launch of `rows` partial rows (<= 256 rows: the 256-thread build, with or without the
// exchange -- the two sum the rows in the same order, so a separate exchange after the launch gives bitwise the row of the
// fused one); the ONE rule for the launcher and the profiling utility
template <typename T, int NJ>
void launch_finalize(const void* workspace, int rows, void* grad, void* loss_total, const dpll_ar* ar, const AdamArgs& adam, hipStream_t stream) {
  const dpll_arx::Peers peers = ar ? ar->peers : dpll_arx::Peers{};
  const int rank = ar ? ar->rank : 0, world = ar ? ar->world : 1;
  uint32_t* seq = ar ? ar->state : (uint32_t*)nullptr;
  uint32_t* err = ar ? ar->state + 1 : (uint32_t*)nullptr;
  if (ar && rows <= 256)
    hipLaunchKernelGGL((finalize_kernel<T, NJ, true, 8>), dim3(1), dim3(256), 0, stream, (const double*)workspace, rows, (T*)grad, (T*)loss_total,
                       peers, rank, world, seq, err, adam);
  else if (ar)
    hipLaunchKernelGGL((finalize_kernel<T, NJ, true>), dim3(1), dim3(kFinalizeThreads), 0, stream, (const double*)workspace, rows, (T*)grad,
                       (T*)loss_total, peers, rank, world, seq, err, adam);
  else if (rows <= 256)
    hipLaunchKernelGGL((finalize_kernel<T, NJ, false, 8>), dim3(1), dim3(256), 0, stream, (const double*)workspace, rows, (T*)grad, (T*)loss_total,
                       peers, rank, world, seq, err, adam);
  else
    hipLaunchKernelGGL((finalize_kernel<T, NJ, false>), dim3(1), dim3(kFinalizeThreads), 0, stream, (const double*)workspace, rows, (T*)grad,
                       (T*)loss_total, peers, rank, world, seq, err, adam);
}

template <typename T, int NJ>
int launch_loss(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp,
                long long ld_xp, long long batch, const void* weights, double scale, void* loss, void* grad,
                void* loss_total, void* force, int32_t* iters, void* workspace, long long workspace_bytes,
                hipStream_t stream, const dpll_ar* ar = nullptr, AdamArgs adam = AdamArgs{}) {
  using D = Dims<T, NJ>;
  const RaceShape shape = race_shape<T, NJ>(m, dtype, batch);
  const int blocks = loss_blocks<T, NJ>(batch, shape.copies, shape.lanes);
  const int want_grad = grad != nullptr;
  if (want_grad) {
    if (!workspace || workspace_bytes < ((long long)blocks * D::PI + D::CHAIN) * (long long)sizeof(double))
      return fail(-3, "dpll_contactnets_loss: workspace too small%s");
  } else if (loss_total) {
    return fail(-3, "dpll_contactnets_loss: loss_total requires grad%s");
  }
  const int rows = launch_loss_kernel<T, NJ>(m, dtype, p, x, ld_x, xp, ld_xp, batch, weights, scale, loss, force, iters, workspace,
                                             want_grad, stream);
  if (int rc = check_launch("loss_kernel")) return rc;
  if (want_grad) {
    launch_finalize<T, NJ>(workspace, rows, grad, loss_total, ar, adam, stream);
    if (int rc = check_launch("finalize_kernel")) return rc;
  }
  return 0;
}

// measuring utility: `reps` back-to-back loss+finalize launches with HIP events recorded on the launch
// stream around each kernel; returns the average duration of each kernel in milliseconds
template <typename T, int NJ>
int profile_loss(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp,
                 long long ld_xp, long long batch, double scale, void* grad, void* workspace, long long workspace_bytes,
                 hipStream_t stream, int reps, float* ms_loss, float* ms_finalize) {
  using D = Dims<T, NJ>;
  const RaceShape shape = race_shape<T, NJ>(m, dtype, batch);
  const int blocks = loss_blocks<T, NJ>(batch, shape.copies, shape.lanes);
  if (!grad || !workspace || workspace_bytes < ((long long)blocks * D::PI + D::CHAIN) * (long long)sizeof(double))
    return fail(-3, "dpll_profile_contactnets_loss: grad and workspace are required%s");
  // two passes, two events each (an event between every pair of kernels costs several microseconds of its own):
  // `reps` loss kernels back to back, then `reps` (loss, finalize) pairs; finalize = the difference
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ok = true;
  for (int i = 0; i < 4; ++i) ok = ok && hipEventCreate(&ev[i]) == hipSuccess;
  int rows = blocks;
  auto launch_loss_only = [&]() {
    rows = launch_loss_kernel<T, NJ>(m, dtype, p, x, ld_x, xp, ld_xp, batch, nullptr, scale, nullptr, nullptr, nullptr, workspace, 1,
                                     stream);
  };
  int rc = 0;
  float t_loss = 0.f, t_pair = 0.f;
  if (ok) {
    ok = hipEventRecord(ev[0], stream) == hipSuccess;
    for (int r = 0; r < reps; ++r) launch_loss_only();
    ok = ok && hipEventRecord(ev[1], stream) == hipSuccess && hipEventRecord(ev[2], stream) == hipSuccess;
    for (int r = 0; r < reps; ++r) {
      launch_loss_only();
      launch_finalize<T, NJ>(workspace, rows, grad, nullptr, nullptr, AdamArgs{}, stream);  // (the build launch_loss ships for these rows)
    }
    ok = ok && hipEventRecord(ev[3], stream) == hipSuccess;
    rc = check_launch("profile launches");
    ok = ok && hipEventSynchronize(ev[3]) == hipSuccess && hipEventElapsedTime(&t_loss, ev[0], ev[1]) == hipSuccess &&
         hipEventElapsedTime(&t_pair, ev[2], ev[3]) == hipSuccess;
  }
  for (int i = 0; i < 4; ++i)
    if (ev[i]) (void)hipEventDestroy(ev[i]);
  if (rc) return rc;
  if (!ok) return fail(-5, "dpll_profile_contactnets_loss: a HIP event call failed%s");
  const double t_fin = t_pair > t_loss ? t_pair - t_loss : 0.0;
  if (ms_loss) *ms_loss = (float)(t_loss / reps);
  if (ms_finalize) *ms_finalize = (float)(t_fin / reps);
  return 0;
}

template <typename T, int NJ>
int launch_simulate(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x0, long long ld_x,
                    long long batch, long long steps, void* out, long long ld_item, long long ld_step, int write_x0,
                    int32_t* iters, hipStream_t stream, const void* witness = nullptr) {
  using D = Dims<T, NJ>;
  long long blocks = (batch + D::IPW - 1) / D::IPW;
  if (blocks > 8192) blocks = 8192;
  // racing copies (the loss launch's rule: four per item while the launch stays within one wave per SIMD)
  // (measured, 4096 cube rollouts of 80 steps, us per step with 1 / 2 / 4 copies: f32 13.1 / 11.4 / 10.4, f64 19.6 / 18.2 / 16.6;
  // elbow f32 18.6 / 18.2: the elbow runs without.  Before the waves claimed their SIMDs, four float copies took 14.8)
  const int copies = witness ? 1 : dpll_racing_copies(m, dtype, batch, 1);
  // wide build: the loss launch's rule (dpll_solver_opts_t::wide overrides)
  const int wide_opt = m->opts[dtype].wide;
  const bool sim_wide = !witness && (wide_opt >= 0 ? wide_opt == 1 : (batch > 32768 && !(std::is_same<T, double>::value && NJ == 1)));
  if (witness)
    hipLaunchKernelGGL((simulate_kernel<T, NJ, true>), dim3((int)blocks), dim3(kWave), 0, stream, m->desc, m->opts[dtype],
                       (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x, batch, steps,
                       (T*)out, ld_item, ld_step, write_x0, (int*)iters, (const T*)witness);
  else if (copies == 2)
    hipLaunchKernelGGL((simulate_kernel<T, NJ, false, 2>), dim3((int)((batch + D::IPW / 2 - 1) / (D::IPW / 2))), dim3(kWave), 0, stream,
                       m->desc, m->opts[dtype], (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x,
                       batch, steps, (T*)out, ld_item, ld_step, write_x0, (int*)iters, (const T*)nullptr);
  else if (copies == 4) {
    if constexpr (D::G * 4 <= 16)
      hipLaunchKernelGGL((simulate_kernel<T, NJ, false, 4>), dim3((int)((batch + D::IPW / 4 - 1) / (D::IPW / 4))), dim3(kWave), 0, stream,
                         m->desc, m->opts[dtype], (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x,
                         batch, steps, (T*)out, ld_item, ld_step, write_x0, (int*)iters, (const T*)nullptr);
  } else if (sim_wide) {
    long long wb = (batch + kWave - 1) / kWave;
    if (wb > 8192) wb = 8192;
    if (wb <= kSimds)
      hipLaunchKernelGGL((simulate_kernel_wide<T, NJ, true>), dim3((int)wb), dim3(kWave), 0, stream, m->desc, m->opts[dtype],
                         (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x, batch, steps,
                         (T*)out, ld_item, ld_step, write_x0, (int*)iters);
    else
      hipLaunchKernelGGL((simulate_kernel_wide<T, NJ>), dim3((int)wb), dim3(kWave), 0, stream, m->desc, m->opts[dtype],
                         (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x, batch, steps,
                         (T*)out, ld_item, ld_step, write_x0, (int*)iters);
  } else if (blocks <= kSimds)
    hipLaunchKernelGGL((simulate_kernel<T, NJ, false, 1, true>), dim3((int)blocks), dim3(kWave), 0, stream, m->desc, m->opts[dtype],
                       (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x, batch, steps,
                       (T*)out, ld_item, ld_step, write_x0, (int*)iters, (const T*)nullptr);
  else
    hipLaunchKernelGGL((simulate_kernel<T, NJ, false>), dim3((int)blocks), dim3(kWave), 0, stream, m->desc, m->opts[dtype],
                       (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x, batch, steps,
                       (T*)out, ld_item, ld_step, write_x0, (int*)iters, (const T*)nullptr);
  return check_launch("simulate_kernel");
}

template <typename T, int NJ>
int launch_step_backward(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x,
                         const void* gx, long long ld_g, long long batch, void* grad, void* workspace,
                         long long workspace_bytes, hipStream_t stream, void* grad_x, long long ld_gx) {
  using D = Dims<T, NJ>;
  const int blocks = loss_blocks<T, NJ>(batch);
  if (!workspace || workspace_bytes < ((long long)blocks * D::PI + D::CHAIN) * (long long)sizeof(double))
    return fail(-3, "dpll_step_backward: workspace too small%s");
  if (grad_x)
    hipLaunchKernelGGL((step_backward_kernel<T, NJ, true>), dim3(blocks + 1), dim3(kWave), 0, stream, m->desc, m->opts[DPLL_F64],
                       (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)gx,
                       ld_g, batch, (double*)workspace, (T*)grad_x, ld_gx, (const T*)nullptr, (T*)nullptr);
  else
    hipLaunchKernelGGL((step_backward_kernel<T, NJ, false>), dim3(blocks + 1), dim3(kWave), 0, stream, m->desc, m->opts[DPLL_F64],
                       (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)gx,
                       ld_g, batch, (double*)workspace, (T*)nullptr, 0LL, (const T*)nullptr, (T*)nullptr);
  if (int rc = check_launch("step_backward_kernel")) return rc;
  hipLaunchKernelGGL((finalize_kernel<T, NJ, false>), dim3(1), dim3(kFinalizeThreads), 0, stream, (const double*)workspace, blocks,
                     (T*)grad, (T*)nullptr,
                     dpll_arx::Peers{}, 0, 1, (uint32_t*)nullptr, (uint32_t*)nullptr, AdamArgs{});
  return check_launch("finalize_kernel");
}

template <typename T, int NJ>
int launch_terms(const dpll_model* m, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm,
                 void* M, void* J, void* phi, void* a, hipStream_t stream, const void* witness = nullptr) {
  using D = Dims<T, NJ>;
  long long blocks = (batch + D::IPW - 1) / D::IPW;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((terms_kernel<T, NJ>), dim3((int)blocks), dim3(kWave), 0, stream, m->desc, (const T*)p->theta,
                     (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, batch, (T*)Dm, (T*)M, (T*)J,
                     (T*)phi, (T*)a, (const T*)witness);
  return check_launch("terms_kernel");
}


// ---- mesh (DeepSupportConvex) path: every body of a specialised build carries its own network; N = 4 * batch support
// queries per network and launch ------------------------------------------------------------------------------------
struct MeshPlan {
  long long N, n_tiles;
  int n_nets, loss_blocks, gemm_blocks, b1_blocks, n_slabs, row_stride;
  // one block of buffers per network (offsets relative to the block): the backward of a network needs its own forward
  size_t off_A, off_AT, off_Af, off_ATf, off_Ab, off_ATb, off_a, off_M1, off_U0, off_Vb, off_U1, off_b1, off_slabs, net_bytes;
  // shared: support points / their adjoints of all networks (batch, 4 n_nets, 3), the loss kernel's rows + chain matrix,
  // the bodies' world quaternions (batch, n_nets, 4; only for n_nets > 1)
  size_t off_P, off_RB, off_rows, off_bq, off_nets, total;
  char* net(char* ws, int g) const { return ws + off_nets + (size_t)g * net_bytes; }
};

// buffers of ONE network for N support queries (offsets relative to the network's block); the weight-preparation outputs
// (A, AT, Af, ATf, a) come first, at offsets that do not depend on N
template <typename T> void plan_network_block(MeshPlan& pl, long long N) {
  pl.N = N;
  constexpr bool kMfma = std::is_same<T, float>::value;  // float: MFMA kernels on 32-row tiles
  const long long tiles = (pl.N + (kMfma ? kMfmaRows : kTileRows) - 1) / (kMfma ? kMfmaRows : kTileRows);
  const long long cap = kMfma ? 256 : 2048;  // MFMA blocks keep their 256 x 32 weight block in registers: one per CU
  pl.gemm_blocks = (int)(tiles < cap ? tiles : cap);
  pl.b1_blocks = (int)(tiles < 256 ? tiles : 256);
  long long slabs = pl.N / (kMfma ? 256 : 1024);
  pl.n_slabs = (int)(slabs < 1 ? 1 : (slabs > 64 ? 64 : slabs));
  pl.n_tiles = tiles;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  pl.off_A = take(sizeof(T) * kW * kW);
  pl.off_AT = take(sizeof(T) * kW * kW);
  pl.off_Af = take(kMfma ? sizeof(T) * kW * kW : 0);   // the same two matrices in the MFMA kernels' fragment order
  pl.off_ATf = take(kMfma ? sizeof(T) * kW * kW : 0);
  pl.off_Ab = take(kMfma ? 2 * 3 * kW * kW : 0);       // ... and as up to three bf16 planes (dpll_mesh_bf16.hpp)
  pl.off_ATb = take(kMfma ? 2 * 3 * kW * kW : 0);
  pl.off_a = take(sizeof(T) * kW);
  pl.off_M1 = take(sizeof(uint32_t) * kMaskWords * pl.N);
  // (float: whole 32-row tiles + one spare -- the pipelined kernels (dpll_icnn_pipe.hip) keep U0 in the accumulator layout
  // of the MFMA and send the stores of tiles that do not exist to the spare one)
  pl.off_U0 = take(sizeof(T) * kW * (kMfma ? kMfmaRows * (tiles + 1) : pl.N));
  // MFMA path: Vb (icnn_bwd1) as operand tiles for icnn_bwd2, whole 32-row tiles + the spare one
  pl.off_Vb = take(kMfma ? sizeof(T) * kW * kMfmaRows * (tiles + 1) : 0);
  pl.off_U1 = take(0);  // (round 2 kept U1 operand tiles here)
  pl.off_b1 = take(sizeof(double) * kB1Cols * pl.b1_blocks);
  pl.off_slabs = take(sizeof(T) * kW * kW * pl.n_slabs);
  pl.net_bytes = off;
}

template <typename T, int NJ> MeshPlan mesh_plan(long long batch) {
  using D = Dims<T, NJ>;
  MeshPlan pl;
  pl.n_nets = D::NB;
  pl.row_stride = D::PI;
  pl.loss_blocks = loss_blocks<T, NJ>(batch);
  plan_network_block<T>(pl, 4 * batch);
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  pl.off_P = take(sizeof(T) * 3 * pl.N * pl.n_nets);
  pl.off_RB = take(sizeof(T) * 3 * pl.N * pl.n_nets);
  pl.off_rows = take(sizeof(double) * ((size_t)D::PI * pl.loss_blocks + D::CHAIN));
  pl.off_bq = take(pl.n_nets > 1 ? sizeof(T) * 4 * pl.n_nets * batch : 0);
  pl.off_nets = off;
  pl.total = off + pl.net_bytes * pl.n_nets;
  return pl;
}

// world <- body quaternions of every body of a serial chain: q_b = q_(b-1) (x) [cos(angle / 2), axis sin(angle / 2)]; the ICNN
// kernels take a body's support direction from "its" quaternion exactly as they do for a single body (the rotation of the
// product is the product of the rotations, also for the un-normalised base quaternion of quirk Q2)
template <typename T, int NJ>
__global__ __launch_bounds__(256) void mesh_body_quat_kernel(ModelDesc md, const T* __restrict__ x, long long ld, long long batch,
                                                             T* __restrict__ bq) {
  const long long it = (long long)blockIdx.x * 256 + threadIdx.x;
  if (it >= batch) return;
  T q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { q[i] = x[it * ld + i]; bq[(it * (NJ + 1)) * 4 + i] = q[i]; }
#pragma unroll
  for (int j = 1; j <= NJ; ++j) {
    T sn, cs;
    tsincos(x[it * ld + 7 + j - 1] * T(0.5), sn, cs);
    const T r[4] = {cs, T(md.joint_axis[j - 1][0]) * sn, T(md.joint_axis[j - 1][1]) * sn, T(md.joint_axis[j - 1][2]) * sn};
    const T o[4] = {q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3],
                    q[0] * r[1] + r[0] * q[1] + (q[2] * r[3] - q[3] * r[2]),
                    q[0] * r[2] + r[0] * q[2] + (q[3] * r[1] - q[1] * r[3]),
                    q[0] * r[3] + r[0] * q[3] + (q[1] * r[2] - q[2] * r[1])};
#pragma unroll
    for (int i = 0; i < 4; ++i) { q[i] = o[i]; bq[(it * (NJ + 1) + j) * 4 + i] = o[i]; }
  }
}

// measuring aid of dpll_profile_contactnets_loss_mesh: when set, an event is recorded on the launch stream after each
// kernel of the mesh pipeline (order: prep, fwd1, fwd2, loss, bwd1, bwd2, reduce)
constexpr int kMeshKernels = 7;
thread_local hipEvent_t* t_mesh_marks = nullptr;
thread_local int t_mesh_mark = 0;
inline void mesh_mark(hipStream_t stream) {
  if (t_mesh_marks && t_mesh_mark < kMeshKernels) (void)hipEventRecord(t_mesh_marks[t_mesh_mark++], stream);
}

// network g of a model with n_nets bodies: its points sit at (item, 4 g + s) of the shared (batch, 4 n_nets, 3) arrays
template <typename T> IcnnWeights<T> mesh_weights(const dpll_mesh_params_t* mp, int g, int n_nets) {
  IcnnWeights<T> w{(const T*)mp[g].hidden_weight, (const T*)mp[g].input_weight0, (const T*)mp[g].input_weight1,
                   (const T*)mp[g].output_weight, (const T*)mp[g].perturbations};
  w.point_stride = 12 * n_nets;
  return w;
}

// where network g's kernels read "their" quaternion: the state itself for a single body, else the body-quaternion buffer
template <typename T> struct QuatSource { const T* ptr; long long ld; };
template <typename T, int NJ>
QuatSource<T> quat_source(const MeshPlan& pl, char* ws, const T* state, long long ld, int g) {
  if (NJ == 0) return QuatSource<T>{state, ld};
  return QuatSource<T>{(const T*)(ws + pl.off_bq) + 4 * g, 4LL * (NJ + 1)};
}

template <typename T, int NJ>
int mesh_body_quats(const dpll_model* m, const MeshPlan& pl, char* ws, const T* state, long long ld, long long batch, hipStream_t stream) {
  if (NJ == 0) return 0;
  hipLaunchKernelGGL((mesh_body_quat_kernel<T, NJ>), dim3((int)((batch + 255) / 256)), dim3(256), 0, stream, m->desc, state, ld, batch,
                     (T*)(ws + pl.off_bq));
  return check_launch("mesh_body_quat_kernel");
}

// which form of the float GEMM kernels a call uses: 0 = v_mfma_f32_32x32x2_f32 (exact f32, the default), 2 / 3 = bf16 matrix
// cores on operands split into 2 / 3 bf16 planes (dpll_solver_opts_t.mesh_gemm); set by the C entry points
thread_local int t_mesh_gemm = 0;

// forward half of network g: prep + the two forward GEMMs -> its support points (and M1, U0 for the backward half)
template <typename T>
int mesh_forward(const MeshPlan& pl, int g, const IcnnWeights<T>& w, char* ws, QuatSource<T> q, hipStream_t stream,
                 bool for_backward = false, bool prep = true, T* points = nullptr) {
  char* nb = pl.net(ws, g);
  T* A = (T*)(nb + pl.off_A); T* AT = (T*)(nb + pl.off_AT); T* a = (T*)(nb + pl.off_a);
  // (`points`: where this launch's support points go when it is not the shared (batch, 4 n_nets, 3) array)
  T* P = points ? points : (T*)(ws + pl.off_P) + 12 * g;
  constexpr bool kMfmaPath = std::is_same<T, float>::value;
  bool split_prep = false;  // (|W| in GEMM order: only the weights enter, so the steps of a rollout after the first skip it)
  if constexpr (std::is_same<T, float>::value) {
    split_prep = t_mesh_gemm >= 2;  // the bf16 forms read their own planes (and |wout|): nothing of the f32 layouts
    if (prep && t_mesh_gemm == 2)
      hipLaunchKernelGGL((icnn_prep_bf16_kernel<2>), dim3(kW * kW / 256), dim3(256), 0, stream, w, (__bf16*)(nb + pl.off_Ab), (__bf16*)(nb + pl.off_ATb), (float*)a);
    if (prep && t_mesh_gemm == 3)
      hipLaunchKernelGGL((icnn_prep_bf16_kernel<3>), dim3(kW * kW / 256), dim3(256), 0, stream, w, (__bf16*)(nb + pl.off_Ab), (__bf16*)(nb + pl.off_ATb), (float*)a);
    if (prep && t_mesh_gemm == 4)  // two fp16 planes, the low one scaled by 2^11: f32-grade products (dpll_mesh_bf16.hpp)
      hipLaunchKernelGGL((icnn_prep_bf16_kernel<2, true>), dim3(kW * kW / 256), dim3(256), 0, stream, w, (__bf16*)(nb + pl.off_Ab), (__bf16*)(nb + pl.off_ATb), (float*)a);
  }
  if (prep && !split_prep)
    hipLaunchKernelGGL((icnn_prep_kernel<T>), dim3(kW * kW / 256), dim3(256), 0, stream, w, A, AT, a,
                       kMfmaPath ? (T*)(nb + pl.off_Af) : (T*)nullptr, kMfmaPath ? (T*)(nb + pl.off_ATf) : (T*)nullptr);
  mesh_mark(stream);
  if constexpr (std::is_same<T, float>::value) {
    float* U1t = nullptr;  // (U1 is rebuilt from the mask words by the weight-gradient GEMM: nothing of it is stored)
    (void)for_backward;
#define DPLL_FWD_BF16(PL_)                                                                                                         \
    do {                                                                                                                          \
      hipLaunchKernelGGL((icnn_fwd1_bf16<PL_>), dim3(pl.gemm_blocks), dim3(512), 0, stream, q.ptr, q.ld, pl.N, w,                   \
                         (const __bf16*)(nb + pl.off_Ab), (uint32_t*)(nb + pl.off_M1));                                           \
      mesh_mark(stream);                                                                                                          \
      hipLaunchKernelGGL((icnn_fwd2_bf16<PL_>), dim3(pl.gemm_blocks), dim3(512), 0, stream, q.ptr, q.ld, pl.N, w,                   \
                         (const __bf16*)(nb + pl.off_ATb), (const float*)a, (const uint32_t*)(nb + pl.off_M1),                    \
                         (float*)(nb + pl.off_U0), (float*)P);                                                                    \
    } while (0)
    if (t_mesh_gemm == 2 || t_mesh_gemm == 4) {
      // 2 planes (bf16, or = 4 fp16): one wave per SIMD, pipelined (dpll_icnn_pipe.hip) -- on the 16-bit matrix cores the side
      // work hides behind the MFMAs
      const bool f16 = t_mesh_gemm == 4;
      if (int rc = dpll_pipe::fwd1_bf16(stream, q.ptr, q.ld, pl.N, w, nb + pl.off_Ab, (uint32_t*)(nb + pl.off_M1), f16)) return rc;
      mesh_mark(stream);
      if (int rc = dpll_pipe::fwd2_bf16(stream, q.ptr, q.ld, pl.N, w, nb + pl.off_ATb, (const float*)a, (const uint32_t*)(nb + pl.off_M1),
                                        (float*)(nb + pl.off_U0), (float*)P, f16)) return rc;
    }
    else if (t_mesh_gemm == 3) DPLL_FWD_BF16(3);
    else if (t_mesh_gemm == 0 && pl.n_tiles > 2 * dpll_pipe::kMaxBlocks) {
      // the default beyond two row tiles per CU: one wave per SIMD, software pipelined (dpll_icnn_pipe.hip)
      if (int rc = dpll_pipe::fwd1(stream, q.ptr, q.ld, pl.N, w, (const float*)(nb + pl.off_Af), (uint32_t*)(nb + pl.off_M1))) return rc;
      mesh_mark(stream);
      if (int rc = dpll_pipe::fwd2(stream, q.ptr, q.ld, pl.N, w, (const float*)(nb + pl.off_ATf), (const float*)a,
                                   (const uint32_t*)(nb + pl.off_M1), (float*)(nb + pl.off_U0), (float*)P)) return rc;
    } else {
      // up to two tiles per CU (the 4096-pair benchmark batch) the 8-wave kernels of rounds 1-4 are ~2 us per launch ahead: a
      // launch is then prologue-bound and eight waves start the 256 KB fragment load sooner; mesh_gemm = 1 runs
      // them at every size (A/B measurements).  With mesh_gemm = 0 U0 leaves in the layout the pipelined icnn_bwd1 reads.
      hipLaunchKernelGGL(icnn_fwd1_mfma, dim3(pl.gemm_blocks), dim3(512), 0, stream, q.ptr, q.ld, pl.N, w,
                         (const float*)(nb + pl.off_Af), (uint32_t*)(nb + pl.off_M1));
      mesh_mark(stream);
      if (t_mesh_gemm == 0) {  // (the pipelined fwd2 is ahead at every size since its row sums are a transposed butterfly)
        if (int rc = dpll_pipe::fwd2(stream, q.ptr, q.ld, pl.N, w, (const float*)(nb + pl.off_ATf), (const float*)a,
                                     (const uint32_t*)(nb + pl.off_M1), (float*)(nb + pl.off_U0), (float*)P)) return rc;
      } else
        hipLaunchKernelGGL(icnn_fwd2_mfma<false>, dim3(pl.gemm_blocks), dim3(512), 0, stream, q.ptr, q.ld, pl.N, w,
                           (const float*)(nb + pl.off_ATf), (const float*)a, (const uint32_t*)(nb + pl.off_M1), (float*)(nb + pl.off_U0),
                           (float*)P, U1t);
    }
#undef DPLL_FWD_BF16
  } else {
    hipLaunchKernelGGL((icnn_fwd1_kernel<T>), dim3(pl.gemm_blocks), dim3(256), 0, stream, q.ptr, q.ld, pl.N, w, (const T*)A,
                       (uint32_t*)(nb + pl.off_M1));
    mesh_mark(stream);
    hipLaunchKernelGGL((icnn_fwd2_kernel<T>), dim3(pl.gemm_blocks), dim3(256), 0, stream, q.ptr, q.ld, pl.N, w, (const T*)AT,
                       (const T*)a, (const uint32_t*)(nb + pl.off_M1), (T*)(nb + pl.off_U0), P);
  }
  mesh_mark(stream);
  return check_launch("icnn forward");
}

// backward half of network g, shared by the loss and by the step backward: r_bar (ws.RB) and the row partials (ws.rows)
// are in place; grad_w = this network's slice of the gradient, grad_head / loss_total only with the first network
template <typename T, int NB>
int mesh_backward(const MeshPlan& pl, int g, const IcnnWeights<T>& w, char* ws, QuatSource<T> q, T* grad_w, T* grad_head,
                  T* loss_total, hipStream_t stream, const T* adjoints = nullptr, const AdamArgs* adam = nullptr, long long w_offset = 0) {
  char* nb = pl.net(ws, g);
  const T* RB = adjoints ? adjoints : (const T*)(ws + pl.off_RB) + 12 * g;
  const int n_slabs = pl.n_slabs;
  if constexpr (std::is_same<T, float>::value) {
#define DPLL_BWD_BF16(PL_)                                                                                                         \
    do {                                                                                                                          \
      hipLaunchKernelGGL((icnn_bwd1_bf16<PL_>), dim3(pl.b1_blocks), dim3(512), 0, stream, (const float*)q.ptr, q.ld, pl.N, w,       \
                         (const __bf16*)(nb + pl.off_Ab), (const float*)(nb + pl.off_a), (const uint32_t*)(nb + pl.off_M1),       \
                         (const float*)(nb + pl.off_U0), (const float*)RB, (double*)(nb + pl.off_b1), (float*)(nb + pl.off_Vb));  \
      mesh_mark(stream);                                                                                                          \
      hipLaunchKernelGGL((icnn_bwd2_bf16<PL_>), dim3(kB2Pieces, pl.n_slabs), dim3(512), 0, stream, pl.N,                           \
                         (const float*)(nb + pl.off_Vb), (const uint32_t*)(nb + pl.off_M1), (const float*)(nb + pl.off_a),        \
                         (float*)(nb + pl.off_slabs), (const unsigned*)nullptr);                                                  \
    } while (0)
    if (t_mesh_gemm == 2 || t_mesh_gemm == 4) {
      const bool f16 = t_mesh_gemm == 4;
      // (fp16 planes: the word behind the two weight planes of Ab -- zeroed by the prep kernel -- carries the launch's largest |r_bar|
      // from icnn_bwd1 to icnn_bwd2)
      unsigned* rbmax = f16 ? reinterpret_cast<unsigned*>(nb + pl.off_Ab + (size_t)2 * 2 * kW * kW) : nullptr;
      if (int rc = dpll_pipe::bwd1_bf16(stream, (const float*)q.ptr, q.ld, pl.N, w, nb + pl.off_Ab, (const float*)(nb + pl.off_a),
                                        (const uint32_t*)(nb + pl.off_M1), (const float*)(nb + pl.off_U0), (const float*)RB,
                                        (double*)(nb + pl.off_b1), (float*)(nb + pl.off_Vb), f16, rbmax)) return rc;
      mesh_mark(stream);
      if (f16)
        hipLaunchKernelGGL((icnn_bwd2_bf16<2, true>), dim3(kB2Pieces, pl.n_slabs), dim3(512), 0, stream, pl.N,
                           (const float*)(nb + pl.off_Vb), (const uint32_t*)(nb + pl.off_M1), (const float*)(nb + pl.off_a),
                           (float*)(nb + pl.off_slabs), (const unsigned*)rbmax);
      else
        hipLaunchKernelGGL((icnn_bwd2_bf16<2>), dim3(kB2Pieces, pl.n_slabs), dim3(512), 0, stream, pl.N,
                           (const float*)(nb + pl.off_Vb), (const uint32_t*)(nb + pl.off_M1), (const float*)(nb + pl.off_a),
                           (float*)(nb + pl.off_slabs), (const unsigned*)nullptr);
    }
    else if (t_mesh_gemm == 3) DPLL_BWD_BF16(3);
    else if (t_mesh_gemm == 0) {
      if (int rc = dpll_pipe::bwd1(stream, (const float*)q.ptr, q.ld, pl.N, w, (const float*)(nb + pl.off_Af), (const float*)(nb + pl.off_a),
                                   (const uint32_t*)(nb + pl.off_M1), (const float*)(nb + pl.off_U0), (const float*)RB,
                                   (double*)(nb + pl.off_b1), (float*)(nb + pl.off_Vb))) return rc;
      mesh_mark(stream);
      // (the weight-gradient GEMM stays with the 8-wave kernel: three pipelined versions measured no better, dpll_icnn_pipe.hip)
      hipLaunchKernelGGL(icnn_bwd2_mfma, dim3(kB2Pieces, pl.n_slabs), dim3(512), 0, stream, pl.N,
                         (const float*)(nb + pl.off_Vb), (const uint32_t*)(nb + pl.off_M1), (const float*)(nb + pl.off_a),
                         (float*)(nb + pl.off_slabs));
    } else {
      hipLaunchKernelGGL(icnn_bwd1_mfma, dim3(pl.b1_blocks), dim3(512), 0, stream, (const float*)q.ptr, q.ld, pl.N, w,
                         (const float*)(nb + pl.off_Af), (const float*)(nb + pl.off_a), (const uint32_t*)(nb + pl.off_M1),
                         (const float*)(nb + pl.off_U0), (const float*)RB, (double*)(nb + pl.off_b1),
                         (float*)(nb + pl.off_Vb));
      mesh_mark(stream);
      hipLaunchKernelGGL(icnn_bwd2_mfma, dim3(kB2Pieces, pl.n_slabs), dim3(512), 0, stream, pl.N,
                         (const float*)(nb + pl.off_Vb), (const uint32_t*)(nb + pl.off_M1), (const float*)(nb + pl.off_a),
                         (float*)(nb + pl.off_slabs));
    }
#undef DPLL_BWD_BF16
  } else {
    hipLaunchKernelGGL((icnn_bwd1_kernel<T>), dim3(pl.b1_blocks), dim3(256), 0, stream, (const T*)q.ptr, q.ld, pl.N, w,
                       (const T*)(nb + pl.off_A), (const T*)(nb + pl.off_a), (const uint32_t*)(nb + pl.off_M1),
                       (const T*)(nb + pl.off_U0), RB, (double*)(nb + pl.off_b1));
    mesh_mark(stream);
    hipLaunchKernelGGL((icnn_bwd2_kernel<T>), dim3(16, pl.n_slabs), dim3(256), 0, stream, (const T*)q.ptr, q.ld, pl.N, w,
                       (const T*)(nb + pl.off_a), (const uint32_t*)(nb + pl.off_M1), RB, (T*)(nb + pl.off_slabs));
  }
  mesh_mark(stream);
  hipLaunchKernelGGL((icnn_reduce_kernel<T, NB>), dim3(kRedBlocks), dim3(256), 0, stream, w,
                     (const double*)(ws + pl.off_rows), pl.loss_blocks, pl.row_stride, (const double*)(nb + pl.off_b1), pl.b1_blocks,
                     (const T*)(nb + pl.off_slabs), n_slabs, grad_w, grad_head, loss_total, adam ? *adam : AdamArgs{}, w_offset);
  mesh_mark(stream);
  return check_launch("icnn backward");
}

constexpr int kNetParams = kW * kW + 7 * kW;  // [Wh | Wd0 | Wd1 | wout] of one network

template <typename T, int NJ>
int mesh_backward_all(const MeshPlan& pl, const dpll_mesh_params_t* mp, char* ws, const T* state, long long ld, void* grad,
                      void* loss_total, hipStream_t stream, const AdamArgs* adam = nullptr) {
  constexpr int NB = NJ + 1, kHead = 10 * NB + 1 + NB;
  for (int g = 0; g < NB; ++g)
    if (int rc = mesh_backward<T, NB>(pl, g, mesh_weights<T>(mp, g, NB), ws, quat_source<T, NJ>(pl, ws, state, ld, g),
                                      (T*)grad + kHead + (size_t)g * kNetParams, g == 0 ? (T*)grad : (T*)nullptr,
                                      g == 0 ? (T*)loss_total : (T*)nullptr, stream, nullptr, adam, kHead + (long long)g * kNetParams))
      return rc;
  if (adam) {  // (every reduce launch has read the optimizer state: one thread moves it a step on)
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, stream, *adam);
    return check_launch("adam_advance_kernel");
  }
  return 0;
}

template <typename T, int NJ>
int mesh_forward_all(const dpll_model* m, const MeshPlan& pl, const dpll_mesh_params_t* mp, char* ws, const T* state,
                     long long ld, long long batch, hipStream_t stream, bool for_backward, bool prep = true) {
  constexpr int NB = NJ + 1;
  if (int rc = mesh_body_quats<T, NJ>(m, pl, ws, state, ld, batch, stream)) return rc;
  for (int g = 0; g < NB; ++g)
    if (int rc = mesh_forward<T>(pl, g, mesh_weights<T>(mp, g, NB), ws, quat_source<T, NJ>(pl, ws, state, ld, g), stream, for_backward,
                                 prep))
      return rc;
  return 0;
}

// Does the loss launch of dpll_contactnets_loss_mesh run four racing copies per item?  ONE predicate for the launch and for
// dpll_racing_copies(what = 4) (the two used to be written out separately and could disagree): a single body, a
// launch the race shape gives four copies of a full lane group, and the same number of partial rows as the plain launch (the
// reduce kernel's row count is fixed by the plan)
template <typename T, int NJ> bool mesh_loss_races(const dpll_model* m, int dtype, long long batch) {
  if constexpr (NJ != 0) return false;
  const RaceShape shape = race_shape<T, NJ>(m, dtype, batch);
  return shape.copies == 4 && shape.lanes == Dims<T, NJ>::G && loss_blocks<T, NJ>(batch, 4, shape.lanes) == loss_blocks<T, NJ>(batch);
}

template <typename T, int NJ>
int launch_mesh_loss(const dpll_model* m, int dtype, const dpll_params_t* p, const dpll_mesh_params_t* mp, const void* x,
                     long long ld_x, const void* xp, long long ld_xp, long long batch, const void* weights, double scale,
                     void* loss, void* grad, void* loss_total, void* force, int32_t* iters, void* workspace,
                     long long workspace_bytes, hipStream_t stream, const AdamArgs* adam = nullptr) {
  const MeshPlan pl = mesh_plan<T, NJ>(batch);
  if (!workspace || (size_t)workspace_bytes < pl.total) return fail(-3, "dpll_contactnets_loss_mesh: workspace too small%s");
  if (!grad && loss_total) return fail(-3, "dpll_contactnets_loss_mesh: loss_total requires grad%s");
  char* ws = (char*)workspace;
  const int want_grad = grad != nullptr;
  // terms live at the NEXT state
  if (int rc = mesh_forward_all<T, NJ>(m, pl, mp, ws, (const T*)xp, ld_xp, batch, stream, want_grad != 0)) return rc;
  // (the float build claims its SIMD, which serves launches of at most one wave per SIMD; beyond -- more than 16,384 cube or
  // 8192 elbow pairs -- the float launch runs the build for two waves per SIMD, as the box path does; the double kernels
  // allocate more than half the register file by themselves and never claim)
#define DPLL_MESH_LOSS(DENSE_)                                                                                                   \
  hipLaunchKernelGGL((loss_kernel<T, NJ, true, DENSE_>), dim3(pl.loss_blocks + 1), dim3(kWave), 0, stream, (const T*)x, ld_x,     \
                     (const T*)xp, ld_xp, batch, (const T*)p->theta, (const T*)p->friction, (const T*)nullptr,                    \
                     (const T*)weights, scale, (T*)loss, (T*)force, (int*)iters, (double*)(ws + pl.off_rows),                     \
                     want_grad, (const T*)(ws + pl.off_P), want_grad ? (T*)(ws + pl.off_RB) : (T*)nullptr, m->desc, m->opts[dtype])
  // Racing copies (round 4; what the box cube has run since round 3): a launch of at most 4096 cube pairs gives every item
  // four copies of its cone solve on the SIMDs a 256-wave launch leaves idle -- the copies read the same support points, the
  // winner writes the witness adjoints; four-wave workgroups share one partial row, so the reduce kernel sums as many rows
  bool raced = false;
  if constexpr (NJ == 0) {
    if (mesh_loss_races<T, NJ>(m, dtype, batch)) {
      hipLaunchKernelGGL((loss_kernel<T, NJ, true, false, 4>), dim3(pl.loss_blocks + 1), dim3(kRaceWaves * kWave), 0, stream, (const T*)x, ld_x,
                         (const T*)xp, ld_xp, batch, (const T*)p->theta, (const T*)p->friction, (const T*)nullptr, (const T*)weights, scale,
                         (T*)loss, (T*)force, (int*)iters, (double*)(ws + pl.off_rows), want_grad, (const T*)(ws + pl.off_P),
                         want_grad ? (T*)(ws + pl.off_RB) : (T*)nullptr, m->desc, m->opts[dtype]);
      raced = true;
    }
  }
  if (!raced) {
    if constexpr (std::is_same<T, float>::value) {
      if (pl.loss_blocks > kSimds) DPLL_MESH_LOSS(true);
      else DPLL_MESH_LOSS(false);
    } else {
      DPLL_MESH_LOSS(false);
    }
  }
#undef DPLL_MESH_LOSS
  mesh_mark(stream);
  if (int rc = check_launch("loss_kernel (mesh)")) return rc;
  if (!want_grad) return 0;
  return mesh_backward_all<T, NJ>(pl, mp, ws, (const T*)xp, ld_xp, grad, loss_total, stream, adam);
}

// Integrator.simulate with the network shapes: per step the two forward GEMMs of every network on the current state
// (the support points depend on it) and the one-step kernel, enqueued back to back; the weights are prepared once.
template <typename T, int NJ>
int launch_mesh_simulate(const dpll_model* m, int dtype, const dpll_params_t* p, const dpll_mesh_params_t* mp, const void* x0,
                         long long ld_x, long long batch, long long steps, void* traj, void* workspace,
                         long long workspace_bytes, hipStream_t stream) {
  using D = Dims<T, NJ>;
  const MeshPlan pl = mesh_plan<T, NJ>(batch);
  if (!workspace || (size_t)workspace_bytes < pl.total) return fail(-3, "dpll_simulate_mesh: workspace too small%s");
  char* ws = (char*)workspace;
  dpll_params_t q = *p;
  q.lengths = nullptr;
  const long long ld_item = (steps + 1) * D::NX;
  for (long long s = 0; s < steps; ++s) {
    const T* state = s == 0 ? (const T*)x0 : (const T*)traj + s * D::NX;
    const long long ld = s == 0 ? ld_x : ld_item;
    if (int rc = mesh_forward_all<T, NJ>(m, pl, mp, ws, state, ld, batch, stream, false, s == 0)) return rc;
    // the first step also writes x0 into traj[:, 0]
    if (int rc = launch_simulate<T, NJ>(m, dtype, &q, state, ld, batch, 1, (T*)traj + (s == 0 ? 0 : (s + 1) * D::NX), ld_item, D::NX,
                                        s == 0 ? 1 : 0, nullptr, stream, ws + pl.off_P))
      return rc;
  }
  return 0;
}

template <typename T, int NJ>
int launch_mesh_step(const dpll_model* m, int dtype, const dpll_params_t* p, const dpll_mesh_params_t* mp, const void* x,
                     long long ld_x, long long batch, void* x_next, long long ld_next, void* workspace,
                     long long workspace_bytes, hipStream_t stream) {
  const MeshPlan pl = mesh_plan<T, NJ>(batch);
  if (!workspace || (size_t)workspace_bytes < pl.total) return fail(-3, "dpll_step_mesh: workspace too small%s");
  char* ws = (char*)workspace;
  if (int rc = mesh_forward_all<T, NJ>(m, pl, mp, ws, (const T*)x, ld_x, batch, stream, false)) return rc;
  dpll_params_t q = *p;
  q.lengths = nullptr;
  return launch_simulate<T, NJ>(m, dtype, &q, x, ld_x, batch, 1, x_next, ld_next, 0, 0, nullptr, stream, ws + pl.off_P);
}

// backward of dpll_step_mesh: support points at x, step backward (emits r_bar and the theta / friction row partials),
// then the ICNN backward kernels as for the loss
template <typename T, int NJ>
int launch_mesh_step_backward(const dpll_model* m, int dtype, const dpll_params_t* p, const dpll_mesh_params_t* mp,
                              const void* x, long long ld_x, const void* gx, long long ld_g, long long batch, void* grad,
                              void* grad_x, long long ld_gx, void* workspace, long long workspace_bytes, hipStream_t stream) {
  const MeshPlan pl = mesh_plan<T, NJ>(batch);
  if (!workspace || (size_t)workspace_bytes < pl.total) return fail(-3, "dpll_step_backward_mesh: workspace too small%s");
  char* ws = (char*)workspace;
  if (int rc = mesh_forward_all<T, NJ>(m, pl, mp, ws, (const T*)x, ld_x, batch, stream, true)) return rc;
#define DPLL_LAUNCH_SB(STATE_)                                                                                             \
  hipLaunchKernelGGL((step_backward_kernel<T, NJ, STATE_, true>), dim3(pl.loss_blocks + 1), dim3(kWave), 0, stream, m->desc,   \
                     m->opts[DPLL_F64], (const T*)p->theta, (const T*)p->friction, (const T*)nullptr, (const T*)x, ld_x,      \
                     (const T*)gx, ld_g, batch, (double*)(ws + pl.off_rows), (T*)grad_x, ld_gx, (const T*)(ws + pl.off_P), \
                     (T*)(ws + pl.off_RB))
  if (grad_x) DPLL_LAUNCH_SB(true);
  else DPLL_LAUNCH_SB(false);
#undef DPLL_LAUNCH_SB
  if (int rc = check_launch("step_backward_kernel (mesh)")) return rc;
  return mesh_backward_all<T, NJ>(pl, mp, ws, (const T*)x, ld_x, grad, nullptr, stream);
}

template <typename T, int NJ>
int launch_mesh_support(const dpll_model* m, const dpll_mesh_params_t* mp, const void* x, long long ld_x, long long batch, void* points,
                        void* workspace, long long workspace_bytes, hipStream_t stream) {
  const MeshPlan pl = mesh_plan<T, NJ>(batch);
  if (!workspace || (size_t)workspace_bytes < pl.total) return fail(-3, "dpll_mesh_support_points: workspace too small%s");
  char* ws = (char*)workspace;
  if (int rc = mesh_forward_all<T, NJ>(m, pl, mp, ws, (const T*)x, ld_x, batch, stream, false)) return rc;
  if (hipMemcpyAsync(points, ws + pl.off_P, sizeof(T) * 3 * pl.N * pl.n_nets, hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return fail(-5, "dpll_mesh_support_points: copy failed%s");
  return 0;
}

template <typename T, int NJ>
int launch_mesh_terms(const dpll_model* m, const dpll_params_t* p, const dpll_mesh_params_t* mp, const void* x, long long ld_x,
                      long long batch, void* Dm, void* M, void* J, void* phi, void* a, void* workspace, long long workspace_bytes,
                      hipStream_t stream) {
  const MeshPlan pl = mesh_plan<T, NJ>(batch);
  if (!workspace || (size_t)workspace_bytes < pl.total) return fail(-3, "dpll_terms_mesh: workspace too small%s");
  char* ws = (char*)workspace;
  if (int rc = mesh_forward_all<T, NJ>(m, pl, mp, ws, (const T*)x, ld_x, batch, stream, false)) return rc;
  dpll_params_t q = *p;
  q.lengths = nullptr;
  return launch_terms<T, NJ>(m, &q, x, ld_x, batch, Dm, M, J, phi, a, stream, ws + pl.off_P);
}

// ---- the general build with learned shapes (csrc/dpll_genmesh.hip): any tree, DeepSupportConvex next to other geometries,
// body-body candidates between two learned shapes (GeometryCollider.collide_mesh_mesh, geometry.py:585-643) -------------
struct GenMeshPlan {
  int n_mesh;                  // learned geometries, in geometry order
  int geom_of[kMaxGeoms];      // k-th learned geometry -> geometry index
  int net_of[kMaxGeoms];       // geometry index -> k (or -1)
  int qpi[kMaxGeoms];          // by geometry index: 4 ground queries + one per candidate the geometry is part of
  int query_a[kMaxPairs], query_b[kMaxPairs];
  int qoff[kMaxGeoms][dpll_genmesh::kMaxQueries];
  long long max_N;
  size_t net_bytes;            // block of one network, sized for the largest query count
  size_t off_W, off_RB, off_PD, off_rows, off_surf, off_hull[kMaxGeoms], off_dirs[kMaxGeoms], off_nets, total;
};

template <typename T> GenMeshPlan genmesh_plan(const dpll_model* m, long long batch) {
  const ModelDesc& md = m->desc;
  GenMeshPlan gp;
  std::memset(&gp, 0, sizeof(gp));
  for (int g = 0; g < kMaxGeoms; ++g) {
    gp.net_of[g] = -1;
    if (g < md.n_geoms && md.geom_kind[g] == kGeomMesh) {
      gp.net_of[g] = gp.n_mesh;
      gp.geom_of[gp.n_mesh++] = g;
      gp.qpi[g] = kQuery;
      for (int s = 0; s < kQuery; ++s) gp.qoff[g][s] = dpll_genmesh::wit_offset(kQuery * g + s, 0);
    }
  }
  for (int p = 0; p < md.n_pairs && p < kMaxPairs; ++p) {
    const int a = md.pair_a[p], b = md.pair_b[p];
    if (md.geom_kind[a] != kGeomMesh || md.geom_kind[b] != kGeomMesh) continue;
    gp.query_a[p] = gp.qpi[a];
    gp.qoff[a][gp.qpi[a]++] = dpll_genmesh::wit_offset(kQuery * kMaxGeoms + p, 1);  // A's support point along d
    gp.query_b[p] = gp.qpi[b];
    gp.qoff[b][gp.qpi[b]++] = dpll_genmesh::wit_offset(kQuery * kMaxGeoms + p, 0);  // B's along -d
  }
  gp.max_N = kHullDirs;
  for (int g = 0; g < kMaxGeoms; ++g)
    if (gp.qpi[g] * batch > gp.max_N) gp.max_N = gp.qpi[g] * batch;
  MeshPlan block;
  plan_network_block<T>(block, gp.max_N);
  gp.net_bytes = block.net_bytes;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  const size_t per_item = (size_t)dpll_genmesh::wit_per_item();
  gp.off_W = take(sizeof(T) * per_item * batch);
  gp.off_RB = take(sizeof(T) * per_item * batch);
  gp.off_PD = take(sizeof(double) * 3 * kMaxPairs * batch);
  gp.off_rows = take((size_t)dpll_genmesh::workspace_bytes(m, batch));
  gp.off_surf = take(sizeof(T) * 3 * kHullDirs);
  for (int g = 0; g < kMaxGeoms; ++g) {
    gp.off_hull[g] = take(gp.qpi[g] ? sizeof(T) * 3 * kHullDirs : 0);
    gp.off_dirs[g] = take(sizeof(T) * 3 * gp.qpi[g] * batch);
  }
  gp.off_nets = off;
  gp.total = off + gp.net_bytes * gp.n_mesh;
  return gp;
}

// the buffer plan of network k for N queries, placed in the general layout
template <typename T> MeshPlan genmesh_block(const GenMeshPlan& gp, long long N) {
  MeshPlan pl;
  std::memset(&pl, 0, sizeof(pl));
  plan_network_block<T>(pl, N);
  pl.net_bytes = gp.net_bytes;
  pl.off_nets = gp.off_nets;
  pl.n_nets = gp.n_mesh;
  return pl;
}

template <typename T> IcnnWeights<T> genmesh_weights(const GenMeshPlan& gp, const dpll_mesh_params_t* mp, int g, char* ws) {
  IcnnWeights<T> w{(const T*)mp[g].hidden_weight, (const T*)mp[g].input_weight0, (const T*)mp[g].input_weight1,
                   (const T*)mp[g].output_weight, (const T*)mp[g].perturbations};
  w.dirs = (const T*)(ws + gp.off_dirs[g]);
  w.qpi = gp.qpi[g];
  w.point_stride = dpll_genmesh::wit_per_item();
  for (int j = 0; j < dpll_genmesh::kMaxQueries; ++j) w.qoff[j] = gp.qoff[g][j];
  return w;
}

// once per parameter set: |W| in GEMM order and the vertex set of every network (its support points over the reference's
// 296 surface directions: the mesh extract_mesh builds for fcl, geometry.py:343-358)
template <typename T>
int genmesh_hulls(const dpll_model* m, int dtype, const GenMeshPlan& gp, const dpll_mesh_params_t* mp, char* ws, hipStream_t stream) {
  bool any_pair = false;
  for (int p = 0; p < m->desc.n_pairs; ++p) any_pair = any_pair || m->desc.geom_kind[m->desc.pair_a[p]] == kGeomMesh;
  if (any_pair)
    if (int rc = dpll_genmesh::surface_directions(dtype, ws + gp.off_surf, stream)) return rc;
  const MeshPlan p296 = genmesh_block<T>(gp, kHullDirs);
  for (int k = 0; k < gp.n_mesh; ++k) {
    const int g = gp.geom_of[k];
    IcnnWeights<T> w = genmesh_weights<T>(gp, mp, g, ws);
    w.dirs = (const T*)(ws + gp.off_surf);
    w.qpi = 1;
    w.point_stride = 3;
    w.qoff[0] = 0;
    if (any_pair) {
      if (int rc = mesh_forward<T>(p296, k, w, ws, QuatSource<T>{nullptr, 0}, stream, false, true, (T*)(ws + gp.off_hull[g]))) return rc;
    } else {  // no vertex set needed: the weights only
      char* nb = p296.net(ws, k);
      constexpr bool kMfmaPath = std::is_same<T, float>::value;
      hipLaunchKernelGGL((icnn_prep_kernel<T>), dim3(kW * kW / 256), dim3(256), 0, stream, w, (T*)(nb + p296.off_A), (T*)(nb + p296.off_AT),
                         (T*)(nb + p296.off_a), kMfmaPath ? (T*)(nb + p296.off_Af) : (T*)nullptr, kMfmaPath ? (T*)(nb + p296.off_ATf) : (T*)nullptr);
      if constexpr (kMfmaPath) {
        if (t_mesh_gemm == 2)
          hipLaunchKernelGGL((icnn_prep_bf16_kernel<2>), dim3(kW * kW / 256), dim3(256), 0, stream, w, (__bf16*)(nb + p296.off_Ab), (__bf16*)(nb + p296.off_ATb), (float*)(nb + p296.off_a));
        if (t_mesh_gemm == 3)
          hipLaunchKernelGGL((icnn_prep_bf16_kernel<3>), dim3(kW * kW / 256), dim3(256), 0, stream, w, (__bf16*)(nb + p296.off_Ab), (__bf16*)(nb + p296.off_ATb), (float*)(nb + p296.off_a));
        if (t_mesh_gemm == 4)
          hipLaunchKernelGGL((icnn_prep_bf16_kernel<2, true>), dim3(kW * kW / 256), dim3(256), 0, stream, w, (__bf16*)(nb + p296.off_Ab), (__bf16*)(nb + p296.off_ATb), (float*)(nb + p296.off_a));
      }
      if (int rc = check_launch("icnn_prep_kernel")) return rc;
    }
  }
  return 0;
}

// per state: the networks' queries (ground directions; GJK / EPA direction of the candidates between learned shapes) and
// the support points at them, into the (slot, side) witness layout
template <typename T>
int genmesh_support(const dpll_model* m, int dtype, const GenMeshPlan& gp, const dpll_mesh_params_t* mp, char* ws, const void* state,
                    long long ld, long long batch, hipStream_t stream, bool for_backward) {
  dpll_genmesh::QueryPlan qp;
  std::memset(&qp, 0, sizeof(qp));
  for (int g = 0; g < kMaxGeoms; ++g) {
    if (gp.net_of[g] < 0) continue;
    qp.pert[g] = mp[g].perturbations;
    qp.dirs[g] = ws + gp.off_dirs[g];
    qp.hull[g] = ws + gp.off_hull[g];
    qp.qpi[g] = gp.qpi[g];
  }
  for (int p = 0; p < kMaxPairs; ++p) { qp.query_a[p] = gp.query_a[p]; qp.query_b[p] = gp.query_b[p]; }
  qp.pdirs = (double*)(ws + gp.off_PD);
  if (int rc = dpll_genmesh::queries(m, dtype, qp, state, ld, batch, stream)) return rc;
  for (int k = 0; k < gp.n_mesh; ++k) {
    const int g = gp.geom_of[k];
    const MeshPlan pl = genmesh_block<T>(gp, gp.qpi[g] * batch);
    if (int rc = mesh_forward<T>(pl, k, genmesh_weights<T>(gp, mp, g, ws), ws, QuatSource<T>{nullptr, 0}, stream, for_backward, false,
                                 (T*)(ws + gp.off_W)))
      return rc;
  }
  return 0;
}

template <typename T>
int launch_genmesh_loss(const dpll_model* m, int dtype, const dpll_params_t* p, const dpll_mesh_params_t* mp, const void* x,
                        long long ld_x, const void* xp, long long ld_xp, long long batch, const void* weights, double scale,
                        void* loss, void* grad, void* loss_total, void* force, int32_t* iters, void* workspace,
                        long long workspace_bytes, hipStream_t stream, const AdamArgs* adam = nullptr) {
  const GenMeshPlan gp = genmesh_plan<T>(m, batch);
  if (!workspace || (size_t)workspace_bytes < gp.total) return fail(-3, "dpll_contactnets_loss_mesh: workspace too small%s");
  if (!grad && loss_total) return fail(-3, "dpll_contactnets_loss_mesh: loss_total requires grad%s");
  char* ws = (char*)workspace;
  const int want_grad = grad != nullptr;
  if (int rc = genmesh_hulls<T>(m, dtype, gp, mp, ws, stream)) return rc;
  if (int rc = genmesh_support<T>(m, dtype, gp, mp, ws, xp, ld_xp, batch, stream, want_grad != 0)) return rc;  // terms at the NEXT state
  if (int rc = dpll_genmesh::loss_items(m, dtype, p, x, ld_x, xp, ld_xp, batch, weights, scale, loss, force, iters, ws + gp.off_rows,
                                        want_grad, ws + gp.off_W, want_grad ? ws + gp.off_RB : nullptr,
                                        (const double*)(ws + gp.off_PD), stream))
    return rc;
  if (!want_grad) return 0;
  const int head = dpll_general::param_count(m);
  for (int k = 0; k < gp.n_mesh; ++k) {
    const int g = gp.geom_of[k];
    const MeshPlan pl = genmesh_block<T>(gp, gp.qpi[g] * batch);
    if (int rc = mesh_backward<T, 1>(pl, k, genmesh_weights<T>(gp, mp, g, ws), ws, QuatSource<T>{nullptr, 0},
                                     (T*)grad + head + (size_t)k * kNetParams, (T*)nullptr, (T*)nullptr, stream,
                                     (const T*)(ws + gp.off_RB), adam, head + (long long)k * kNetParams))
      return rc;
  }
  // (fused training step: every network's reduce kernel has applied Adam to the weights it owns, reading the optimizer state;
  // the finalize kernel -- the last launch -- applies it to the head and moves the state a step on)
  return dpll_genmesh::finalize(m, dtype, batch, ws + gp.off_rows, grad, loss_total, stream, adam);
}

// backward of dpll_step_mesh for a general model with learned shapes: support points at x, the step's backward (rows for the
// head of the gradient, the state adjoint, the witness adjoints), then the networks' backward kernels
template <typename T>
int launch_genmesh_step_backward(const dpll_model* m, int dtype, const dpll_params_t* p, const dpll_mesh_params_t* mp, const void* x,
                                 long long ld_x, const void* gx, long long ld_g, long long batch, void* grad, void* grad_x, long long ld_gx,
                                 void* workspace, long long workspace_bytes, hipStream_t stream) {
  const GenMeshPlan gp = genmesh_plan<T>(m, batch);
  if (!workspace || (size_t)workspace_bytes < gp.total) return fail(-3, "dpll_step_backward_mesh: workspace too small%s");
  char* ws = (char*)workspace;
  if (int rc = genmesh_hulls<T>(m, dtype, gp, mp, ws, stream)) return rc;
  if (int rc = genmesh_support<T>(m, dtype, gp, mp, ws, x, ld_x, batch, stream, true)) return rc;
  if (int rc = dpll_genmesh::step_backward_items(m, dtype, p, x, ld_x, gx, ld_g, batch, ws + gp.off_rows, grad_x, ld_gx, ws + gp.off_W,
                                                 ws + gp.off_RB, (const double*)(ws + gp.off_PD), stream))
    return rc;
  const int head = dpll_general::param_count(m);
  for (int k = 0; k < gp.n_mesh; ++k) {
    const int g = gp.geom_of[k];
    const MeshPlan pl = genmesh_block<T>(gp, gp.qpi[g] * batch);
    if (int rc = mesh_backward<T, 1>(pl, k, genmesh_weights<T>(gp, mp, g, ws), ws, QuatSource<T>{nullptr, 0},
                                     (T*)grad + head + (size_t)k * kNetParams, (T*)nullptr, (T*)nullptr, stream,
                                     (const T*)(ws + gp.off_RB)))
      return rc;
  }
  return dpll_genmesh::finalize(m, dtype, batch, ws + gp.off_rows, grad, nullptr, stream);
}

template <typename T>
int launch_genmesh_simulate(const dpll_model* m, int dtype, const dpll_params_t* p, const dpll_mesh_params_t* mp, const void* x0,
                            long long ld_x, long long batch, long long steps, void* traj, long long ld_item, long long ld_step,
                            bool write_x0, void* workspace, long long workspace_bytes, hipStream_t stream) {
  const GenMeshPlan gp = genmesh_plan<T>(m, batch);
  if (!workspace || (size_t)workspace_bytes < gp.total) return fail(-3, "dpll_step_mesh / dpll_simulate_mesh: workspace too small%s");
  char* ws = (char*)workspace;
  const int nx = 13 + 2 * m->desc.n_joints;
  if (int rc = genmesh_hulls<T>(m, dtype, gp, mp, ws, stream)) return rc;
  T* out = (T*)traj;
  if (write_x0) {  // traj[:, 0] = x0
    if (hipMemcpy2DAsync(out, sizeof(T) * ld_item, x0, sizeof(T) * ld_x, sizeof(T) * nx, (size_t)batch, hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return fail(-5, "dpll_simulate_mesh: copy of the initial states failed%s");
    out += ld_step;
  }
  const T* state = (const T*)x0;
  long long ld = ld_x;
  for (long long s = 0; s < steps; ++s) {
    if (int rc = genmesh_support<T>(m, dtype, gp, mp, ws, state, ld, batch, stream, false)) return rc;
    if (int rc = dpll_genmesh::step_items(m, dtype, p, state, ld, batch, out, ld_item, ws + gp.off_W, (const double*)(ws + gp.off_PD), stream))
      return rc;
    state = out;
    ld = ld_item;
    out += ld_step;
  }
  return 0;
}

template <typename T>
int launch_genmesh_terms(const dpll_model* m, int dtype, const dpll_params_t* p, const dpll_mesh_params_t* mp, const void* x,
                         long long ld_x, long long batch, void* Dm, void* M, void* J, void* phi, void* a, void* workspace,
                         long long workspace_bytes, hipStream_t stream) {
  const GenMeshPlan gp = genmesh_plan<T>(m, batch);
  if (!workspace || (size_t)workspace_bytes < gp.total) return fail(-3, "dpll_terms_mesh: workspace too small%s");
  char* ws = (char*)workspace;
  if (int rc = genmesh_hulls<T>(m, dtype, gp, mp, ws, stream)) return rc;
  if (int rc = genmesh_support<T>(m, dtype, gp, mp, ws, x, ld_x, batch, stream, false)) return rc;
  return dpll_genmesh::terms_items(m, dtype, p, x, ld_x, batch, Dm, M, J, phi, a, ws + gp.off_W, (const double*)(ws + gp.off_PD), stream);
}

// support points of every contact SLOT: points (batch, 16, 3) -- slot 4 g + s = query s of geometry g, slot 12 + p = B's
// support point of body-body candidate p; rows of slots that belong to no learned shape hold no information
template <typename T>
int launch_genmesh_support(const dpll_model* m, int dtype, const dpll_mesh_params_t* mp, const void* x, long long ld_x, long long batch,
                           void* points, void* workspace, long long workspace_bytes, hipStream_t stream) {
  const GenMeshPlan gp = genmesh_plan<T>(m, batch);
  if (!workspace || (size_t)workspace_bytes < gp.total) return fail(-3, "dpll_mesh_support_points: workspace too small%s");
  char* ws = (char*)workspace;
  if (int rc = genmesh_hulls<T>(m, dtype, gp, mp, ws, stream)) return rc;
  if (int rc = genmesh_support<T>(m, dtype, gp, mp, ws, x, ld_x, batch, stream, false)) return rc;
  // (slot, side 0) of the slots of the geometries: element (item, slot, 0, :) -> points (item, slot, :)
  const size_t per_item = (size_t)dpll_genmesh::wit_per_item();
  if (hipMemcpy2DAsync(points, sizeof(T) * 3, ws + gp.off_W, sizeof(T) * 6, sizeof(T) * 3, (size_t)batch * (per_item / 6), hipMemcpyDeviceToDevice,
                       stream) != hipSuccess)
    return fail(-5, "dpll_mesh_support_points: copy failed%s");
  return 0;
}

// `mp`: one dpll_mesh_params_t per body (n_joints + 1 of them)
int check_mesh(const dpll_model* m, const dpll_mesh_params_t* mp, const char* who) {
  if (m->forest) return fail(-2, "%s: a model of the forest build has no learned shapes", who);
  if (!mp) return fail(-1, "%s: null mesh parameter pointer", who);
  if (m->desc.n_geoms > 0) {  // general build: one entry per geometry, those of the learned shapes filled in
    int n_mesh = 0;
    for (int g = 0; g < m->desc.n_geoms; ++g) {
      if (m->desc.geom_kind[g] != kGeomMesh) continue;
      ++n_mesh;
      if (!mp[g].hidden_weight || !mp[g].input_weight0 || !mp[g].input_weight1 || !mp[g].output_weight || !mp[g].perturbations)
        return fail(-1, "%s: null mesh parameter pointer", who);
    }
    if (n_mesh == 0) return fail(-2, "%s: the model has no learned shape (DPLL_GEOM_MESH geometry)", who);
    return 0;
  }
  if (m->desc.n_joints > 1) return fail(-2, "%s: the specialised mesh builds take 0 or 1 joints (describe the model with n_geoms > 0 for the general build)", who);
  for (int g = 0; g <= m->desc.n_joints; ++g)
    if (!mp[g].hidden_weight || !mp[g].input_weight0 || !mp[g].input_weight1 || !mp[g].output_weight || !mp[g].perturbations)
      return fail(-1, "%s: null mesh parameter pointer", who);
  return 0;
}

#define DPLL_GENMESH(FN, ...)                                                                \
  do {                                                                                       \
    if (model->desc.n_geoms > 0) {                                                           \
      if (dtype == DPLL_F32) return FN<float>(__VA_ARGS__);                                  \
      return FN<double>(__VA_ARGS__);                                                        \
    }                                                                                        \
  } while (0)

#define DPLL_MESH_DISPATCH(FN, ...)                                                          \
  do {                                                                                       \
    const int nj = model->desc.n_joints;                                                     \
    if (dtype == DPLL_F32 && nj == 0) return FN<float, 0>(__VA_ARGS__);                      \
    if (dtype == DPLL_F32 && nj == 1) return FN<float, 1>(__VA_ARGS__);                      \
    if (dtype == DPLL_F64 && nj == 0) return FN<double, 0>(__VA_ARGS__);                     \
    if (dtype == DPLL_F64 && nj == 1) return FN<double, 1>(__VA_ARGS__);                     \
    return fail(-2, "%s: mesh kernels are built for 0 or 1 joints", #FN);                    \
  } while (0)

int check_common(const dpll_model* m, int dtype, const dpll_params_t* p, long long batch, const char* who) {
  if (!m) return fail(-1, "%s: null model", who);
  for (int g = 0; g < m->desc.n_geoms; ++g)
    if (m->desc.geom_kind[g] == kGeomMesh) return fail(-2, "%s: the model has learned shapes: use the *_mesh entry point", who);
  if (dtype != DPLL_F32 && dtype != DPLL_F64) return fail(-1, "%s: dtype must be DPLL_F32 or DPLL_F64", who);
  if (!p || !p->theta || !p->friction || !p->lengths) return fail(-1, "%s: null parameter pointer", who);
  if (batch < 0) return fail(-1, "%s: negative batch", who);
  // actuation inputs: only a model with actuators takes them (anything else would be a silent drop of B u), with rows of at
  // least n_u numbers
  const int n_u = m->forest ? dpll_forest_api::n_u(m) : m->desc.n_u;
  if (p->u && n_u == 0) return fail(-1, "%s: actuation inputs (params->u) for a model without actuators", who);
  if (p->u && p->ld_u < n_u) return fail(-1, "%s: ld_u smaller than n_u", who);
  return 0;
}

#define DPLL_DISPATCH(FN, ...)                                                               \
  do {                                                                                       \
    const int nj = model->desc.n_joints;                                                     \
    if (dtype == DPLL_F32 && nj == 0) return FN<float, 0>(__VA_ARGS__);                      \
    if (dtype == DPLL_F32 && nj == 1) return FN<float, 1>(__VA_ARGS__);                      \
    if (dtype == DPLL_F64 && nj == 0) return FN<double, 0>(__VA_ARGS__);                     \
    if (dtype == DPLL_F64 && nj == 1) return FN<double, 1>(__VA_ARGS__);                     \
    return fail(-2, "%s: kernels are built for 0 or 1 joints", #FN);                         \
  } while (0)

}  // namespace

extern "C" {

#ifdef DPLL_STAMPS
int dpll_debug_read_stamps(unsigned long long* host_out, int n_rows) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * (size_t)n_rows);
}
#endif

const char* dpll_last_error(void) { return g_error; }
int dpll_abi_version(void) { return 25; }

int dpll_model_create(const dpll_model_desc_t* desc, dpll_model_t** out) {
  if (!desc || !out) return fail(-1, "dpll_model_create: null argument%s");
  const bool general = desc->n_geoms > 0;
  if (desc->n_joints < 0 || desc->n_joints > (general ? DPLL_MAX_JOINTS : 1))
    return fail(-2, "dpll_model_create: the specialised builds take 0 or 1 revolute joints, the general build (n_geoms > 0) up to 3%s");
  if (general) {
    if (desc->n_geoms > DPLL_MAX_GEOMS) return fail(-2, "dpll_model_create: at most 3 collision geometries%s");
    for (int j = 0; j < desc->n_joints; ++j)
      if (desc->parent[j] < 0 || desc->parent[j] > j) return fail(-1, "dpll_model_create: parent[j] must be a body listed before body j + 1%s");
    for (int g = 0; g < desc->n_geoms; ++g) {
      if (desc->geom_body[g] < 0 || desc->geom_body[g] > desc->n_joints) return fail(-1, "dpll_model_create: geom_body out of range%s");
      if (desc->geom_kind[g] != DPLL_GEOM_BOX && desc->geom_kind[g] != DPLL_GEOM_SPHERE && desc->geom_kind[g] != DPLL_GEOM_POLYGON &&
          desc->geom_kind[g] != DPLL_GEOM_MESH)
        return fail(-1, "dpll_model_create: unknown geometry kind%s");
      // (top-4 of the vertex set, geometry.py:196: fewer than 4 vertices cannot answer a support query)
      if (desc->geom_kind[g] == DPLL_GEOM_POLYGON && (desc->geom_nverts[g] < 4 || desc->geom_nverts[g] > DPLL_MAX_POLYGON_VERTICES))
        return fail(-2, "dpll_model_create: a polygon has 4 to 8 vertices%s");
    }
    if (desc->n_pairs < 0 || desc->n_pairs > DPLL_MAX_PAIRS) return fail(-2, "dpll_model_create: at most 4 body-body collision candidates%s");
    for (int p = 0; p < desc->n_pairs; ++p) {
      const int a = desc->pair_a[p], b = desc->pair_b[p];
      if (a < 0 || b < 0 || a >= desc->n_geoms || b >= desc->n_geoms || a == b) return fail(-1, "dpll_model_create: pair geometry out of range%s");
      if (desc->geom_body[a] == desc->geom_body[b]) return fail(-1, "dpll_model_create: a collision candidate joins geometries of two different bodies%s");
      // (the reference collides a learned shape with another learned shape or the ground only, geometry.py:543-551)
      if ((desc->geom_kind[a] == DPLL_GEOM_MESH) != (desc->geom_kind[b] == DPLL_GEOM_MESH))
        return fail(-2, "dpll_model_create: a candidate with a learned shape (DPLL_GEOM_MESH) needs one on both sides%s");
    }
    if (desc->rotated & ~3) return fail(-1, "dpll_model_create: rotated holds bits 0 and 1 only%s");
    // a frame rotation is a proper rotation; identities where the flag says so
    auto proper = [](const double (&R)[3][3], bool identity) {
      double worst = 0.0;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double dot = 0.0;
          for (int k = 0; k < 3; ++k) dot += R[r][k] * R[c][k];
          worst = std::fmax(worst, std::fabs(dot - (r == c ? 1.0 : 0.0)));
          if (identity) worst = std::fmax(worst, std::fabs(R[r][c] - (r == c ? 1.0 : 0.0)));
        }
      const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                         R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
      return worst < 1e-9 && det > 0.0;
    };
    for (int b = 0; b <= desc->n_joints; ++b)
      if (!proper(desc->body_rot[b], !(desc->rotated & 1))) return fail(-1, "dpll_model_create: body_rot must hold rotations (identities unless rotated & 1)%s");
    for (int g = 0; g < desc->n_geoms; ++g)
      if (!proper(desc->geom_rot[g], !(desc->rotated & 2))) return fail(-1, "dpll_model_create: geom_rot must hold rotations (identities unless rotated & 2)%s");
    for (int j = 0; j < desc->n_joints; ++j)
      if (desc->joint_kind[j] != DPLL_JOINT_REVOLUTE && desc->joint_kind[j] != DPLL_JOINT_PRISMATIC)
        return fail(-1, "dpll_model_create: unknown joint kind%s");
  } else if (desc->n_joints > 0 && desc->joint_kind[0] != DPLL_JOINT_REVOLUTE) {
    return fail(-2, "dpll_model_create: prismatic joints need the general build (n_geoms > 0)%s");
  } else if (desc->n_pairs != 0) {
    return fail(-2, "dpll_model_create: body-body collision candidates need the general build (n_geoms > 0)%s");
  } else if (desc->rotated != 0) {
    return fail(-2, "dpll_model_create: rotated frames need the general build (n_geoms > 0)%s");
  }
  if (!(desc->dt > 0.0)) return fail(-1, "dpll_model_create: dt must be positive%s");
  if (desc->inertia_mode != DPLL_INERTIA_REFERENCE_LITERAL && desc->inertia_mode != DPLL_INERTIA_PHYSICAL &&
      desc->inertia_mode != DPLL_INERTIA_COMPOSED)
    return fail(-1, "dpll_model_create: unknown inertia_mode%s");
  if (desc->inertia_mode == DPLL_INERTIA_COMPOSED && !general)
    return fail(-2, "dpll_model_create: composed inertial rows (welded links) need the general build (n_geoms > 0)%s");
  // actuators (B u of lagrangian_forces, multibody_terms.py:142-146): the general build without learned shapes
  if (desc->n_u < 0 || desc->n_u > desc->n_joints) return fail(-1, "dpll_model_create: n_u must be between 0 and n_joints%s");
  if (desc->n_u > 0) {
    if (!general) return fail(-2, "dpll_model_create: actuated joints need the general build (n_geoms > 0)%s");
    for (int g = 0; g < desc->n_geoms; ++g)
      if (desc->geom_kind[g] == DPLL_GEOM_MESH) return fail(-2, "dpll_model_create: actuated joints on a model with learned shapes are not built%s");
    for (int k = 0; k < desc->n_u; ++k)
      if (desc->act_joint[k] < 0 || desc->act_joint[k] >= desc->n_joints) return fail(-1, "dpll_model_create: act_joint out of range%s");
  }
  dpll_model* m = new (std::nothrow) dpll_model;
  if (!m) return fail(-4, "dpll_model_create: out of memory%s");
  std::memcpy(&m->desc, desc, sizeof(ModelDesc));
  m->opts[DPLL_F32] = default_opts(DPLL_F32, desc->n_joints, general);
  m->opts[DPLL_F64] = default_opts(DPLL_F64, desc->n_joints, general);
  *out = m;
  return 0;
}

int dpll_forest_model_create(const dpll_forest_desc_t* desc, dpll_model_t** out) {
  if (!desc || !out) return fail(-1, "dpll_forest_model_create: null argument%s");
  if (int rc = dpll_forest_api::check_desc(desc)) return rc;
  dpll_model* m = new (std::nothrow) dpll_model;
  dpll_forest::ForestDesc* fd = new (std::nothrow) dpll_forest::ForestDesc;
  if (!m || !fd) {
    delete m;
    delete fd;
    return fail(-4, "dpll_forest_model_create: out of memory%s");
  }
  std::memset(&m->desc, 0, sizeof(ModelDesc));
  m->desc.n_joints = desc->n_v;  // (never read for a forest model; kept non-zero so that a stray specialised dispatch fails loudly)
  m->desc.dt = desc->dt;
  m->desc.inertia_mode = desc->inertia_mode;
  static_assert(sizeof(dpll_forest::ForestDesc) == sizeof(dpll_forest_desc_t), "ForestDesc must mirror dpll_forest_desc_t");
  std::memcpy(fd, desc, sizeof(dpll_forest::ForestDesc));
  m->forest = fd;
  m->opts[DPLL_F32] = default_opts(DPLL_F32, 2, true);
  m->opts[DPLL_F64] = default_opts(DPLL_F64, 2, true);
  m->opts[DPLL_F32].portfolio = m->opts[DPLL_F64].portfolio = 1;
  *out = m;
  return 0;
}

void dpll_model_destroy(dpll_model_t* model) {
  if (model && model->forest) dpll_forest_api::release(model);
  delete model;
}

int dpll_model_set_solver(dpll_model_t* model, int dtype, const dpll_solver_opts_t* opts) {
  if (!model || !opts || (dtype != DPLL_F32 && dtype != DPLL_F64)) return fail(-1, "dpll_model_set_solver: bad argument%s");
  if (opts->max_iter < 1 || opts->max_ls < 1 || opts->n_stages < 1 || opts->stage_max_iter < 1 || opts->stage_max_ls < 1 || !(opts->stage_factor >= 1.0))
    return fail(-1, "dpll_model_set_solver: iteration limits must be >= 1%s");
  if (opts->wide < -1 || opts->wide > 1) return fail(-1, "dpll_model_set_solver: wide must be -1, 0 or 1%s");
  if (opts->mesh_gemm < 0 || opts->mesh_gemm > 4)
    return fail(-1, "dpll_model_set_solver: mesh_gemm must be 0 (f32 MFMA, pipelined), 1 (f32 MFMA, the 8-wave kernels), 2 or 3 (bf16 planes) or 4 (two fp16 planes)%s");
  if (opts->portfolio != 0 && opts->portfolio != 1 && opts->portfolio != 2 && opts->portfolio != 4)
    return fail(-1, "dpll_model_set_solver: portfolio must be 0 (by batch size), 1 (off), 2 or 4%s");
  for (int k = 0; k < 3; ++k)
    if (opts->race_stages[k] < 1 || opts->race_stages[k] > kRaceMaxStages || !(opts->race_factor[k] >= 1.0) || (opts->race_flags[k] & ~3))
      return fail(-1, "dpll_model_set_solver: racing schedules need 1 <= race_stages <= 8, race_factor >= 1, race_flags in 0..3%s");
  // (the builds with racing copies form the starting regularisation with a loop of kRaceMaxStages steps, sap_newton: a longer
  // schedule -- of the dynamics solve or of the loss solve, whose schedule loss_item copies into n_stages -- would start too low
  // and END below the reference's eps)
  if (opts->n_stages > kRaceMaxStages && opts->portfolio != 1) return fail(-1, "dpll_model_set_solver: n_stages > 8 needs portfolio = 1%s");
  if (opts->loss_n_stages < 0 || (opts->loss_n_stages > 0 && !(opts->loss_stage_factor >= 1.0)))
    return fail(-1, "dpll_model_set_solver: loss_n_stages >= 0 and, when set, loss_stage_factor >= 1%s");
  if (opts->loss_n_stages > kRaceMaxStages && opts->portfolio != 1)
    return fail(-1, "dpll_model_set_solver: loss_n_stages > 8 needs portfolio = 1%s");
  std::memcpy(&model->opts[dtype], opts, sizeof(SolverOpts));
  return 0;
}

int dpll_model_get_solver(const dpll_model_t* model, int dtype, dpll_solver_opts_t* opts) {
  if (!model || !opts || (dtype != DPLL_F32 && dtype != DPLL_F64)) return fail(-1, "dpll_model_get_solver: bad argument%s");
  std::memcpy(opts, &model->opts[dtype], sizeof(SolverOpts));
  return 0;
}

int dpll_n_x(const dpll_model_t* model) { return !model ? -1 : (model->forest ? dpll_forest_api::n_x(model) : 13 + 2 * model->desc.n_joints); }
int dpll_n_contacts(const dpll_model_t* model) {
  if (!model) return -1;
  if (model->forest) return dpll_forest_api::n_contacts(model);
  return kQuery * (model->desc.n_geoms > 0 ? DPLL_GEN_SLOTS : model->desc.n_joints + 1);  // contact SLOTS of the build
}
int dpll_param_count(const dpll_model_t* model) {
  if (!model) return -1;
  if (model->forest) return dpll_forest_api::param_count(model);
  if (model->desc.n_geoms > 0) return dpll_general::param_count(model);
  const int nb = model->desc.n_joints + 1;
  return 10 * nb + (nb + 1) + 3 * nb;
}

int64_t dpll_workspace_bytes(const dpll_model_t* model, int64_t batch) {
  if (!model || batch < 0) return -1;
  if (model->forest) return dpll_forest_api::workspace_bytes(model, batch);
  if (model->desc.n_geoms > 0) return dpll_general::workspace_bytes(model, batch);
  const int nb = model->desc.n_joints + 1;
  // (the racing build of the loss launch: kRaceWaves-wave workgroups of kRaceWaves * 16 / copies items)
  const int lanes = kQuery * nb, copies = race_copies(0, lanes, batch);
  const int64_t plain = blocks_for(batch, kWave / lanes);
  const int64_t raced = copies > 1 ? blocks_for(batch, kRaceWaves * kWave / (lanes * copies)) : 0;
  const int64_t blocks = plain > raced ? plain : raced;
  const int64_t pi = 1 + 10 * nb + (nb + 1) + 3 * nb;
  const int64_t chain = 100 * nb + (nb + 1) * nb + 3 * nb;  // the rows-to-parameters matrix behind the rows
  return (blocks * pi + chain) * (int64_t)sizeof(double);
}

int dpll_racing_copies(const dpll_model_t* model, int dtype, int64_t batch, int what) {
  if (!model || (dtype != DPLL_F32 && dtype != DPLL_F64) || batch < 0 || what < 0 || what > 4) return -1;
  if (model->forest) return what >= 2 && what != 4 ? -1 : 1;
  if (what == 4) {  // the loss launch of dpll_contactnets_loss_mesh: the single-body build races like the box cube (launch_mesh_loss)
    if (model->desc.n_geoms > 0 || model->desc.n_joints != 0) return 1;
    return (dtype == DPLL_F32 ? mesh_loss_races<float, 0>(model, dtype, batch) : mesh_loss_races<double, 0>(model, dtype, batch)) ? 4 : 1;
  }
  if (what >= 2) {  // the shape of the loss launch: 2 = item workgroups (= partial rows), 3 = lanes of one copy of an item
    if (model->desc.n_geoms > 0 || model->desc.n_joints > 1) return -1;
    LossPlan pl;
    if (dtype == DPLL_F32) pl = model->desc.n_joints == 0 ? plan_loss<float, 0>(model, dtype, batch) : plan_loss<float, 1>(model, dtype, batch);
    else pl = model->desc.n_joints == 0 ? plan_loss<double, 0>(model, dtype, batch) : plan_loss<double, 1>(model, dtype, batch);
    return what == 2 ? pl.rows : (pl.build == kLossWide || pl.build == kLossWideShared ? 1 : pl.lanes);
  }
  if (model->desc.n_geoms > 0 || model->desc.n_joints > 1) return 1;
  const int asked = model->opts[dtype].portfolio;
  if (what == 0) return race_shape(model->desc.n_joints, dtype == DPLL_F32, asked, batch).copies;
  // rollouts: one lane per contact always; the elbow's gain nothing from two copies, so its default is none; the cube's gain
  // from two as well (4096 trajectories, us per step with 1 / 2 / 4 copies: 13.1 / 11.4 / 10.4), so 4097 .. 8192 run with two
  return (model->desc.n_joints > 0 && asked == 0) ? 1 : race_copies(asked, kQuery * (model->desc.n_joints + 1), batch, 4, true);
}

int dpll_contactnets_loss(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x,
                          int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, const void* weights,
                          double scale, void* loss, void* grad, void* loss_total, void* force, int32_t* iters,
                          void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_common(model, dtype, params, batch, "dpll_contactnets_loss")) return rc;
  if (batch == 0 && !grad) return fail(-1, "dpll_contactnets_loss: empty batch%s");
  if (batch > 0 && (!x || !x_plus)) return fail(-1, "dpll_contactnets_loss: null state pointer%s");
  const int nx = dpll_n_x(model);
  if (ld_x < nx || ld_xp < nx) return fail(-1, "dpll_contactnets_loss: row stride smaller than n_x%s");
  if (model->forest)
    return dpll_forest_api::loss(model, dtype, params, x, ld_x, x_plus, ld_xp, batch, weights, scale, loss, grad, loss_total, force, iters,
                                 workspace, workspace_bytes, (hipStream_t)stream);
  if (model->desc.n_geoms > 0)
    return dpll_general::loss(model, dtype, params, x, ld_x, x_plus, ld_xp, batch, weights, scale, loss, grad, loss_total, force,
                              iters, workspace, workspace_bytes, (hipStream_t)stream);
  DPLL_DISPATCH(launch_loss, model, dtype, params, x, ld_x, x_plus, ld_xp, batch, weights, scale, loss, grad, loss_total,
                force, iters, workspace, workspace_bytes, (hipStream_t)stream);
}

int dpll_contactnets_loss_allreduce(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x,
                                    int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, const void* weights,
                                    double scale, void* grad, void* loss_total, void* workspace, int64_t workspace_bytes,
                                    dpll_ar_t* ar, void* stream) {
  if (int rc = check_common(model, dtype, params, batch, "dpll_contactnets_loss_allreduce")) return rc;
  if (model->desc.n_geoms > 0 || model->forest)
    return fail(-2, "dpll_contactnets_loss_allreduce: the general build exchanges gradients with dpll_ar_allreduce after dpll_contactnets_loss%s");
  // batch == 0 is a rank whose shard of a ragged tail batch is empty: it contributes a zero row and still takes part
  if ((batch > 0 && (!x || !x_plus)) || !grad || !loss_total || !ar) return fail(-1, "dpll_contactnets_loss_allreduce: null argument%s");
  const int nx = dpll_n_x(model);
  if (ld_x < nx || ld_xp < nx) return fail(-1, "dpll_contactnets_loss_allreduce: row stride smaller than n_x%s");
  if ((dpll_param_count(model) + 1) * (dtype == DPLL_F64 ? 2 : 1) > dpll_arx::kMaxWords)
    return fail(-1, "dpll_contactnets_loss_allreduce: gradient row too long for the one-shot exchange%s");
  DPLL_DISPATCH(launch_loss, model, dtype, params, x, ld_x, x_plus, ld_xp, batch, weights, scale, nullptr, grad, loss_total,
                nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream, ar);
}

int dpll_contactnets_train_step(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x,
                                int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, const void* weights,
                                double scale, void* grad, void* loss_total, void* workspace, int64_t workspace_bytes,
                                dpll_ar_t* ar, const dpll_adam_t* adam, void* stream) {
  if (int rc = check_common(model, dtype, params, batch, "dpll_contactnets_train_step")) return rc;
  if ((batch > 0 && (!x || !x_plus)) || !grad || !loss_total || !adam || !adam->params || !adam->exp_avg || !adam->exp_avg_sq || !adam->state)
    return fail(-1, "dpll_contactnets_train_step: null argument%s");
  if (model->desc.n_geoms > 0 || model->forest) {
    // the general and the forest build: Adam in the kernel that chains the folded rows to the parameters; the gradient exchange
    // of a data-parallel job stays a call of its own (dpll_ar_allreduce between dpll_contactnets_loss and an optimizer)
    if (ar) return fail(-2, "dpll_contactnets_train_step: the general and forest builds take no exchange handle%s");
    if (batch == 0) return fail(-1, "dpll_contactnets_train_step: empty batch%s");
    const int nx_g = dpll_n_x(model);
    if (ld_x < nx_g || ld_xp < nx_g) return fail(-1, "dpll_contactnets_train_step: row stride smaller than n_x%s");
    if ((const void*)adam->params != params->theta)
      return fail(-1, "dpll_contactnets_train_step: adam.params must be the flat buffer params.theta points to ([theta | friction | lengths])%s");
    if (!(adam->lr >= 0.0) || !(adam->beta1 >= 0.0 && adam->beta1 < 1.0) || !(adam->beta2 >= 0.0 && adam->beta2 < 1.0) || !(adam->eps >= 0.0) || !(adam->weight_decay >= 0.0))
      return fail(-1, "dpll_contactnets_train_step: Adam hyper-parameters out of range%s");
    const AdamArgs args{adam->params, adam->exp_avg, adam->exp_avg_sq, adam->state, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->weight_decay};
    if (model->forest)
      return dpll_forest_api::loss(model, dtype, params, x, ld_x, x_plus, ld_xp, batch, weights, scale, nullptr, grad, loss_total, nullptr, nullptr,
                                   workspace, workspace_bytes, (hipStream_t)stream, &args);
    return dpll_general::loss(model, dtype, params, x, ld_x, x_plus, ld_xp, batch, weights, scale, nullptr, grad, loss_total, nullptr, nullptr,
                              workspace, workspace_bytes, (hipStream_t)stream, &args);
  }
  if (batch == 0 && !ar) return fail(-1, "dpll_contactnets_train_step: empty batch%s");
  const int nx = dpll_n_x(model);
  if (ld_x < nx || ld_xp < nx) return fail(-1, "dpll_contactnets_train_step: row stride smaller than n_x%s");
  if ((const void*)adam->params != params->theta)
    return fail(-1, "dpll_contactnets_train_step: adam.params must be the flat buffer params.theta points to ([theta | friction | lengths])%s");
  if (!(adam->lr >= 0.0) || !(adam->beta1 >= 0.0 && adam->beta1 < 1.0) || !(adam->beta2 >= 0.0 && adam->beta2 < 1.0) || !(adam->eps >= 0.0) || !(adam->weight_decay >= 0.0))
    return fail(-1, "dpll_contactnets_train_step: Adam hyper-parameters out of range%s");
  if (ar && (dpll_param_count(model) + 1) * (dtype == DPLL_F64 ? 2 : 1) > dpll_arx::kMaxWords)
    return fail(-1, "dpll_contactnets_train_step: gradient row too long for the one-shot exchange%s");
  AdamArgs args{adam->params, adam->exp_avg, adam->exp_avg_sq, adam->state, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->weight_decay};
  DPLL_DISPATCH(launch_loss, model, dtype, params, x, ld_x, x_plus, ld_xp, batch, weights, scale, nullptr, grad, loss_total,
                nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream, ar, args);
}

int dpll_profile_contactnets_loss(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x,
                                  int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, double scale,
                                  void* grad, void* workspace, int64_t workspace_bytes, void* stream, int32_t reps,
                                  float* ms_loss_kernel, float* ms_finalize_kernel) {
  if (int rc = check_common(model, dtype, params, batch, "dpll_profile_contactnets_loss")) return rc;
  if (batch == 0 || !x || !x_plus || reps < 1 || reps > 100000) return fail(-1, "dpll_profile_contactnets_loss: bad argument%s");
  if (model->desc.n_geoms > 0 || model->forest) return fail(-2, "dpll_profile_contactnets_loss: specialised builds only%s");
  DPLL_DISPATCH(profile_loss, model, dtype, params, x, ld_x, x_plus, ld_xp, batch, scale, grad, workspace,
                workspace_bytes, (hipStream_t)stream, reps, ms_loss_kernel, ms_finalize_kernel);
}

int dpll_step(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x, int64_t ld_x,
              int64_t batch, void* x_next, int64_t ld_next, int32_t* iters, void* stream) {
  if (int rc = check_common(model, dtype, params, batch, "dpll_step")) return rc;
  if (batch == 0) return 0;
  if (!x || !x_next) return fail(-1, "dpll_step: null state pointer%s");
  const int nx = dpll_n_x(model);
  if (ld_x < nx || ld_next < nx) return fail(-1, "dpll_step: row stride smaller than n_x%s");
  if (model->forest)
    return dpll_forest_api::simulate(model, dtype, params, x, ld_x, batch, 1, x_next, ld_next, 0, 0, iters, (hipStream_t)stream);
  if (model->desc.n_geoms > 0)
    return dpll_general::simulate(model, dtype, params, x, ld_x, batch, 1, x_next, ld_next, 0, 0, iters, (hipStream_t)stream);
  DPLL_DISPATCH(launch_simulate, model, dtype, params, x, ld_x, batch, 1, x_next, ld_next, 0, 0, iters,
                (hipStream_t)stream);
}

int dpll_step_backward(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x, int64_t ld_x,
                       const void* grad_x_next, int64_t ld_g, int64_t batch, void* grad, void* grad_x, int64_t ld_gx,
                       void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_common(model, dtype, params, batch, "dpll_step_backward")) return rc;
  if (batch == 0 || !x || !grad_x_next || !grad) return fail(-1, "dpll_step_backward: bad argument%s");
  const int nx = dpll_n_x(model);
  if (ld_x < nx || ld_g < nx || (grad_x && ld_gx < nx)) return fail(-1, "dpll_step_backward: row stride smaller than n_x%s");
  if (model->forest)
    return dpll_forest_api::step_backward(model, dtype, params, x, ld_x, grad_x_next, ld_g, batch, grad, grad_x, ld_gx, workspace,
                                          workspace_bytes, (hipStream_t)stream);
  if (model->desc.n_geoms > 0)
    return dpll_general::step_backward(model, dtype, params, x, ld_x, grad_x_next, ld_g, batch, grad, grad_x, ld_gx, workspace,
                                       workspace_bytes, (hipStream_t)stream);
  DPLL_DISPATCH(launch_step_backward, model, dtype, params, x, ld_x, grad_x_next, ld_g, batch, grad, workspace,
                workspace_bytes, (hipStream_t)stream, grad_x, ld_gx);
}

int dpll_simulate(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x0, int64_t ld_x,
                  int64_t batch, int64_t steps, void* traj, void* stream) {
  if (int rc = check_common(model, dtype, params, batch, "dpll_simulate")) return rc;
  if (steps < 0) return fail(-1, "dpll_simulate: negative steps%s");
  if (batch == 0) return 0;
  if (!x0 || !traj) return fail(-1, "dpll_simulate: null state pointer%s");
  const int nx = dpll_n_x(model);
  if (ld_x < nx) return fail(-1, "dpll_simulate: row stride smaller than n_x%s");
  if (model->forest)
    return dpll_forest_api::simulate(model, dtype, params, x0, ld_x, batch, steps, traj, (long long)(steps + 1) * nx, nx, 1, nullptr,
                                     (hipStream_t)stream);
  if (model->desc.n_geoms > 0)
    return dpll_general::simulate(model, dtype, params, x0, ld_x, batch, steps, traj, (long long)(steps + 1) * nx, nx, 1, nullptr,
                                  (hipStream_t)stream);
  DPLL_DISPATCH(launch_simulate, model, dtype, params, x0, ld_x, batch, steps, traj, (long long)(steps + 1) * nx, nx, 1,
                nullptr, (hipStream_t)stream);
}

int dpll_mesh_param_count(const dpll_model_t* model) {
  if (!model) return -1;
  if (model->desc.n_geoms > 0) {  // general build: [theta | friction | lengths | one network per learned shape, geometry order]
    int n_mesh = 0;
    for (int g = 0; g < model->desc.n_geoms; ++g) n_mesh += model->desc.geom_kind[g] == kGeomMesh;
    return dpll_general::param_count(model) + n_mesh * kNetParams;
  }
  const int nb = model->desc.n_joints + 1;
  return 10 * nb + 1 + nb + nb * kNetParams;  // [theta | friction | one network per body]
}

int64_t dpll_mesh_workspace_bytes(const dpll_model_t* model, int64_t batch, int dtype) {
  if (!model || batch < 1) return -1;
  if (model->desc.n_geoms > 0)
    return (int64_t)(dtype == DPLL_F64 ? genmesh_plan<double>(model, batch).total : genmesh_plan<float>(model, batch).total);
  if (model->desc.n_joints > 1) return -1;
  if (model->desc.n_joints == 0) return (int64_t)(dtype == DPLL_F64 ? mesh_plan<double, 0>(batch).total : mesh_plan<float, 0>(batch).total);
  return (int64_t)(dtype == DPLL_F64 ? mesh_plan<double, 1>(batch).total : mesh_plan<float, 1>(batch).total);
}

namespace {
int check_mesh_call(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                    const char* who) {
  if (!model) return fail(-1, "%s: null model", who);
  if (dtype != DPLL_F32 && dtype != DPLL_F64) return fail(-1, "%s: bad dtype", who);
  if (params && (!params->theta || !params->friction)) return fail(-1, "%s: null parameter pointer", who);
  if (params && params->u) return fail(-1, "%s: actuation inputs (params->u) for a model without actuators", who);
  t_mesh_gemm = dtype == DPLL_F32 ? model->opts[DPLL_F32].mesh_gemm : 0;  // (the float64 path has no matrix-core form)
  return check_mesh(model, mesh, who);
}
}  // namespace

int dpll_contactnets_loss_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params,
                               const dpll_mesh_params_t* mesh, const void* x, int64_t ld_x, const void* x_plus,
                               int64_t ld_xp, int64_t batch, const void* weights, double scale, void* loss, void* grad,
                               void* loss_total, void* force, int32_t* iters, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  if (!params) return fail(-1, "dpll_contactnets_loss_mesh: null parameter pointer%s");
  if (int rc = check_mesh_call(model, dtype, params, mesh, "dpll_contactnets_loss_mesh")) return rc;
  const int nx = dpll_n_x(model);
  if (batch < 1 || !x || !x_plus || ld_x < nx || ld_xp < nx) return fail(-1, "dpll_contactnets_loss_mesh: bad state arguments%s");
  DPLL_GENMESH(launch_genmesh_loss, model, dtype, params, mesh, x, ld_x, x_plus, ld_xp, batch, weights, scale, loss, grad, loss_total,
               force, iters, workspace, workspace_bytes, (hipStream_t)stream);
  DPLL_MESH_DISPATCH(launch_mesh_loss, model, dtype, params, mesh, x, ld_x, x_plus, ld_xp, batch, weights, scale, loss, grad,
                     loss_total, force, iters, workspace, workspace_bytes, (hipStream_t)stream);
}

int dpll_contactnets_train_step_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                                     const void* x, int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, const void* weights,
                                     double scale, void* grad, void* loss_total, void* workspace, int64_t workspace_bytes,
                                     const dpll_adam_t* adam, void* stream) {
  if (!params) return fail(-1, "dpll_contactnets_train_step_mesh: null parameter pointer%s");
  if (int rc = check_mesh_call(model, dtype, params, mesh, "dpll_contactnets_train_step_mesh")) return rc;
  if (batch == 0 || !x || !x_plus || !grad || !loss_total || !adam || !adam->params || !adam->exp_avg || !adam->exp_avg_sq || !adam->state)
    return fail(-1, "dpll_contactnets_train_step_mesh: null argument or empty batch%s");
  const int nx = dpll_n_x(model);
  if (ld_x < nx || ld_xp < nx) return fail(-1, "dpll_contactnets_train_step_mesh: row stride smaller than n_x%s");
  if ((const void*)adam->params != params->theta)
    return fail(-1, "dpll_contactnets_train_step_mesh: adam.params must be the flat buffer params.theta points to ([theta | friction | networks])%s");
  if (!(adam->lr >= 0.0) || !(adam->beta1 >= 0.0 && adam->beta1 < 1.0) || !(adam->beta2 >= 0.0 && adam->beta2 < 1.0) || !(adam->eps >= 0.0) || !(adam->weight_decay >= 0.0))
    return fail(-1, "dpll_contactnets_train_step_mesh: Adam hyper-parameters out of range%s");
  const AdamArgs args{adam->params, adam->exp_avg, adam->exp_avg_sq, adam->state, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->weight_decay};
  DPLL_GENMESH(launch_genmesh_loss, model, dtype, params, mesh, x, ld_x, x_plus, ld_xp, batch, weights, scale, nullptr, grad, loss_total,
               nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream, &args);
  DPLL_MESH_DISPATCH(launch_mesh_loss, model, dtype, params, mesh, x, ld_x, x_plus, ld_xp, batch, weights, scale, nullptr, grad, loss_total,
                     nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream, &args);
}

int dpll_profile_contactnets_loss_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params,
                                       const dpll_mesh_params_t* mesh, const void* x, int64_t ld_x, const void* x_plus,
                                       int64_t ld_xp, int64_t batch, double scale, void* grad, void* workspace,
                                       int64_t workspace_bytes, void* stream, int32_t reps, float* ms_kernels) {
  if (!grad || !ms_kernels || reps < 1 || reps > 10000) return fail(-1, "dpll_profile_contactnets_loss_mesh: bad argument%s");
  if (!model || model->desc.n_joints != 0) return fail(-2, "dpll_profile_contactnets_loss_mesh: single-body systems (one network: seven kernels)%s");
  const int per = kMeshKernels + 1;
  hipEvent_t* ev = new (std::nothrow) hipEvent_t[(size_t)per * reps];
  if (!ev) return fail(-4, "dpll_profile_contactnets_loss_mesh: out of memory%s");
  bool ok = true;
  for (int i = 0; i < per * reps; ++i) ok = (hipEventCreate(&ev[i]) == hipSuccess) && ok;
  int rc = ok ? 0 : fail(-5, "dpll_profile_contactnets_loss_mesh: hipEventCreate failed%s");
  for (int r = 0; r < reps && rc == 0; ++r) {
    ok = hipEventRecord(ev[per * r], (hipStream_t)stream) == hipSuccess && ok;
    t_mesh_marks = ev + per * r + 1;
    t_mesh_mark = 0;
    rc = dpll_contactnets_loss_mesh(model, dtype, params, mesh, x, ld_x, x_plus, ld_xp, batch, nullptr, scale, nullptr, grad,
                                    nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
    if (rc == 0 && t_mesh_mark != kMeshKernels) rc = fail(-5, "dpll_profile_contactnets_loss_mesh: pipeline recorded an unexpected number of kernels%s");
  }
  t_mesh_marks = nullptr;
  if (rc == 0) {
    ok = hipEventSynchronize(ev[per * reps - 1]) == hipSuccess && ok;
    for (int k = 0; k < kMeshKernels; ++k) {
      double total = 0.0;
      for (int r = 0; r < reps; ++r) {
        float ms = 0.f;
        ok = hipEventElapsedTime(&ms, ev[per * r + k], ev[per * r + k + 1]) == hipSuccess && ok;
        total += ms;
      }
      ms_kernels[k] = (float)(total / reps);
    }
    if (!ok) rc = fail(-5, "dpll_profile_contactnets_loss_mesh: a HIP event call failed%s");
  } else {
    (void)hipStreamSynchronize((hipStream_t)stream);
  }
  for (int i = 0; i < per * reps; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

int dpll_step_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                   const void* x, int64_t ld_x, int64_t batch, void* x_next, int64_t ld_next, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  if (!params) return fail(-1, "dpll_step_mesh: null parameter pointer%s");
  if (int rc = check_mesh_call(model, dtype, params, mesh, "dpll_step_mesh")) return rc;
  const int nx = dpll_n_x(model);
  if (batch < 1 || !x || !x_next || ld_x < nx || ld_next < nx) return fail(-1, "dpll_step_mesh: bad state arguments%s");
  DPLL_GENMESH(launch_genmesh_simulate, model, dtype, params, mesh, x, ld_x, batch, 1, x_next, ld_next, 0, false, workspace,
               workspace_bytes, (hipStream_t)stream);
  DPLL_MESH_DISPATCH(launch_mesh_step, model, dtype, params, mesh, x, ld_x, batch, x_next, ld_next, workspace, workspace_bytes,
                     (hipStream_t)stream);
}

int dpll_simulate_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                       const void* x0, int64_t ld_x, int64_t batch, int64_t steps, void* traj, void* workspace,
                       int64_t workspace_bytes, void* stream) {
  if (!params) return fail(-1, "dpll_simulate_mesh: null parameter pointer%s");
  if (int rc = check_mesh_call(model, dtype, params, mesh, "dpll_simulate_mesh")) return rc;
  const int nx = dpll_n_x(model);
  if (batch < 1 || steps < 1 || !x0 || !traj || ld_x < nx) return fail(-1, "dpll_simulate_mesh: bad state arguments%s");
  DPLL_GENMESH(launch_genmesh_simulate, model, dtype, params, mesh, x0, ld_x, batch, steps, traj, (long long)(steps + 1) * nx, nx, true,
               workspace, workspace_bytes, (hipStream_t)stream);
  DPLL_MESH_DISPATCH(launch_mesh_simulate, model, dtype, params, mesh, x0, ld_x, batch, steps, traj, workspace, workspace_bytes,
                     (hipStream_t)stream);
}

int dpll_step_backward_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                            const void* x, int64_t ld_x, const void* grad_x_next, int64_t ld_g, int64_t batch, void* grad,
                            void* grad_x, int64_t ld_gx, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!params) return fail(-1, "dpll_step_backward_mesh: null parameter pointer%s");
  if (int rc = check_mesh_call(model, dtype, params, mesh, "dpll_step_backward_mesh")) return rc;
  const int nx = dpll_n_x(model);
  if (batch < 1 || !x || !grad_x_next || !grad || ld_x < nx || ld_g < nx || (grad_x && ld_gx < nx))
    return fail(-1, "dpll_step_backward_mesh: bad arguments%s");
  DPLL_GENMESH(launch_genmesh_step_backward, model, dtype, params, mesh, x, ld_x, grad_x_next, ld_g, batch, grad, grad_x, ld_gx, workspace,
               workspace_bytes, (hipStream_t)stream);
  DPLL_MESH_DISPATCH(launch_mesh_step_backward, model, dtype, params, mesh, x, ld_x, grad_x_next, ld_g, batch, grad, grad_x, ld_gx,
                     workspace, workspace_bytes, (hipStream_t)stream);
}

int dpll_mesh_support_points(const dpll_model_t* model, int dtype, const dpll_mesh_params_t* mesh, const void* x,
                             int64_t ld_x, int64_t batch, void* points, void* workspace, int64_t workspace_bytes,
                             void* stream) {
  if (int rc = check_mesh_call(model, dtype, nullptr, mesh, "dpll_mesh_support_points")) return rc;
  if (batch < 1 || !x || !points || ld_x < (model->desc.n_joints == 0 ? 4 : dpll_n_x(model)))
    return fail(-1, "dpll_mesh_support_points: bad arguments%s");
  DPLL_GENMESH(launch_genmesh_support, model, dtype, mesh, x, ld_x, batch, points, workspace, workspace_bytes, (hipStream_t)stream);
  DPLL_MESH_DISPATCH(launch_mesh_support, model, mesh, x, ld_x, batch, points, workspace, workspace_bytes, (hipStream_t)stream);
}

int dpll_terms_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                    const void* x, int64_t ld_x, int64_t batch, void* delassus, void* M, void* J, void* phi, void* a,
                    void* workspace, int64_t workspace_bytes, void* stream) {
  if (!params) return fail(-1, "dpll_terms_mesh: null parameter pointer%s");
  if (int rc = check_mesh_call(model, dtype, params, mesh, "dpll_terms_mesh")) return rc;
  if (batch < 1 || !x || ld_x < dpll_n_x(model)) return fail(-1, "dpll_terms_mesh: bad state arguments%s");
  DPLL_GENMESH(launch_genmesh_terms, model, dtype, params, mesh, x, ld_x, batch, delassus, M, J, phi, a, workspace, workspace_bytes,
               (hipStream_t)stream);
  DPLL_MESH_DISPATCH(launch_mesh_terms, model, params, mesh, x, ld_x, batch, delassus, M, J, phi, a, workspace, workspace_bytes,
                     (hipStream_t)stream);
}

int dpll_terms(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x, int64_t ld_x,
               int64_t batch, void* delassus, void* M, void* J, void* phi, void* a, void* stream) {
  if (int rc = check_common(model, dtype, params, batch, "dpll_terms")) return rc;
  if (batch == 0) return 0;
  if (!x) return fail(-1, "dpll_terms: null state pointer%s");
  if (ld_x < dpll_n_x(model)) return fail(-1, "dpll_terms: row stride smaller than n_x%s");
  if (model->forest) return dpll_forest_api::terms(model, dtype, params, x, ld_x, batch, delassus, M, J, phi, a, (hipStream_t)stream);
  if (model->desc.n_geoms > 0)
    return dpll_general::terms(model, dtype, params, x, ld_x, batch, delassus, M, J, phi, a, (hipStream_t)stream);
  DPLL_DISPATCH(launch_terms, model, params, x, ld_x, batch, delassus, M, J, phi, a, (hipStream_t)stream);
}

}  // extern "C"

// ---- one-shot all-reduce over peer memory (dpll_allreduce.hpp) -------------------------------------

extern "C" {

int64_t dpll_ar_handle_bytes(void) { return (int64_t)sizeof(hipIpcMemHandle_t); }

/* Allocates this rank's receive buffer and writes its IPC handle (dpll_ar_handle_bytes() bytes) to handle_out. */
int dpll_ar_create(int rank, int world, void* handle_out, dpll_ar_t** out) {
  if (!handle_out || !out || world < 1 || world > dpll_arx::kMaxWorld || rank < 0 || rank >= world)
    return fail(-1, "dpll_ar_create: bad argument%s");
  dpll_ar* ar = new (std::nothrow) dpll_ar;
  if (!ar) return fail(-4, "dpll_ar_create: out of memory%s");
  std::memset(ar, 0, sizeof(*ar));
  ar->rank = rank;
  ar->world = world;
  const size_t bytes = sizeof(unsigned long long) * 2 * (size_t)world * dpll_arx::kMaxWords;
  if (hipExtMallocWithFlags(&ar->local, bytes, hipDeviceMallocUncached) != hipSuccess) {
    delete ar;
    return fail(-5, "dpll_ar_create: hipExtMallocWithFlags(uncached) failed%s");
  }
  if (hipMemset(ar->local, 0, bytes) != hipSuccess || hipMalloc((void**)&ar->state, 2 * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(ar->state, 0, 2 * sizeof(uint32_t)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    dpll_ar_destroy(ar);  // frees whatever was allocated
    return fail(-5, "dpll_ar_create: buffer setup failed%s");
  }
  hipIpcMemHandle_t handle;
  if (hipIpcGetMemHandle(&handle, ar->local) != hipSuccess) {
    dpll_ar_destroy(ar);
    return fail(-5, "dpll_ar_create: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 needed)%s");
  }
  std::memcpy(handle_out, &handle, sizeof(handle));
  *out = ar;
  return 0;
}

/* handles: world * dpll_ar_handle_bytes() bytes, the handle of rank r at offset r * dpll_ar_handle_bytes(). */
int dpll_ar_connect(dpll_ar_t* ar, const void* handles) {
  if (!ar || !handles) return fail(-1, "dpll_ar_connect: bad argument%s");
  for (int r = 0; r < ar->world; ++r) {
    if (r == ar->rank) {
      ar->peers.recv[r] = (unsigned long long*)ar->local;
      continue;
    }
    hipIpcMemHandle_t handle;
    std::memcpy(&handle, (const char*)handles + (size_t)r * sizeof(handle), sizeof(handle));
    void* ptr = nullptr;
    if (hipIpcOpenMemHandle(&ptr, handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
      return fail(-5, "dpll_ar_connect: hipIpcOpenMemHandle failed%s");
    ar->opened[r] = ptr;
    ar->peers.recv[r] = (unsigned long long*)ptr;
  }
  return 0;
}

/* In-place SUM over the ranks of `n` elements (device pointer, dtype float or double); every rank must call it the
 * same number of times.  Launches one small kernel on `stream` (graph-capturable).  n * sizeof(element) <= 1 KiB. */
int dpll_ar_allreduce(dpll_ar_t* ar, int dtype, void* data, int n, void* stream) {
  if (!ar || !data || n < 1) return fail(-1, "dpll_ar_allreduce: bad argument%s");
  const int words = n * (dtype == DPLL_F64 ? 2 : 1);
  if (words > dpll_arx::kMaxWords) return fail(-1, "dpll_ar_allreduce: message too long%s");
  if (dtype == DPLL_F64)
    hipLaunchKernelGGL(dpll_arx::allreduce_kernel<double>, dim3(1), dim3(256), 0, (hipStream_t)stream, (double*)data, n,
                       ar->peers, ar->rank, ar->world, ar->state, ar->state + 1);
  else if (dtype == DPLL_F32)
    hipLaunchKernelGGL(dpll_arx::allreduce_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, (float*)data, n,
                       ar->peers, ar->rank, ar->world, ar->state, ar->state + 1);
  else
    return fail(-1, "dpll_ar_allreduce: bad dtype%s");
  return check_launch("allreduce_kernel");
}

/* Synchronises the device and returns 0 if no call timed out so far, 1 otherwise. */
int dpll_ar_status(dpll_ar_t* ar) {
  if (!ar) return -1;
  uint32_t host[2] = {0, 0};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, ar->state, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(-5, "dpll_ar_status: device error%s");
  return host[1] ? 1 : 0;
}

void dpll_ar_destroy(dpll_ar_t* ar) {
  if (!ar) return;
  for (int r = 0; r < ar->world; ++r)
    if (ar->opened[r]) (void)hipIpcCloseMemHandle(ar->opened[r]);
  if (ar->local) (void)hipFree(ar->local);
  if (ar->state) (void)hipFree(ar->state);
  delete ar;
}

}  // extern "C"
