// dpll_general.hip -- the GENERAL build of the contact-dynamics kernels: any floating-base tree of up to kMaxJoints
// revolute joints with up to kMaxGeoms box / sphere / polygon collision geometries on any of its bodies, against the ground.
//
// What the reference handles generically in Python -- MultibodyTerms for any number of bodies and geometries
// (multibody_terms.py:328-382, 428-521), the state space inferred from the tree (drake_utils.py:309-335), Sphere next
// to Box (geometry.py:367-456) -- beyond the two topologies the specialised builds of dpll_kernels.hip are written for
// (cube, elbow).  Same per-item math (dpll_core.hpp, with the tree / geometry-table branches selected by GeneralDesc),
// same C ABI, same partial-row + chain-matrix gradient reduction; the mapping is one lane per contact SLOT, 16 lanes (a
// DPP row) per item: always kMaxGeoms x 4 geometry slots + 4 for the body-body candidates (one contact each), where geometries
// or candidates a model does not have and the three slots a sphere does not use are masked to "far away" (idle lanes, no branch in the solver).
// Coverage before speed: every contact Jacobian is dense, the Newton system up to 8 x 8.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "dpll_common.hpp"
#include "dpll_general.hpp"

namespace {

using namespace dpll;

constexpr int kNG = kGenSlots;  // the geometries + the slot of a body-body pair
constexpr int kGP = GeneralDesc::kGeoStride;  // numbers per geometry parameter block
template <typename T, int NJ> using GD = Dims<T, NJ, kNG, kGP>;
template <typename T, int NJ> using GenGrad = LossGrad<T, NJ, kNG, kGP>;
using GenLanes = GpuLanes<kQuery * kNG>;  // one lane per contact slot

// One lane per contact SLOT: an item is owned by the 16 lanes of a DPP row (3 geometries x 4 witnesses + up to 4
// body-body candidates), four items per wave -- the mapping of the specialised builds (cube 4, elbow 8 lanes per
// item), so each lane's contact state stays in registers and 4096 items are 1024 waves, one per SIMD.
constexpr int kIPW = kWave / (kQuery * kNG);  // items per wave
template <typename T, int NJ>
__global__ __launch_bounds__(kWave) void gen_loss_kernel(GeneralDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                         const T* __restrict__ friction, const T* __restrict__ lengths,
                                                         const T* __restrict__ x, long long ld_x, const T* __restrict__ xp,
                                                         long long ld_xp, long long batch, const T* __restrict__ weights,
                                                         double scale, T* __restrict__ loss, T* __restrict__ force,
                                                         int* __restrict__ iters, double* __restrict__ partials, int want_grad) {
  using D = GD<T, NJ>;
  static_assert(D::G == 16 && kIPW == 4, "16 contact slots per item");
  const int lane = threadIdx.x, cidx = lane % D::G, slot = lane / D::G;
  const int item_blocks = (int)gridDim.x - 1;  // the last workgroup owns no items: it writes the chain matrix
  if ((int)blockIdx.x == item_blocks) {
    if (want_grad)
      write_chain_matrix<T, T, D::NB, kNG, kGP>(md.inertia_mode, theta, friction, lengths, partials + (long long)item_blocks * D::PI, &md);
    return;
  }
  Derived<T, NJ, kNG> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  GenGrad<T, NJ> acc;
  zero_grad(acc);
  double loss_acc = 0.0;
  const long long stride = (long long)item_blocks * kIPW;
  for (long long base = (long long)blockIdx.x * kIPW; base < batch; base += stride) {
    const long long item = base + slot;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;  // idle groups shadow the last item: every lane stays live for DPP
    T xr[D::NX], xpr[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) { xr[i] = x[it * ld_x + i]; xpr[i] = xp[it * ld_xp + i]; }
    const T w = valid ? T(scale) * (weights ? weights[it] : T(1)) : T(0);
    T f[1][3];
    int n_it = 0;
    const T L = loss_item<T, typename Acc<T>::type, NJ, 1, GenLanes>(md, dp, opt, xr, xpr, cidx, w, want_grad != 0, acc, f, n_it);
    if (valid) {
      if (cidx == 0) {
        if (loss) loss[it] = L;
        if (iters) iters[it] = n_it;
      }
      if (force) {
        T* row = force + it * (3 * D::K);
        row[cidx] = f[0][2];
        row[D::K + 2 * cidx] = f[0][0];
        row[D::K + 2 * cidx + 1] = f[0][1];
      }
    }
    loss_acc += (cidx == 0) ? double(w) * double(L) : 0.0;
  }
  if (!want_grad) return;
  store_iota_row<T, NJ, D::G, kNG, kGP>(acc, loss_acc, partials);
}

// Fixed-order sum of the partial rows in two stages (a 4096-pair launch leaves 1024 rows, one per wave): blocks of
// kFoldRows rows are folded first, eight loads in flight per thread, into a second row array behind the chain matrix; the
// finalize kernel sums that (<= 32 rows) and applies the chain to the parameters.  One thread per column.
constexpr int kFoldRows = 64;
template <typename T, int NJ>
__global__ __launch_bounds__(256) void gen_fold_rows_kernel(const double* __restrict__ partials, int n_rows, double* __restrict__ folded) {
  using D = GD<T, NJ>;
  const int col = threadIdx.x;
  if (col >= D::PIOTA) return;
  const int r0 = (int)blockIdx.x * kFoldRows, r1 = r0 + kFoldRows < n_rows ? r0 + kFoldRows : n_rows;
  double s = 0.0;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partials[(long long)(r + u) * D::PI + col];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; r < r1; ++r) s += partials[(long long)r * D::PI + col];
  folded[(long long)blockIdx.x * D::PI + col] = s;
}
template <typename T, int NJ>
__global__ __launch_bounds__(256) void gen_finalize_kernel(const double* __restrict__ folded, int n_rows, const double* __restrict__ chain,
                                                           T* __restrict__ grad, T* __restrict__ loss_total) {
  using D = GD<T, NJ>;
  static_assert(D::PI <= 256, "row must fit 256 columns");
  __shared__ double tot[256];
  const int col = threadIdx.x;
  double s = 0.0;
  if (col < D::PIOTA)
    for (int r = 0; r < n_rows; ++r) s += folded[(long long)r * D::PI + col];
  tot[col] = s;
  __syncthreads();
  if (threadIdx.x < D::PI) {
    const int k = (int)threadIdx.x - 1;
    const double v = k < 0 ? tot[0] : apply_chain<D::NB, kNG, kGP>(tot, chain, k);
    if (k < 0) {
      if (loss_total) *loss_total = T(v);
    } else {
      grad[k] = T(v);
    }
  }
}

template <typename T, int NJ>
__global__ __launch_bounds__(kWave) void gen_simulate_kernel(GeneralDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                             const T* __restrict__ friction, const T* __restrict__ lengths,
                                                             const T* __restrict__ x0, long long ld_x, long long batch,
                                                             long long steps, T* __restrict__ out, long long ld_item,
                                                             long long ld_step, int write_x0, int* __restrict__ iters) {
  using D = GD<T, NJ>;
  const int lane = threadIdx.x, cidx = lane % D::G, slot = lane / D::G;
  Derived<T, NJ, kNG> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  const long long stride = (long long)gridDim.x * kIPW;
  for (long long base = (long long)blockIdx.x * kIPW; base < batch; base += stride) {
    const long long item = base + slot;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    const bool writer = valid && cidx == 0;
    T xr[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) xr[i] = x0[it * ld_x + i];
    T* dst = out + it * ld_item;
    if (write_x0) {
      if (writer) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) dst[i] = xr[i];
      }
      dst += ld_step;
    }
    int total = 0;
    for (long long s = 0; s < steps; ++s) {
      T xn[D::NX], imp[1][3];
      int n_it = 0;
      step_item<T, typename Acc<T>::type, NJ, 1, GenLanes>(md, dp, opt, xr, cidx, xn, imp, n_it);
      total += n_it;
#pragma unroll
      for (int i = 0; i < D::NX; ++i) xr[i] = xn[i];
      if (writer) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) dst[i] = xr[i];
      }
      dst += ld_step;
    }
    if (iters && writer) iters[it] = total;
  }
}

// backward of one step: parameter gradient and state adjoint (double arithmetic, as in the specialised builds)
template <typename T, int NJ>
__global__ __launch_bounds__(kWave) void gen_step_backward_kernel(GeneralDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                                  const T* __restrict__ friction, const T* __restrict__ lengths,
                                                                  const T* __restrict__ x, long long ld_x,
                                                                  const T* __restrict__ gx, long long ld_g, long long batch,
                                                                  double* __restrict__ partials, T* __restrict__ xbar_out,
                                                                  long long ld_xb) {
  using D = GD<T, NJ>;
  using C = double;
  const int lane = threadIdx.x, cidx = lane % D::G, slot = lane / D::G;
  const int item_blocks = (int)gridDim.x - 1;
  if ((int)blockIdx.x == item_blocks) {
    write_chain_matrix<C, T, D::NB, kNG, kGP>(md.inertia_mode, theta, friction, lengths, partials + (long long)item_blocks * D::PI, &md);
    return;
  }
  // the parameters in double: local copies (a polygon's vertices are read through dp.geo)
  C theta_c[D::NB * 10], friction_c[kNG + 1], lengths_c[kNG * kGP];
#pragma unroll
  for (int i = 0; i < D::NB * 10; ++i) theta_c[i] = C(theta[i]);
#pragma unroll
  for (int i = 0; i < kNG + 1; ++i) friction_c[i] = C(friction[i]);
  for (int i = 0; i < kNG * kGP; ++i) lengths_c[i] = C(lengths[i]);
  Derived<C, NJ, kNG> dp;
  derive_params<C, NJ>(md, theta_c, friction_c, lengths_c, dp);
  GenGrad<C, NJ> acc;
  zero_grad(acc);
  const long long stride = (long long)item_blocks * kIPW;
  for (long long base = (long long)blockIdx.x * kIPW; base < batch; base += stride) {
    const long long item = base + slot;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    C xr[D::NX], gr[D::NX], xb[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) { xr[i] = C(x[it * ld_x + i]); gr[i] = valid ? C(gx[it * ld_g + i]) : C(0); xb[i] = C(0); }
    // (an idle group's seed is zero, so what it adds to the sums below is zero)
    step_item_backward<C, C, NJ, 1, GenLanes>(md, dp, opt, xr, cidx, gr, acc, nullptr, nullptr, &xb);
    if (xbar_out && valid && cidx == 0) {
#pragma unroll
      for (int i = 0; i < D::NX; ++i) xbar_out[it * ld_xb + i] = T(xb[i]);
    }
  }
  store_iota_row<C, NJ, D::G, kNG, kGP>(acc, 0.0, partials);
}

// MultibodyTerms.forward (multibody_terms.py:584-609) over all kMaxGeoms x 4 contact slots; the host keeps the real ones
template <typename T, int NJ>
__global__ __launch_bounds__(kWave) void gen_terms_kernel(GeneralDesc md, const T* __restrict__ theta, const T* __restrict__ friction,
                                                          const T* __restrict__ lengths, const T* __restrict__ x, long long ld_x,
                                                          long long batch, T* __restrict__ Dout, T* __restrict__ Mout,
                                                          T* __restrict__ Jout, T* __restrict__ phiout, T* __restrict__ aout) {
  using D = GD<T, NJ>;
  constexpr int NV = D::NV, K = D::K;
  Derived<T, NJ, kNG> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  const long long it = (long long)blockIdx.x * kWave + threadIdx.x;
  if (it >= batch) return;
  T xr[D::NX];
#pragma unroll
  for (int i = 0; i < D::NX; ++i) xr[i] = x[it * ld_x + i];
  Terms<T, NJ> t;
  Kin<typename Acc<T>::type, NJ> kinA;
  compute_terms<T, typename Acc<T>::type, NJ>(md, dp, xr, xr + D::NQ, t, kinA);
  if (Mout) {
    for (int i = 0; i < NV; ++i)
      for (int j = 0; j < NV; ++j) Mout[(it * NV + i) * NV + j] = t.M[i][j];
  }
  if (aout) {
    for (int i = 0; i < NV; ++i) aout[it * NV + i] = t.a[i];
  }
  // rows of J in the reference order [normals | mu (t_x, t_y) per contact] (multibody_terms.py:415-426), into Jout
  // (required when Dout is requested: the Delassus rows are formed from it)
  T* Jrows = Jout + it * (3 * K) * NV;
  for (int c = 0; c < K; ++c) {
    ContactGeom<T, NJ, true> cg;
    compute_contact<T, typename Acc<T>::type, NJ>(md, dp, t.kin, kinA, c, cg);
    if (phiout) phiout[it * K + c] = cg.phi;
    const int rows[3] = {c, K + 2 * c, K + 2 * c + 1};
    for (int i = 0; i < NV; ++i) {
      T col[3];
      cjac_column<T, NJ>(cg.J, i, col);
      Jrows[rows[0] * NV + i] = col[2];
      Jrows[rows[1] * NV + i] = cg.mu * col[0];
      Jrows[rows[2] * NV + i] = cg.mu * col[1];
    }
  }
  if (Dout) {
    for (int r = 0; r < 3 * K; ++r) {
      T row[NV], w[NV];
      for (int i = 0; i < NV; ++i) row[i] = Jrows[r * NV + i];
      chol_solve<T, NV>(t.LM, t.invdM, row, w);
      for (int c = 0; c < 3 * K; ++c) {
        T s = T(0);
        for (int i = 0; i < NV; ++i) s += w[i] * Jrows[c * NV + i];
        Dout[(it * 3 * K + r) * (3 * K) + c] = s;
      }
    }
  }
}

GeneralDesc general_desc(const dpll_model* m) {
  GeneralDesc gd;
  static_cast<ModelDesc&>(gd) = m->desc;
  return gd;
}

int row_blocks(long long batch) {
  long long blocks = (batch + kIPW - 1) / kIPW;
  if (blocks > kMaxLossBlocks) blocks = kMaxLossBlocks;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// workspace: [rows (n, PI) | chain matrix | folded rows (ceil(n / kFoldRows), PI)]
long long folded_rows(long long rows) { return (rows + kFoldRows - 1) / kFoldRows; }
template <typename T, int NJ>
int finalize_rows(double* workspace, int rows, T* grad, T* loss_total, hipStream_t stream) {
  using D = GD<T, NJ>;
  double* chain = workspace + (long long)rows * D::PI;
  double* folded = chain + D::CHAIN;
  const int n_folded = (int)folded_rows(rows);
  if (n_folded > 0) {
    hipLaunchKernelGGL((gen_fold_rows_kernel<T, NJ>), dim3(n_folded), dim3(256), 0, stream, (const double*)workspace, rows, folded);
    if (int rc = dpll_check_launch("gen_fold_rows_kernel")) return rc;
  }
  hipLaunchKernelGGL((gen_finalize_kernel<T, NJ>), dim3(1), dim3(256), 0, stream, (const double*)folded, n_folded, (const double*)chain,
                     grad, loss_total);
  return dpll_check_launch("gen_finalize_kernel");
}

template <typename T, int NJ>
int launch_loss(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp,
                long long ld_xp, long long batch, const void* weights, double scale, void* loss, void* grad, void* loss_total,
                void* force, int32_t* iters, void* workspace, long long workspace_bytes, hipStream_t stream) {
  using D = GD<T, NJ>;
  const int rows = batch > 0 ? row_blocks(batch) : 0;
  const int want_grad = grad != nullptr;
  if (want_grad) {
    if (!workspace || workspace_bytes < dpll_general::workspace_bytes(m, batch))
      return dpll_fail(-3, "dpll_contactnets_loss: workspace too small%s");
  } else if (loss_total) {
    return dpll_fail(-3, "dpll_contactnets_loss: loss_total requires grad%s");
  }
  hipLaunchKernelGGL((gen_loss_kernel<T, NJ>), dim3(rows + 1), dim3(kWave), 0, stream, general_desc(m), m->opts[dtype],
                     (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)xp, ld_xp,
                     batch, (const T*)weights, scale, (T*)loss, (T*)force, (int*)iters, (double*)workspace, want_grad);
  if (int rc = dpll_check_launch("gen_loss_kernel")) return rc;
  if (want_grad) return finalize_rows<T, NJ>((double*)workspace, rows, (T*)grad, (T*)loss_total, stream);
  return 0;
}

template <typename T, int NJ>
int launch_simulate(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x0, long long ld_x, long long batch,
                    long long steps, void* out, long long ld_item, long long ld_step, int write_x0, int32_t* iters,
                    hipStream_t stream) {
  long long blocks = (batch + kIPW - 1) / kIPW;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((gen_simulate_kernel<T, NJ>), dim3((int)blocks), dim3(kWave), 0, stream, general_desc(m), m->opts[dtype],
                     (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x0, ld_x, batch, steps, (T*)out,
                     ld_item, ld_step, write_x0, (int*)iters);
  return dpll_check_launch("gen_simulate_kernel");
}

template <typename T, int NJ>
int launch_step_backward(const dpll_model* m, const dpll_params_t* p, const void* x, long long ld_x, const void* gx,
                         long long ld_g, long long batch, void* grad, void* grad_x, long long ld_gx, void* workspace,
                         long long workspace_bytes, hipStream_t stream) {
  using D = GD<T, NJ>;
  const int rows = row_blocks(batch);
  if (!workspace || workspace_bytes < dpll_general::workspace_bytes(m, batch))
    return dpll_fail(-3, "dpll_step_backward: workspace too small%s");
  hipLaunchKernelGGL((gen_step_backward_kernel<T, NJ>), dim3(rows + 1), dim3(kWave), 0, stream, general_desc(m), m->opts[DPLL_F64],
                     (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)gx, ld_g, batch,
                     (double*)workspace, (T*)grad_x, ld_gx);
  if (int rc = dpll_check_launch("gen_step_backward_kernel")) return rc;
  return finalize_rows<T, NJ>((double*)workspace, rows, (T*)grad, (T*)nullptr, stream);
}

template <typename T, int NJ>
int launch_terms(const dpll_model* m, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm, void* M,
                 void* J, void* phi, void* a, hipStream_t stream) {
  if (!J) return dpll_fail(-1, "dpll_terms (general build): the J output is required%s");
  const long long blocks = (batch + kWave - 1) / kWave;
  hipLaunchKernelGGL((gen_terms_kernel<T, NJ>), dim3((int)blocks), dim3(kWave), 0, stream, general_desc(m), (const T*)p->theta,
                     (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, batch, (T*)Dm, (T*)M, (T*)J, (T*)phi, (T*)a);
  return dpll_check_launch("gen_terms_kernel");
}

#define DPLL_GEN_DISPATCH(FN, ...)                                                            \
  do {                                                                                        \
    const int nj = m->desc.n_joints;                                                          \
    if (dtype == DPLL_F32 && nj == 0) return FN<float, 0>(__VA_ARGS__);                       \
    if (dtype == DPLL_F32 && nj == 1) return FN<float, 1>(__VA_ARGS__);                       \
    if (dtype == DPLL_F32 && nj == 2) return FN<float, 2>(__VA_ARGS__);                       \
    if (dtype == DPLL_F32 && nj == 3) return FN<float, 3>(__VA_ARGS__);                       \
    if (dtype == DPLL_F64 && nj == 0) return FN<double, 0>(__VA_ARGS__);                      \
    if (dtype == DPLL_F64 && nj == 1) return FN<double, 1>(__VA_ARGS__);                      \
    if (dtype == DPLL_F64 && nj == 2) return FN<double, 2>(__VA_ARGS__);                      \
    if (dtype == DPLL_F64 && nj == 3) return FN<double, 3>(__VA_ARGS__);                      \
    return dpll_fail(-2, "%s: the general build covers 0 to 3 joints", #FN);                   \
  } while (0)

}  // namespace

namespace dpll_general {

int param_count(const dpll_model* m) { return 10 * (m->desc.n_joints + 1) + (kNG + 1) + kGP * kNG; }

// [rows (n, PI) | chain matrix (CHAIN) | folded rows (ceil(n / kFoldRows), PI)] with the constants of the very Dims the
// kernels lay the workspace out with (round 2 restated them by hand and left out the body-body block of the chain
// matrix: the last folded row was written 152 bytes past the end)
template <int NJ> long long workspace_doubles(long long rows) {
  using D = GD<double, NJ>;
  return (rows + folded_rows(rows)) * D::PI + D::CHAIN;
}
long long workspace_bytes(const dpll_model* m, long long batch) {
  const long long rows = row_blocks(batch);
  long long doubles = 0;
  switch (m->desc.n_joints) {
    case 0: doubles = workspace_doubles<0>(rows); break;
    case 1: doubles = workspace_doubles<1>(rows); break;
    case 2: doubles = workspace_doubles<2>(rows); break;
    default: doubles = workspace_doubles<3>(rows); break;
  }
  return doubles * (long long)sizeof(double);
}

int loss(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp, long long ld_xp,
         long long batch, const void* weights, double scale, void* loss_out, void* grad, void* loss_total, void* force,
         int32_t* iters, void* workspace, long long ws_bytes, hipStream_t stream) {
  DPLL_GEN_DISPATCH(launch_loss, m, dtype, p, x, ld_x, xp, ld_xp, batch, weights, scale, loss_out, grad, loss_total, force, iters,
                    workspace, ws_bytes, stream);
}

int simulate(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x0, long long ld_x, long long batch,
             long long steps, void* out, long long ld_item, long long ld_step, int write_x0, int32_t* iters, hipStream_t stream) {
  DPLL_GEN_DISPATCH(launch_simulate, m, dtype, p, x0, ld_x, batch, steps, out, ld_item, ld_step, write_x0, iters, stream);
}

int step_backward(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* gx,
                  long long ld_g, long long batch, void* grad, void* grad_x, long long ld_gx, void* workspace, long long ws_bytes,
                  hipStream_t stream) {
  DPLL_GEN_DISPATCH(launch_step_backward, m, p, x, ld_x, gx, ld_g, batch, grad, grad_x, ld_gx, workspace, ws_bytes, stream);
}

int terms(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm, void* M,
          void* J, void* phi, void* a, hipStream_t stream) {
  DPLL_GEN_DISPATCH(launch_terms, m, p, x, ld_x, batch, Dm, M, J, phi, a, stream);
}

}  // namespace dpll_general
