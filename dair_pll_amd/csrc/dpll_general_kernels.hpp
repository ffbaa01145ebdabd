// dpll_general_kernels.hpp -- the kernels of the GENERAL build (any tree, up to kMaxGeoms geometries of any kind, body-body
// candidates), shared by its two translation units: dpll_general.hip instantiates them for box / sphere / polygon models
// (MESH = false), dpll_genmesh.hip for models with learned shapes (MESH = true: DeepSupportConvex geometries whose support
// points the ICNN kernels evaluate; the kernels then read a witness per contact slot and write its adjoint).
#pragma once
#include <hip/hip_runtime.h>

#include "dpll_common.hpp"

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

// MESH kernels: per item and contact slot two 3-vectors, [slot][0] = the support point of the slot's geometry (a body-body
// candidate: of B along -d), [slot][1] = of A along d (candidates only), written by the ICNN kernels; the adjoints come back
// in the same layout.  `pdirs` (batch, kMaxPairs, 3) doubles: the candidates' directions from the GJK / EPA kernel.
constexpr int kWitPerItem = kQuery * kNG * 2 * 3;
template <typename T, typename TA>
__device__ __forceinline__ void load_mesh_inputs(const T* __restrict__ wit, const double* __restrict__ pdirs, long long it, int cidx,
                                                 T (&w)[1][3], MeshPairIn<T, TA, 1>& in) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    w[0][i] = wit[it * kWitPerItem + (cidx * 2 + 0) * 3 + i];
    in.wit_a[0][i] = wit[it * kWitPerItem + (cidx * 2 + 1) * 3 + i];
  }
#pragma unroll
  for (int p = 0; p < kMaxPairs; ++p)
#pragma unroll
    for (int i = 0; i < 3; ++i) in.dirs[p][i] = TA(pdirs[(it * kMaxPairs + p) * 3 + i]);
}
template <typename T, int NJ, bool MESH = false>
__global__ __launch_bounds__(kWave) void gen_loss_kernel(GeneralDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                         const T* __restrict__ friction, const T* __restrict__ lengths,
                                                         const T* __restrict__ x, long long ld_x, const T* __restrict__ xp,
                                                         long long ld_xp, long long batch, const T* __restrict__ weights,
                                                         double scale, T* __restrict__ loss, T* __restrict__ force,
                                                         int* __restrict__ iters, double* __restrict__ partials, int want_grad,
                                                         const T* __restrict__ wit, T* __restrict__ rbar,
                                                         const double* __restrict__ pdirs,
                                                         const T* __restrict__ u = nullptr, long long ld_u = 0) {
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
    load_actuation<T, NJ>(md, u ? u + it * ld_u : (const T*)nullptr, dp);
    T xr[D::NX], xpr[D::NX];
#pragma unroll
    for (int i = 0; i < D::NX; ++i) { xr[i] = x[it * ld_x + i]; xpr[i] = xp[it * ld_xp + i]; }
    const T w = valid ? T(scale) * (weights ? weights[it] : T(1)) : T(0);
    T f[1][3];
    int n_it = 0;
    T L;
    if constexpr (MESH) {
      using TA = typename Acc<T>::type;
      T wt[1][3], rb[1][3] = {{T(0), T(0), T(0)}}, rba[1][3] = {{T(0), T(0), T(0)}};
      MeshPairIn<T, TA, 1> in;
      load_mesh_inputs<T, TA>(wit, pdirs, it, cidx, wt, in);
      L = loss_item<T, TA, NJ, 1, GenLanes>(md, dp, opt, xr, xpr, cidx, w, want_grad != 0, acc, f, n_it, wt, rb, &in, rba);
      if (rbar && valid) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          rbar[it * kWitPerItem + (cidx * 2 + 0) * 3 + i] = rb[0][i];
          rbar[it * kWitPerItem + (cidx * 2 + 1) * 3 + i] = rba[0][i];
        }
      }
    } else {
      L = loss_item<T, typename Acc<T>::type, NJ, 1, GenLanes>(md, dp, opt, xr, xpr, cidx, w, want_grad != 0, acc, f, n_it);
    }
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
// adam (fused training step, dpll_contactnets_train_step): the thread that writes gradient entry k also applies Adam to
// parameter k -- entries of the flat buffer that are padding (unused geometry slots, the tail of a geometry's block) are left alone
template <typename T, int NJ>
__global__ __launch_bounds__(256) void gen_finalize_kernel(const double* __restrict__ folded, int n_rows, const double* __restrict__ chain,
                                                           T* __restrict__ grad, T* __restrict__ loss_total, AdamArgs adam, GeneralDesc md) {
  using D = GD<T, NJ>;
  static_assert(D::PI <= 256, "row must fit 256 columns");
  __shared__ double tot[256];
  const int col = threadIdx.x;
  double s = 0.0;
  if (col < D::PIOTA)
    for (int r = 0; r < n_rows; ++r) s += folded[(long long)r * D::PI + col];
  tot[col] = s;
  double steps = 0.0, pow1 = 0.0, pow2 = 0.0;
  if (adam.params) adam_powers(adam, steps, pow1, pow2);
  __syncthreads();  // (column totals in place; every thread has read the optimizer state before thread 0 advances it)
  if (threadIdx.x < D::PI) {
    const int k = (int)threadIdx.x - 1;
    const double v = k < 0 ? tot[0] : apply_chain<D::NB, kNG, kGP>(tot, chain, k);
    if (k < 0) {
      if (loss_total) *loss_total = T(v);
      if (adam.params) { adam.state[0] = steps; adam.state[1] = pow1; adam.state[2] = pow2; }
    } else {
      grad[k] = T(v);
      if (adam.params && general_param_is_real(md, k)) adam_apply<T>(adam, k, double(T(v)), pow1, pow2);
    }
  }
}

// MESH: one step per launch (the support points belong to the state the step starts from)
template <typename T, int NJ, bool MESH = false>
__global__ __launch_bounds__(kWave) void gen_simulate_kernel(GeneralDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                             const T* __restrict__ friction, const T* __restrict__ lengths,
                                                             const T* __restrict__ x0, long long ld_x, long long batch,
                                                             long long steps, T* __restrict__ out, long long ld_item,
                                                             long long ld_step, int write_x0, int* __restrict__ iters,
                                                             const T* __restrict__ wit, const double* __restrict__ pdirs,
                                                             const T* __restrict__ u = nullptr, long long ld_u = 0) {
  using D = GD<T, NJ>;
  const int lane = threadIdx.x, cidx = lane % D::G, slot = lane / D::G;
  Derived<T, NJ, kNG> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  const long long stride = (long long)gridDim.x * kIPW;
  for (long long base = (long long)blockIdx.x * kIPW; base < batch; base += stride) {
    const long long item = base + slot;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    load_actuation<T, NJ>(md, u ? u + it * ld_u : (const T*)nullptr, dp);
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
      if constexpr (MESH) {
        using TA = typename Acc<T>::type;
        T wt[1][3];
        MeshPairIn<T, TA, 1> in;
        load_mesh_inputs<T, TA>(wit, pdirs, it, cidx, wt, in);
        step_item<T, TA, NJ, 1, GenLanes>(md, dp, opt, xr, cidx, xn, imp, n_it, wt, &in);
      } else {
        step_item<T, typename Acc<T>::type, NJ, 1, GenLanes>(md, dp, opt, xr, cidx, xn, imp, n_it);
      }
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
template <typename T, int NJ, bool MESH = false>
__global__ __launch_bounds__(kWave) void gen_step_backward_kernel(GeneralDesc md, SolverOpts opt, const T* __restrict__ theta,
                                                                  const T* __restrict__ friction, const T* __restrict__ lengths,
                                                                  const T* __restrict__ x, long long ld_x,
                                                                  const T* __restrict__ gx, long long ld_g, long long batch,
                                                                  double* __restrict__ partials, T* __restrict__ xbar_out,
                                                                  long long ld_xb, const T* __restrict__ wit, T* __restrict__ rbar,
                                                                  const double* __restrict__ pdirs,
                                                                  const T* __restrict__ u = nullptr, long long ld_u = 0) {
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
    {  // the item's actuation inputs, in the compute type
      C u_c[NJ > 0 ? NJ : 1];
#pragma unroll
      for (int k = 0; k < (NJ > 0 ? NJ : 1); ++k) u_c[k] = (u && k < md.n_u) ? C(u[it * ld_u + k]) : C(0);
      load_actuation<C, NJ>(md, u ? u_c : (const C*)nullptr, dp);
    }
    // (an idle group's seed is zero, so what it adds to the sums below is zero)
    if constexpr (MESH) {
      C wt[1][3], rb[1][3] = {{C(0), C(0), C(0)}}, rba[1][3] = {{C(0), C(0), C(0)}};
      MeshPairIn<C, C, 1> in;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        wt[0][i] = C(wit[it * kWitPerItem + (cidx * 2 + 0) * 3 + i]);
        in.wit_a[0][i] = C(wit[it * kWitPerItem + (cidx * 2 + 1) * 3 + i]);
      }
#pragma unroll
      for (int p = 0; p < kMaxPairs; ++p)
#pragma unroll
        for (int i = 0; i < 3; ++i) in.dirs[p][i] = pdirs[(it * kMaxPairs + p) * 3 + i];
      step_item_backward<C, C, NJ, 1, GenLanes>(md, dp, opt, xr, cidx, gr, acc, wt, rb, &xb, &in, rba);
      if (valid) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          rbar[it * kWitPerItem + (cidx * 2 + 0) * 3 + i] = T(rb[0][i]);
          rbar[it * kWitPerItem + (cidx * 2 + 1) * 3 + i] = T(rba[0][i]);
        }
      }
    } else {
      step_item_backward<C, C, NJ, 1, GenLanes>(md, dp, opt, xr, cidx, gr, acc, nullptr, nullptr, &xb);
    }
    if (xbar_out && valid && cidx == 0) {
#pragma unroll
      for (int i = 0; i < D::NX; ++i) xbar_out[it * ld_xb + i] = T(xb[i]);
    }
  }
  store_iota_row<C, NJ, D::G, kNG, kGP>(acc, 0.0, partials);
}

// MultibodyTerms.forward (multibody_terms.py:584-609) over all kMaxGeoms x 4 contact slots; the host keeps the real ones
template <typename T, int NJ, bool MESH = false>
__global__ __launch_bounds__(kWave) void gen_terms_kernel(GeneralDesc md, const T* __restrict__ theta, const T* __restrict__ friction,
                                                          const T* __restrict__ lengths, const T* __restrict__ x, long long ld_x,
                                                          long long batch, T* __restrict__ Dout, T* __restrict__ Mout,
                                                          T* __restrict__ Jout, T* __restrict__ phiout, T* __restrict__ aout,
                                                          const T* __restrict__ wit, const double* __restrict__ pdirs,
                                                          const T* __restrict__ u = nullptr, long long ld_u = 0) {
  using D = GD<T, NJ>;
  constexpr int NV = D::NV, K = D::K;
  Derived<T, NJ, kNG> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  const long long it = (long long)blockIdx.x * kWave + threadIdx.x;
  if (it >= batch) return;
  load_actuation<T, NJ>(md, u ? u + it * ld_u : (const T*)nullptr, dp);
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
    if constexpr (MESH) {
      using TA = typename Acc<T>::type;
      T wb[3], wa[3];
      TA dir[3] = {TA(0), TA(0), TA(1)};
      for (int i = 0; i < 3; ++i) {
        wb[i] = wit[it * kWitPerItem + (c * 2 + 0) * 3 + i];
        wa[i] = wit[it * kWitPerItem + (c * 2 + 1) * 3 + i];
      }
      const int pp = c - kQuery * kMaxGeoms;
      if (pp >= 0 && pp < kMaxPairs)
        for (int i = 0; i < 3; ++i) dir[i] = TA(pdirs[(it * kMaxPairs + pp) * 3 + i]);
      // (candidates between other kinds of geometry are searched inside, one lane: dir = nullptr for them)
      const bool mesh_pair = pp >= 0 && pp < md.n_pairs && md.geom_kind[md.pair_a[pp < kMaxPairs ? pp : 0]] == kGeomMesh;
      compute_contact<T, TA, NJ>(md, dp, t.kin, kinA, c, cg, wb, mesh_pair ? dir : nullptr, wa);
    } else {
      compute_contact<T, typename Acc<T>::type, NJ>(md, dp, t.kin, kinA, c, cg);
    }
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
int finalize_rows(double* workspace, int rows, T* grad, T* loss_total, hipStream_t stream, const dpll_model* m = nullptr,
                  const AdamArgs* adam = nullptr) {
  using D = GD<T, NJ>;
  double* chain = workspace + (long long)rows * D::PI;
  double* folded = chain + D::CHAIN;
  const int n_folded = (int)folded_rows(rows);
  if (n_folded > 0) {
    hipLaunchKernelGGL((gen_fold_rows_kernel<T, NJ>), dim3(n_folded), dim3(256), 0, stream, (const double*)workspace, rows, folded);
    if (int rc = dpll_check_launch("gen_fold_rows_kernel")) return rc;
  }
  hipLaunchKernelGGL((gen_finalize_kernel<T, NJ>), dim3(1), dim3(256), 0, stream, (const double*)folded, n_folded, (const double*)chain,
                     grad, loss_total, (adam && m) ? *adam : AdamArgs{}, m ? general_desc(m) : GeneralDesc{});
  return dpll_check_launch("gen_finalize_kernel");
}


}  // namespace
