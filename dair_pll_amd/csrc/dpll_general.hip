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

#include "dpll_general_kernels.hpp"
#include "dpll_general.hpp"

namespace {

template <typename T, int NJ>
int launch_loss(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp,
                long long ld_xp, long long batch, const void* weights, double scale, void* loss, void* grad, void* loss_total,
                void* force, int32_t* iters, void* workspace, long long workspace_bytes, hipStream_t stream, const AdamArgs* adam) {
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
                     batch, (const T*)weights, scale, (T*)loss, (T*)force, (int*)iters, (double*)workspace, want_grad,
                     (const T*)nullptr, (T*)nullptr, (const double*)nullptr, (const T*)p->u, (long long)p->ld_u);
  if (int rc = dpll_check_launch("gen_loss_kernel")) return rc;
  if (want_grad) return finalize_rows<T, NJ>((double*)workspace, rows, (T*)grad, (T*)loss_total, stream, m, adam);
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
                     ld_item, ld_step, write_x0, (int*)iters, (const T*)nullptr, (const double*)nullptr,
                     steps == 1 ? (const T*)p->u : (const T*)nullptr, (long long)p->ld_u);  // (rollouts run unactuated, as sim_step does)
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
                     (double*)workspace, (T*)grad_x, ld_gx, (const T*)nullptr, (T*)nullptr, (const double*)nullptr, (const T*)p->u,
                     (long long)p->ld_u);
  if (int rc = dpll_check_launch("gen_step_backward_kernel")) return rc;
  return finalize_rows<T, NJ>((double*)workspace, rows, (T*)grad, (T*)nullptr, stream);
}

template <typename T, int NJ>
int launch_terms(const dpll_model* m, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm, void* M,
                 void* J, void* phi, void* a, hipStream_t stream) {
  if (!J) return dpll_fail(-1, "dpll_terms (general build): the J output is required%s");
  const long long blocks = (batch + kWave - 1) / kWave;
  hipLaunchKernelGGL((gen_terms_kernel<T, NJ>), dim3((int)blocks), dim3(kWave), 0, stream, general_desc(m), (const T*)p->theta,
                     (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, batch, (T*)Dm, (T*)M, (T*)J, (T*)phi, (T*)a,
                     (const T*)nullptr, (const double*)nullptr, (const T*)p->u, (long long)p->ld_u);
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
         int32_t* iters, void* workspace, long long ws_bytes, hipStream_t stream, const AdamArgs* adam) {
  DPLL_GEN_DISPATCH(launch_loss, m, dtype, p, x, ld_x, xp, ld_xp, batch, weights, scale, loss_out, grad, loss_total, force, iters,
                    workspace, ws_bytes, stream, adam);
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
