// TEST INFRASTRUCTURE ONLY -- never loaded by dair_pll_amd (the product path is the HIP library and
// fails loudly without it).  This compiles dair_pll_amd/csrc/dpll_core.hpp for the host with ONE lane
// per item (G = 1) so that the per-item math the GPU kernels run can be checked on CPU against the
// oracle, and run under -fsanitize=address,undefined (GPU sanitizers are not available on the pool).
#include <cstdint>
#include <vector>

#include "../../dair_pll_amd/csrc/dpll_core.hpp"

using namespace dpll;

namespace {

template <typename T, typename TA, int NJ>
void loss_batch(const ModelDesc& md, const SolverOpts& opt, const T* theta, const T* friction, const T* lengths,
                const T* x, const T* xp, int64_t B, const T* weights, double scale, T* loss, double* grad, T* force,
                int32_t* iters) {
  constexpr int NB = NJ + 1, K = kQuery * NB, NX = 13 + 2 * NJ;
  Derived<T, NJ> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  double g_iota[NB][kIota] = {}, g_mu[NB] = {}, g_len[NB][3] = {};
  for (int64_t i = 0; i < B; ++i) {
    LossGrad<T, NJ> g;
    zero_grad(g);
    T f[K][3];
    int it = 0;
    const T w = T(scale) * (weights ? weights[i] : T(1));
    loss[i] = loss_item<T, TA, NJ, K, OneLane>(md, dp, opt, x + i * NX, xp + i * NX, 0, w, grad != nullptr, g, f, it);
    if (iters) iters[i] = it;
    if (force)
      for (int c = 0; c < K; ++c) {
        // reference ordering: normals first, then (t_x, t_y) interleaved per contact (multibody_terms.py:415-426)
        force[i * 3 * K + c] = f[c][2];
        force[i * 3 * K + K + 2 * c] = f[c][0];
        force[i * 3 * K + K + 2 * c + 1] = f[c][1];
      }
    for (int b = 0; b < NB; ++b) {
      for (int k = 0; k < kIota; ++k) g_iota[b][k] += double(g.g_iota[b][k]);
      g_mu[b] += double(g.g_mu[b]);
      for (int k = 0; k < 3; ++k) g_len[b][k] += double(g.g_len[b][k]);
    }
  }
  if (!grad) return;
  double th[NB * 10], fr[NB + 1], ln[NB * 3];
  for (int i = 0; i < NB * 10; ++i) th[i] = double(theta[i]);
  for (int i = 0; i < NB + 1; ++i) fr[i] = double(friction[i]);
  for (int i = 0; i < NB * 3; ++i) ln[i] = double(lengths[i]);
  for (int b = 0; b < NB; ++b)
    for (int k = 0; k < 10; ++k) grad[b * 10 + k] = theta_grad_component(md.inertia_mode, th + 10 * b, g_iota[b], k);
  for (int k = 0; k < NB + 1; ++k) grad[NB * 10 + k] = friction_grad_component(NB, fr, g_mu, k);
  for (int k = 0; k < NB * 3; ++k) grad[NB * 10 + NB + 1 + k] = length_grad_component(ln, &g_len[0][0], k);
}

template <typename T, typename TA, int NJ>
void step_batch(const ModelDesc& md, const SolverOpts& opt, const T* theta, const T* friction, const T* lengths,
                const T* x, int64_t B, T* x_next, int32_t* iters) {
  constexpr int NB = NJ + 1, K = kQuery * NB, NX = 13 + 2 * NJ;
  Derived<T, NJ> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  for (int64_t i = 0; i < B; ++i) {
    T imp[K][3];
    int it = 0;
    step_item<T, TA, NJ, K, OneLane>(md, dp, opt, x + i * NX, 0, x_next + i * NX, imp, it);
    if (iters) iters[i] = it;
  }
}

}  // namespace

extern "C" {

int hostsim_loss_f64(const ModelDesc* md, const SolverOpts* opt, const double* theta, const double* friction,
                     const double* lengths, const double* x, const double* xp, int64_t B, const double* weights,
                     double scale, double* loss, double* grad, double* force, int32_t* iters) {
  if (md->n_joints == 0) loss_batch<double, double, 0>(*md, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
  else if (md->n_joints == 1) loss_batch<double, double, 1>(*md, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
  else return -1;
  return 0;
}

// mixed != 0: the cone residual and y are carried in double (what the float GPU kernels do)
int hostsim_loss_f32(const ModelDesc* md, const SolverOpts* opt, const float* theta, const float* friction,
                     const float* lengths, const float* x, const float* xp, int64_t B, const float* weights,
                     double scale, float* loss, double* grad, float* force, int32_t* iters, int mixed) {
  if (md->n_joints == 0) {
    if (mixed) loss_batch<float, double, 0>(*md, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
    else loss_batch<float, float, 0>(*md, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
  } else if (md->n_joints == 1) {
    if (mixed) loss_batch<float, double, 1>(*md, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
    else loss_batch<float, float, 1>(*md, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
  } else return -1;
  return 0;
}

int hostsim_step_f64(const ModelDesc* md, const SolverOpts* opt, const double* theta, const double* friction,
                     const double* lengths, const double* x, int64_t B, double* x_next, int32_t* iters) {
  if (md->n_joints == 0) step_batch<double, double, 0>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
  else if (md->n_joints == 1) step_batch<double, double, 1>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
  else return -1;
  return 0;
}

int hostsim_step_f32(const ModelDesc* md, const SolverOpts* opt, const float* theta, const float* friction,
                     const float* lengths, const float* x, int64_t B, float* x_next, int32_t* iters, int mixed) {
  if (md->n_joints == 0) {
    if (mixed) step_batch<float, double, 0>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
    else step_batch<float, float, 0>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
  } else if (md->n_joints == 1) {
    if (mixed) step_batch<float, double, 1>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
    else step_batch<float, float, 1>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
  } else return -1;
  return 0;
}

int hostsim_sizeof_model_desc() { return (int)sizeof(ModelDesc); }
}
