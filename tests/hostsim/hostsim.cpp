// TEST INFRASTRUCTURE ONLY -- never loaded by dair_pll_amd (the product path is the HIP library and
// fails loudly without it).  This compiles dair_pll_amd/csrc/dpll_core.hpp for the host with ONE lane
// per item (G = 1) so that the per-item math the GPU kernels run can be checked on CPU against the
// oracle, and run under -fsanitize=address,undefined (GPU sanitizers are not available on the pool).
#include <cstdint>
#include <vector>

// solver statistics for tools/diag/host_iters.py: bit `it` of an item's mask is set when iteration `it` did not take
// the full Newton step (alpha != 1)
static thread_local uint64_t g_reject_mask = 0;
static std::vector<uint64_t> g_reject_masks;
#define DPLL_ITER_HOOK(it, active, alpha) do { if ((active) && !((alpha) == 1) && (it) < 64) g_reject_mask |= (1ull << (it)); } while (0)

#include "../../dair_pll_amd/csrc/dpll_core.hpp"
#include "../../dair_pll_amd/csrc/dpll_weld.hpp"
#include "../../dair_pll_amd/csrc/dpll_icnn.hpp"

using namespace dpll;

// actuation inputs of the next calls (general models with actuators): (B, n_u) doubles, or null -- set through
// hostsim_set_actuation so that the batch entry points keep their argument lists
static const double* g_actuation = nullptr;
static int64_t g_actuation_ld = 0;

namespace {
template <typename T, int NJ, int NG, class MD> void item_actuation(const MD& md, int64_t item, Derived<T, NJ, NG>& dp) {
  T row[NJ > 0 ? NJ : 1] = {};
  if (g_actuation)
    for (int k = 0; k < NJ && k < md.n_u; ++k) row[k] = T(g_actuation[item * g_actuation_ld + k]);
  load_actuation<T, NJ>(md, g_actuation ? row : (const T*)nullptr, dp);
}

// the rotation the inertial parameters of body b go through (general models whose frames are turned), or nullptr
template <class MD> const double (*body_rot_of(const MD& md, int b))[3][3] {
  if constexpr (MD::kGeneral) return (md.rotated & 1) ? &md.body_rot[b] : nullptr;
  return nullptr;
}

template <typename T, typename TA, int NJ, int NG = NJ + 1, class MD = ModelDesc>
void loss_batch(const MD& md, const SolverOpts& opt, const T* theta, const T* friction, const T* lengths,
                const T* x, const T* xp, int64_t B, const T* weights, double scale, T* loss, double* grad, T* force,
                int32_t* iters) {
  constexpr int NB = NJ + 1, K = kQuery * NG, NX = 13 + 2 * NJ;
  Derived<T, NJ, NG> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  constexpr int GP = MD::kGeoStride;
  double g_iota[NB][kIota] = {}, g_mu[NG] = {}, g_len[NG][GP] = {};
  for (int64_t i = 0; i < B; ++i) {
    LossGrad<T, NJ, NG, GP> g;
    zero_grad(g);
    T f[K][3];
    int it = 0;
    const T w = T(scale) * (weights ? weights[i] : T(1));
    g_reject_mask = 0;
    item_actuation<T, NJ>(md, i, dp);
    loss[i] = loss_item<T, TA, NJ, K, OneLane>(md, dp, opt, x + i * NX, xp + i * NX, 0, w, grad != nullptr, g, f, it);
    if ((int64_t)g_reject_masks.size() < B) g_reject_masks.resize(B);
    g_reject_masks[i] = g_reject_mask;
    if (iters) iters[i] = it;
    if (force)
      for (int c = 0; c < K; ++c) {
        // reference ordering: normals first, then (t_x, t_y) interleaved per contact (multibody_terms.py:415-426)
        force[i * 3 * K + c] = f[c][2];
        force[i * 3 * K + K + 2 * c] = f[c][0];
        force[i * 3 * K + K + 2 * c + 1] = f[c][1];
      }
    for (int b = 0; b < NB; ++b)
      for (int k = 0; k < kIota; ++k) g_iota[b][k] += double(g.g_iota[b][k]);
    for (int b = 0; b < NG; ++b) {
      g_mu[b] += double(g.g_mu[b]);
      for (int k = 0; k < GP; ++k) g_len[b][k] += double(g.g_len[b][k]);
    }
  }
  if (!grad) return;
  double th[NB * 10], fr[NG + 1], ln[NG * GP];
  for (int i = 0; i < NB * 10; ++i) th[i] = double(theta[i]);
  for (int i = 0; i < NG + 1; ++i) fr[i] = double(friction[i]);
  for (int i = 0; i < NG * GP; ++i) ln[i] = double(lengths[i]);
  for (int b = 0; b < NB; ++b)
    for (int k = 0; k < 10; ++k) grad[b * 10 + k] = theta_grad_component(md.inertia_mode, th + 10 * b, g_iota[b], k, body_rot_of(md, b));
  for (int k = 0; k < NG + 1; ++k) grad[NB * 10 + k] = friction_grad_component(NG, fr, g_mu, k, MD::kGeneral ? &md : nullptr,
                                                                                  MD::kGeneral ? g_len[NG > kMaxGeoms ? kMaxGeoms : 0] : nullptr);
  for (int k = 0; k < NG * GP; ++k) {  // a polygon's vertices are signed parameters; lengths and radii enter through |.|
    const bool polygon = MD::kGeneral && k / GP < kMaxGeoms && md.geom_kind[k / GP < kMaxGeoms ? k / GP : 0] == kGeomPolygon;
    grad[NB * 10 + NG + 1 + k] = polygon ? (&g_len[0][0])[k] : length_grad_component(ln, &g_len[0][0], k);
    if (MD::kGeneral && k / GP >= kMaxGeoms) grad[NB * 10 + NG + 1 + k] = 0.0;  // (no parameters there: the pairs' d/d mu)
  }
}

template <typename T, typename TA, int NJ, int NG = NJ + 1, class MD = ModelDesc>
void step_batch(const MD& md, const SolverOpts& opt, const T* theta, const T* friction, const T* lengths,
                const T* x, int64_t B, T* x_next, int32_t* iters) {
  constexpr int K = kQuery * NG, NX = 13 + 2 * NJ;
  Derived<T, NJ, NG> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  for (int64_t i = 0; i < B; ++i) {
    T imp[K][3];
    int it = 0;
    item_actuation<T, NJ>(md, i, dp);
    step_item<T, TA, NJ, K, OneLane>(md, dp, opt, x + i * NX, 0, x_next + i * NX, imp, it);
    if (iters) iters[i] = it;
  }
}


// ---- mesh (DeepSupportConvex) reference path: plain loops, cube only (one body) --------------------
template <typename T> struct IcnnEval {
  T q[3], m0[kIcnnWidth], m1[kIcnnWidth], u1[kIcnnWidth], u0[kIcnnWidth], p[3];
};
template <typename T> void icnn_forward_ref(const IcnnWeights<T>& w, const T* quat, int s, IcnnEval<T>& e) {
  constexpr int W = kIcnnWidth;
  icnn_query<T>(quat, w.pert + 3 * s, e.q);
  T h0[W];
  for (int k = 0; k < W; ++k) {
    const T pre = e.q[0] * w.Wd0[k] + e.q[1] * w.Wd0[W + k] + e.q[2] * w.Wd0[2 * W + k];
    h0[k] = icnn_act(pre);
    e.m0[k] = icnn_mask(pre);
  }
  for (int j = 0; j < W; ++j) {
    T pre = e.q[0] * w.Wd1[j] + e.q[1] * w.Wd1[W + j] + e.q[2] * w.Wd1[2 * W + j];
    for (int k = 0; k < W; ++k) pre += h0[k] * tabs(w.Wh[k * W + j]);
    e.m1[j] = icnn_mask(pre);
    e.u1[j] = tabs(w.wout[j]) * e.m1[j];
  }
  for (int k = 0; k < W; ++k) {
    T v = T(0);
    for (int j = 0; j < W; ++j) v += tabs(w.Wh[k * W + j]) * e.u1[j];
    e.u0[k] = v * e.m0[k];
  }
  for (int i = 0; i < 3; ++i) {
    T s1 = T(0);
    for (int j = 0; j < W; ++j) s1 += w.Wd1[i * W + j] * e.u1[j] + w.Wd0[i * W + j] * e.u0[j];
    e.p[i] = s1;
  }
}
// accumulates d/d(Wh, Wd0, Wd1, wout) (double) for upstream pbar
template <typename T> void icnn_backward_ref(const IcnnWeights<T>& w, const IcnnEval<T>& e, const T* pbar, double* gWh,
                                             double* gWd0, double* gWd1, double* gwout) {
  constexpr int W = kIcnnWidth;
  double vbar[W], u1bar[W];
  for (int k = 0; k < W; ++k) {
    const double u0bar = double(w.Wd0[k]) * pbar[0] + double(w.Wd0[W + k]) * pbar[1] + double(w.Wd0[2 * W + k]) * pbar[2];
    vbar[k] = u0bar * double(e.m0[k]);
    for (int i = 0; i < 3; ++i) { gWd0[i * W + k] += double(pbar[i]) * double(e.u0[k]); gWd1[i * W + k] += double(pbar[i]) * double(e.u1[k]); }
  }
  for (int j = 0; j < W; ++j) u1bar[j] = double(w.Wd1[j]) * pbar[0] + double(w.Wd1[W + j]) * pbar[1] + double(w.Wd1[2 * W + j]) * pbar[2];
  for (int k = 0; k < W; ++k)
    for (int j = 0; j < W; ++j) {
      const double a = double(tabs(w.Wh[k * W + j]));
      const double sg = w.Wh[k * W + j] > T(0) ? 1.0 : (w.Wh[k * W + j] < T(0) ? -1.0 : 0.0);
      gWh[k * W + j] += vbar[k] * double(e.u1[j]) * sg;
      u1bar[j] += a * vbar[k];
    }
  for (int j = 0; j < W; ++j) {
    const double sg = w.wout[j] > T(0) ? 1.0 : (w.wout[j] < T(0) ? -1.0 : 0.0);
    gwout[j] += u1bar[j] * double(e.m1[j]) * sg;
  }
}

template <typename T, typename TA>
void mesh_loss_batch(const ModelDesc& md, const SolverOpts& opt, const T* theta, const T* friction,
                     const IcnnWeights<T>& w, const T* x, const T* xp, int64_t B, double scale, T* loss, double* grad,
                     T* x_next) {
  constexpr int NJ = 0, K = 4, NX = 13, W = kIcnnWidth;
  Derived<T, NJ> dp;
  const T zero_len[3] = {T(0), T(0), T(0)};
  derive_params<T, NJ>(md, theta, friction, zero_len, dp);
  double g_iota[kIota] = {}, g_mu[1] = {};
  std::vector<double> gWh(W * W, 0.0), gWd0(3 * W, 0.0), gWd1(3 * W, 0.0), gwout(W, 0.0);
  std::vector<IcnnEval<T>> ev(K);
  for (int64_t i = 0; i < B; ++i) {
    T wit[K][3], rbar[K][3];
    if (loss) {
      for (int s = 0; s < K; ++s) { icnn_forward_ref(w, xp + i * NX, s, ev[s]); for (int a = 0; a < 3; ++a) wit[s][a] = ev[s].p[a]; }
      LossGrad<T, NJ> g;
      zero_grad(g);
      T f[K][3];
      int it = 0;
      loss[i] = loss_item<T, TA, NJ, K, OneLane>(md, dp, opt, x + i * NX, xp + i * NX, 0, T(scale), grad != nullptr, g, f, it, wit, rbar);
      if (grad) {
        for (int k = 0; k < kIota; ++k) g_iota[k] += double(g.g_iota[0][k]);
        g_mu[0] += double(g.g_mu[0]);
        for (int s = 0; s < K; ++s) icnn_backward_ref(w, ev[s], rbar[s], gWh.data(), gWd0.data(), gWd1.data(), gwout.data());
      }
    }
    if (x_next) {
      for (int s = 0; s < K; ++s) { icnn_forward_ref(w, x + i * NX, s, ev[s]); for (int a = 0; a < 3; ++a) wit[s][a] = ev[s].p[a]; }
      T imp[K][3];
      int it = 0;
      step_item<T, TA, NJ, K, OneLane>(md, dp, opt, x + i * NX, 0, x_next + i * NX, imp, it, wit);
    }
  }
  if (!grad) return;
  double th[10], fr[2];
  for (int i = 0; i < 10; ++i) th[i] = double(theta[i]);
  for (int i = 0; i < 2; ++i) fr[i] = double(friction[i]);
  for (int k = 0; k < 10; ++k) grad[k] = theta_grad_component(md.inertia_mode, th, g_iota, k);
  for (int k = 0; k < 2; ++k) grad[10 + k] = friction_grad_component(1, fr, g_mu, k);
  double* out = grad + 12;  // [Wh | Wd0 | Wd1 | wout]
  for (int i = 0; i < W * W; ++i) out[i] = gWh[i];
  for (int i = 0; i < 3 * W; ++i) { out[W * W + i] = gWd0[i]; out[W * W + 3 * W + i] = gWd1[i]; }
  for (int i = 0; i < W; ++i) out[W * W + 6 * W + i] = gwout[i];
}


template <typename T, typename TA, int NJ, int NG = NJ + 1, class MD = ModelDesc>
void step_backward_batch(const MD& md, const SolverOpts& opt, const T* theta, const T* friction, const T* lengths,
                         const T* x, const T* xbar_next, int64_t B, double* grad, T* xbar) {
  constexpr int NB = NJ + 1, K = kQuery * NG, NX = 13 + 2 * NJ;
  Derived<T, NJ, NG> dp;
  derive_params<T, NJ>(md, theta, friction, lengths, dp);
  constexpr int GP = MD::kGeoStride;
  double g_iota[NB][kIota] = {}, g_mu[NG] = {}, g_len[NG][GP] = {};
  for (int64_t i = 0; i < B; ++i) {
    LossGrad<T, NJ, NG, GP> g;
    zero_grad(g);
    T xb[NX] = {};
    item_actuation<T, NJ>(md, i, dp);
    step_item_backward<T, TA, NJ, K, OneLane>(md, dp, opt, x + i * NX, 0, xbar_next + i * NX, g, nullptr, nullptr,
                                              xbar ? &xb : nullptr);
    if (xbar) for (int k = 0; k < NX; ++k) xbar[i * NX + k] = xb[k];
    for (int b = 0; b < NB; ++b)
      for (int k = 0; k < kIota; ++k) g_iota[b][k] += double(g.g_iota[b][k]);
    for (int b = 0; b < NG; ++b) {
      g_mu[b] += double(g.g_mu[b]);
      for (int k = 0; k < GP; ++k) g_len[b][k] += double(g.g_len[b][k]);
    }
  }
  double th[NB * 10], fr[NG + 1], ln[NG * GP];
  for (int i = 0; i < NB * 10; ++i) th[i] = double(theta[i]);
  for (int i = 0; i < NG + 1; ++i) fr[i] = double(friction[i]);
  for (int i = 0; i < NG * GP; ++i) ln[i] = double(lengths[i]);
  for (int b = 0; b < NB; ++b)
    for (int k = 0; k < 10; ++k) grad[b * 10 + k] = theta_grad_component(md.inertia_mode, th + 10 * b, g_iota[b], k, body_rot_of(md, b));
  for (int k = 0; k < NG + 1; ++k) grad[NB * 10 + k] = friction_grad_component(NG, fr, g_mu, k, MD::kGeneral ? &md : nullptr,
                                                                                  MD::kGeneral ? g_len[NG > kMaxGeoms ? kMaxGeoms : 0] : nullptr);
  for (int k = 0; k < NG * GP; ++k) {  // a polygon's vertices are signed parameters; lengths and radii enter through |.|
    const bool polygon = MD::kGeneral && k / GP < kMaxGeoms && md.geom_kind[k / GP < kMaxGeoms ? k / GP : 0] == kGeomPolygon;
    grad[NB * 10 + NG + 1 + k] = polygon ? (&g_len[0][0])[k] : length_grad_component(ln, &g_len[0][0], k);
    if (MD::kGeneral && k / GP >= kMaxGeoms) grad[NB * 10 + NG + 1 + k] = 0.0;  // (no parameters there: the pairs' d/d mu)
  }
}

// general models (tree topology, geometry table): always three geometry slots, the unused ones masked
template <typename T, typename TA, class F0, class F1, class F2, class F3>
int general_dispatch(const ModelDesc* md, F0 f0, F1 f1, F2 f2, F3 f3) {
  const GeneralDesc& gd = *static_cast<const GeneralDesc*>(md);
  if (md->n_joints == 0) f0(gd);
  else if (md->n_joints == 1) f1(gd);
  else if (md->n_joints == 2) f2(gd);
  else if (md->n_joints == 3) f3(gd);
  else return -1;
  return 0;
}

}  // namespace

extern "C" {
// actuation inputs (B, n_u) of the next loss / step / step-backward calls on an actuated general model; null = none
void hostsim_set_actuation(const double* u, int64_t ld_u) { g_actuation = u; g_actuation_ld = ld_u; }

// dpll_weld_compose / dpll_weld_compose_backward on the host: the entries the device kernels compute, one thread each
// (dair_pll_amd/csrc/dpll_weld.hpp)
void hostsim_weld_compose(int inertia_mode, int n_rows, int n_bodies, const int32_t* host, const double* X, const double* theta,
                          double* iota) {
  for (int t = 0; t < n_bodies * kIota; ++t) iota[t] = weld_compose_entry<double>(inertia_mode, n_rows, host, X, theta, t / kIota, t % kIota);
}
void hostsim_weld_backward(int inertia_mode, int n_rows, const int32_t* host, const double* X, const double* theta,
                           const double* grad_iota, double* grad_theta) {
  for (int t = 0; t < n_rows * 10; ++t) grad_theta[t] = weld_backward_entry<double>(inertia_mode, host, X, theta, grad_iota, t / 10, t % 10);
}


int hostsim_loss_f64(const ModelDesc* md, const SolverOpts* opt, const double* theta, const double* friction,
                     const double* lengths, const double* x, const double* xp, int64_t B, const double* weights,
                     double scale, double* loss, double* grad, double* force, int32_t* iters) {
  if (md->n_geoms > 0)
    return general_dispatch<double, double>(
        md, [&](const GeneralDesc& g) { loss_batch<double, double, 0, kGenSlots>(g, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters); },
        [&](const GeneralDesc& g) { loss_batch<double, double, 1, kGenSlots>(g, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters); },
        [&](const GeneralDesc& g) { loss_batch<double, double, 2, kGenSlots>(g, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters); },
        [&](const GeneralDesc& g) { loss_batch<double, double, 3, kGenSlots>(g, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters); });
  if (md->n_joints == 0) loss_batch<double, double, 0>(*md, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
  else if (md->n_joints == 1) loss_batch<double, double, 1>(*md, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
  else return -1;
  return 0;
}

// mixed != 0: the cone residual and y are carried in double (what the float GPU kernels do)
int hostsim_loss_f32(const ModelDesc* md, const SolverOpts* opt, const float* theta, const float* friction,
                     const float* lengths, const float* x, const float* xp, int64_t B, const float* weights,
                     double scale, float* loss, double* grad, float* force, int32_t* iters, int mixed) {
  if (md->n_geoms > 0)
    return general_dispatch<float, double>(
        md, [&](const GeneralDesc& g) { loss_batch<float, double, 0, kGenSlots>(g, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters); },
        [&](const GeneralDesc& g) { loss_batch<float, double, 1, kGenSlots>(g, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters); },
        [&](const GeneralDesc& g) { loss_batch<float, double, 2, kGenSlots>(g, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters); },
        [&](const GeneralDesc& g) { loss_batch<float, double, 3, kGenSlots>(g, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters); });
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
  if (md->n_geoms > 0)
    return general_dispatch<double, double>(
        md, [&](const GeneralDesc& g) { step_batch<double, double, 0, kGenSlots>(g, *opt, theta, friction, lengths, x, B, x_next, iters); },
        [&](const GeneralDesc& g) { step_batch<double, double, 1, kGenSlots>(g, *opt, theta, friction, lengths, x, B, x_next, iters); },
        [&](const GeneralDesc& g) { step_batch<double, double, 2, kGenSlots>(g, *opt, theta, friction, lengths, x, B, x_next, iters); },
        [&](const GeneralDesc& g) { step_batch<double, double, 3, kGenSlots>(g, *opt, theta, friction, lengths, x, B, x_next, iters); });
  if (md->n_joints == 0) step_batch<double, double, 0>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
  else if (md->n_joints == 1) step_batch<double, double, 1>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
  else return -1;
  return 0;
}

int hostsim_step_f32(const ModelDesc* md, const SolverOpts* opt, const float* theta, const float* friction,
                     const float* lengths, const float* x, int64_t B, float* x_next, int32_t* iters, int mixed) {
  if (md->n_geoms > 0)
    return general_dispatch<float, double>(
        md, [&](const GeneralDesc& g) { step_batch<float, double, 0, kGenSlots>(g, *opt, theta, friction, lengths, x, B, x_next, iters); },
        [&](const GeneralDesc& g) { step_batch<float, double, 1, kGenSlots>(g, *opt, theta, friction, lengths, x, B, x_next, iters); },
        [&](const GeneralDesc& g) { step_batch<float, double, 2, kGenSlots>(g, *opt, theta, friction, lengths, x, B, x_next, iters); },
        [&](const GeneralDesc& g) { step_batch<float, double, 3, kGenSlots>(g, *opt, theta, friction, lengths, x, B, x_next, iters); });
  if (md->n_joints == 0) {
    if (mixed) step_batch<float, double, 0>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
    else step_batch<float, float, 0>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
  } else if (md->n_joints == 1) {
    if (mixed) step_batch<float, double, 1>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
    else step_batch<float, float, 1>(*md, *opt, theta, friction, lengths, x, B, x_next, iters);
  } else return -1;
  return 0;
}

// mesh cube: weights = [Wh(256x256) | Wd0(3x256) | Wd1(3x256) | wout(256)], pert (4,3); grad layout [theta|friction|weights]
int hostsim_mesh_f64(const ModelDesc* md, const SolverOpts* opt, const double* theta, const double* friction,
                     const double* weights, const double* pert, const double* x, const double* xp, int64_t B,
                     double scale, double* loss, double* grad, double* x_next) {
  constexpr int W = kIcnnWidth;
  IcnnWeights<double> w{weights, weights + W * W, weights + W * W + 3 * W, weights + W * W + 6 * W, pert};
  mesh_loss_batch<double, double>(*md, *opt, theta, friction, w, x, xp, B, scale, loss, grad, x_next);
  return 0;
}
int hostsim_mesh_f32(const ModelDesc* md, const SolverOpts* opt, const float* theta, const float* friction,
                     const float* weights, const float* pert, const float* x, const float* xp, int64_t B,
                     double scale, float* loss, double* grad, float* x_next) {
  constexpr int W = kIcnnWidth;
  IcnnWeights<float> w{weights, weights + W * W, weights + W * W + 3 * W, weights + W * W + 6 * W, pert};
  mesh_loss_batch<float, double>(*md, *opt, theta, friction, w, x, xp, B, scale, loss, grad, x_next);
  return 0;
}
int hostsim_step_backward_f64(const ModelDesc* md, const SolverOpts* opt, const double* theta, const double* friction,
                              const double* lengths, const double* x, const double* xbar_next, int64_t B, double* grad,
                              double* xbar) {
  if (md->n_geoms > 0)
    return general_dispatch<double, double>(
        md, [&](const GeneralDesc& g) { step_backward_batch<double, double, 0, kGenSlots>(g, *opt, theta, friction, lengths, x, xbar_next, B, grad, xbar); },
        [&](const GeneralDesc& g) { step_backward_batch<double, double, 1, kGenSlots>(g, *opt, theta, friction, lengths, x, xbar_next, B, grad, xbar); },
        [&](const GeneralDesc& g) { step_backward_batch<double, double, 2, kGenSlots>(g, *opt, theta, friction, lengths, x, xbar_next, B, grad, xbar); },
        [&](const GeneralDesc& g) { step_backward_batch<double, double, 3, kGenSlots>(g, *opt, theta, friction, lengths, x, xbar_next, B, grad, xbar); });
  if (md->n_joints == 0) step_backward_batch<double, double, 0>(*md, *opt, theta, friction, lengths, x, xbar_next, B, grad, xbar);
  else if (md->n_joints == 1) step_backward_batch<double, double, 1>(*md, *opt, theta, friction, lengths, x, xbar_next, B, grad, xbar);
  else return -1;
  return 0;
}

int hostsim_sizeof_model_desc() { return (int)sizeof(ModelDesc); }

// the body-body direction search of csrc/dpll_core.hpp on two vertex sets given in one frame (kinds: dpll_geom_kind)
int hostsim_pair_direction(int kind_a, const double* va, int na, int kind_b, const double* vb, int nb, double* d) {
  if (na < 1 || nb < 1 || na > kMaxPolyVerts || nb > kMaxPolyVerts) return -1;
  double a[kMaxPolyVerts][3], b[kMaxPolyVerts][3], out[3];
  for (int i = 0; i < na; ++i) for (int c = 0; c < 3; ++c) a[i][c] = va[3 * i + c];
  for (int i = 0; i < nb; ++i) for (int c = 0; c < 3; ++c) b[i][c] = vb[3 * i + c];
  pair_direction<double, OneLane>(a, na, kind_a, b, nb, kind_b, out);
  for (int c = 0; c < 3; ++c) d[c] = out[c];
  return 0;
}
int64_t hostsim_reject_masks(uint64_t* out, int64_t n) {
  const int64_t m = n < (int64_t)g_reject_masks.size() ? n : (int64_t)g_reject_masks.size();
  for (int64_t i = 0; i < m; ++i) out[i] = g_reject_masks[i];
  return m;
}
}
