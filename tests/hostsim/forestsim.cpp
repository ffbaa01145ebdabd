// TEST INFRASTRUCTURE ONLY -- never loaded by dair_pll_amd.  Compiles dair_pll_amd/csrc/dpll_forest.hpp (the forest build's
// per-item program) for the host with a team of ONE, so that the math the GPU kernels run -- one wave per item -- can be held
// against the oracle and the reference-run fixtures in the CPU container, in float64 and float32.
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../dair_pll_amd/csrc/dpll_forest.hpp"

using namespace dpll_forest;

namespace {

// actuation inputs of the calls that follow (forestsim_set_actuation): row i = the B u inputs of item i, or none
const double* g_u = nullptr;
int64_t g_ld_u = 0;
const double* item_u(int64_t i) { return g_u ? g_u + i * g_ld_u : nullptr; }

template <typename S, typename SA> struct Item {
  std::vector<char> storage;
  Arena<S, SA> arena;
  explicit Item(const ForestDesc& fd, bool lite = false) : storage(arena_bytes<S, SA>(fd, lite) + 64) {
    arena.carve(storage.data(), fd.n_bodies, fd.n_v, fd.n_q, fd.n_contacts, fd.n_geoms, fd.n_pairs, lite);
  }
};

template <typename T>
int loss_batch(const ForestDesc& fd, const SolverOpts& opt, const T* theta, const T* friction, const T* lengths, const T* x, const T* xp,
               int64_t B, const T* weights, double scale, T* loss, double* grad, T* force, int32_t* iters) {
  Item<T, double> item(fd);
  Forest<T, double, HostTeam> prog(fd, item.arena);
  const int nx = fd.n_q + fd.n_v, K = fd.n_contacts;
  std::vector<double> row(row_width(fd), 0.0);
  prog.derive(theta, friction, lengths);
  for (int64_t i = 0; i < B; ++i) {
    int it = 0;
    const T w = T(scale) * (weights ? weights[i] : T(1));
    prog.load_actuation(item_u(i));
    const T L = prog.loss(x + i * nx, xp + i * nx, lengths, opt, w, grad != nullptr, row.data(), it);
    if (loss) loss[i] = L;
    if (iters) iters[i] = it;
    if (force)
      for (int c = 0; c < K; ++c) {  // reference ordering: normals, then (t_x, t_y) per contact (multibody_terms.py:415-426)
        force[i * 3 * K + c] = item.arena.force[3 * c + 2];
        force[i * 3 * K + K + 2 * c] = item.arena.force[3 * c];
        force[i * 3 * K + K + 2 * c + 1] = item.arena.force[3 * c + 1];
      }
  }
  if (grad)
    for (int k = 0; k < param_count(fd); ++k) grad[k] = chain_param(fd, theta, friction, lengths, row.data(), k);
  return 0;
}

template <typename T>
int step_batch(const ForestDesc& fd, const SolverOpts& opt, const T* theta, const T* friction, const T* lengths, const T* x, int64_t B,
               T* x_next, int32_t* iters) {
  Item<T, double> item(fd);
  Forest<T, double, HostTeam> prog(fd, item.arena);
  const int nx = fd.n_q + fd.n_v;
  prog.derive(theta, friction, lengths);
  for (int64_t i = 0; i < B; ++i) {
    prog.load_actuation(item_u(i));
    const int it = prog.step(x + i * nx, lengths, opt, x_next + i * nx);
    if (iters) iters[i] = it;
  }
  return 0;
}

// MultibodyTerms.forward (multibody_terms.py:584-609): M (nv, nv), a (nv), phi (K), J (3K, nv) rows [normals | mu (t_x, t_y) per contact]
template <typename T>
int terms_batch(const ForestDesc& fd, const T* theta, const T* friction, const T* lengths, const T* x, int64_t B, T* M, T* a, T* phi, T* J) {
  Item<T, double> item(fd);
  Forest<T, double, HostTeam> prog(fd, item.arena);
  const int nx = fd.n_q + fd.n_v, nv = fd.n_v, K = fd.n_contacts;
  prog.derive(theta, friction, lengths);
  for (int64_t i = 0; i < B; ++i) {
    prog.load_actuation(item_u(i));
    prog.load_state(x + i * nx);
    prog.terms();
    prog.contacts(lengths);
    for (int e = 0; e < nv * nv; ++e) M[i * nv * nv + e] = item.arena.M[e];
    for (int e = 0; e < nv; ++e) a[i * nv + e] = item.arena.a[e];
    for (int c = 0; c < K; ++c) {
      phi[i * K + c] = item.arena.ct[c].phi;
      const T* Jrow = item.arena.J + (size_t)c * 3 * nv;
      const T mu = item.arena.ct[c].mu;
      for (int e = 0; e < nv; ++e) {
        J[(i * 3 * K + c) * nv + e] = Jrow[2 * nv + e];
        J[(i * 3 * K + K + 2 * c) * nv + e] = mu * Jrow[e];
        J[(i * 3 * K + K + 2 * c + 1) * nv + e] = mu * Jrow[nv + e];
      }
    }
  }
  return 0;
}

// gradient of sum(xbar_next . x_next) with respect to [theta | friction | lengths] and (xbar != null) to the state
int step_backward_batch(const ForestDesc& fd, const SolverOpts& opt, const double* theta, const double* friction, const double* lengths,
                        const double* x, const double* xbar_next, int64_t B, double* grad, double* xbar) {
  Item<double, double> item(fd);
  Item<DualT<double>, DualT<double>> dual(fd, true);
  ForestBackward<HostTeam> back(fd, item.arena, dual.arena);
  Forest<double, double, HostTeam> prog(fd, item.arena);
  const int nx = fd.n_q + fd.n_v;
  std::vector<double> row(row_width(fd), 0.0);
  prog.derive(theta, friction, lengths);
  for (int64_t i = 0; i < B; ++i) {
    prog.load_actuation(item_u(i));
    back.run(x + i * nx, xbar_next + i * nx, theta, friction, lengths, opt, row.data(), xbar ? xbar + i * nx : (double*)nullptr);
  }
  for (int k = 0; k < param_count(fd); ++k) grad[k] = chain_param(fd, theta, friction, lengths, row.data(), k);
  return 0;
}

}  // namespace

extern "C" {
// the actuation inputs (B, ld) of the batch calls that follow; nullptr: none
void forestsim_set_actuation(const double* u, int64_t ld) { g_u = u; g_ld_u = ld; }
int forestsim_step_backward_f64(const ForestDesc* fd, const SolverOpts* opt, const double* theta, const double* friction, const double* lengths,
                                const double* x, const double* xbar_next, int64_t B, double* grad, double* xbar) {
  return step_backward_batch(*fd, *opt, theta, friction, lengths, x, xbar_next, B, grad, xbar);
}
int forestsim_sizeof_desc() { return (int)sizeof(ForestDesc); }
int forestsim_param_count(const ForestDesc* fd) { return param_count(*fd); }
// bytes of an item's arena: kind 0 float storage, 1 double, 2 the dual (lite) arena of the state adjoint
int64_t forestsim_arena_bytes(const ForestDesc* fd, int kind) {
  return kind == 0 ? (int64_t)arena_bytes<float, double>(*fd) : (kind == 1 ? (int64_t)arena_bytes<double, double>(*fd)
                                                                           : (int64_t)arena_bytes<DualT<double>, DualT<double>>(*fd, true));
}
int forestsim_loss_f64(const ForestDesc* fd, const SolverOpts* opt, const double* theta, const double* friction, const double* lengths,
                       const double* x, const double* xp, int64_t B, const double* weights, double scale, double* loss, double* grad,
                       double* force, int32_t* iters) {
  return loss_batch<double>(*fd, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
}
int forestsim_loss_f32(const ForestDesc* fd, const SolverOpts* opt, const float* theta, const float* friction, const float* lengths,
                       const float* x, const float* xp, int64_t B, const float* weights, double scale, float* loss, double* grad,
                       float* force, int32_t* iters) {
  return loss_batch<float>(*fd, *opt, theta, friction, lengths, x, xp, B, weights, scale, loss, grad, force, iters);
}
int forestsim_step_f64(const ForestDesc* fd, const SolverOpts* opt, const double* theta, const double* friction, const double* lengths,
                       const double* x, int64_t B, double* x_next, int32_t* iters) {
  return step_batch<double>(*fd, *opt, theta, friction, lengths, x, B, x_next, iters);
}
int forestsim_step_f32(const ForestDesc* fd, const SolverOpts* opt, const float* theta, const float* friction, const float* lengths,
                       const float* x, int64_t B, float* x_next, int32_t* iters) {
  return step_batch<float>(*fd, *opt, theta, friction, lengths, x, B, x_next, iters);
}
int forestsim_terms_f64(const ForestDesc* fd, const double* theta, const double* friction, const double* lengths, const double* x, int64_t B,
                        double* M, double* a, double* phi, double* J) {
  return terms_batch<double>(*fd, theta, friction, lengths, x, B, M, a, phi, J);
}
}
