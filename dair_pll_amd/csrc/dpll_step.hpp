// dpll_step.hpp -- one simulation step (forward_dynamics + Lie-group Euler update), its adjoints with respect to parameters and state, the chain to the learnable parameters
// (part of the per-item math of the contact-dynamics hot path: see dpll_core.hpp for the overview and the reference citations)
#pragma once

#include "dpll_terms.hpp"
#include "dpll_solver.hpp"
#include "dpll_contact.hpp"
#include "dpll_loss.hpp"

namespace dpll {
// ---------------------------------------------------------------------------------------------
// One simulation step: forward_dynamics (multibody_learnable_system.py:199-304) + the Lie-group
// Euler update of VelocityIntegrator.step (integrator.py:153-162, state_space.py:466-486).
// ---------------------------------------------------------------------------------------------
template <typename T> DPLL_HD void quat_exp_mul(const T* q, const T (&r)[3], T* out) {
  // out = q (x) exp(r), quaternion.py:276-309 (exp via sinc), :89-105 (multiply); no re-normalisation
  const T angle = tsqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const T half = angle * T(0.5);
  T s, c;
  tsincos(half, s, c);
  const T sinc = tabs(half) > T(0) ? s / half : T(1);
  const T e[4] = {c, r[0] * sinc * T(0.5), r[1] * sinc * T(0.5), r[2] * sinc * T(0.5)};
  out[0] = q[0] * e[0] - q[1] * e[1] - q[2] * e[2] - q[3] * e[3];
  out[1] = q[0] * e[1] + e[0] * q[1] + (q[2] * e[3] - q[3] * e[2]);
  out[2] = q[0] * e[2] + e[0] * q[2] + (q[3] * e[1] - q[1] * e[3]);
  out[3] = q[0] * e[3] + e[0] * q[3] + (q[1] * e[2] - q[2] * e[1]);
}

template <typename T, typename TA, int NJ, int KPL, class Lanes, int NG, class MD>
DPLL_HD void step_item(const MD& md, const Derived<T, NJ, NG>& dp, const SolverOpts& opt, const T* x,
                       int first_contact, T* x_next, T (&impulse)[KPL][3], int& iters,
                       const T (*witness)[3] = nullptr, const MeshPairIn<T, TA, KPL>* mesh_in = nullptr) {
  constexpr int NV = 6 + NJ, NQ = 7 + NJ;
  const T dt = T(md.dt), eps = T(kDynamicsEps);
  const T* q = x;
  const T* v = x + NQ;
  ItemStore<T, TA, NJ> own_store;
  ItemStore<T, TA, NJ>& store = Lanes::template item_store<ItemStore<T, TA, NJ>>(own_store);
  Terms<T, NJ>& t = store.t;
  Kin<TA, NJ>& kinA = store.kinA;
  compute_terms<T, TA, NJ>(md, dp, q, v, t, kinA);
  T vm[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) vm[i] = v[i] + dt * t.a[i];
  CJac<T, NJ, MD::kGeneral> Jc[KPL];
  T mu[KPL], qc[KPL][3];
  const T idt = T(1) / dt;
  TA pdirs[kMaxPairs][3];
  const bool have_dirs = pair_find_directions<T, TA, Lanes, NJ>(md, dp, kinA, pdirs, mesh_in ? mesh_in->dirs : nullptr);
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    ContactGeom<T, NJ, MD::kGeneral> cg;
    compute_contact<T, TA, NJ>(md, dp, t.kin, kinA, first_contact + c, cg, witness ? witness[c] : nullptr,
                               pair_dir_of<TA>(have_dirs, pdirs, first_contact + c), mesh_in ? mesh_in->wit_a[c] : nullptr);
    Jc[c] = cg.J;
    mu[c] = cg.mu;
    T jv[3];
    cjac_apply<T, T, NJ>(Jc[c], vm, jv);
    qc[c][0] = cg.mu * jv[0];
    qc[c][1] = cg.mu * jv[1];
    qc[c][2] = jv[2] + cg.phi * idt;
  }
  TA y[NV];
  bool winner = true;
  iters = sap_solve<T, TA, NJ, KPL, Lanes>(t.M, Jc, mu, qc, eps, opt, y, impulse, vm, false, true, &winner);
  if constexpr (Lanes::kVariants > 1) {
    // racing copies (rollouts of a batch that leaves SIMDs idle): every copy goes on from the winner's velocity change, so
    // the copies of an item hold the same state at every step (`impulse` stays each copy's own: the rollout does not use it)
    DPLL_UNROLL for (int i = 0; i < NV; ++i) y[i] = Lanes::item_pick(winner, y[i]);
    iters = Lanes::item_pick(winner, iters);
  }
  // v+ = v- + M^-1 J^T impulse = v- + y*: the primal optimum IS that velocity change (M y* = J^T f), and
  // taking it from y instead of re-solving with the projected impulse avoids amplifying the impulse's
  // rounding error by |J|^2 / (eps M).
  T vn[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) vn[i] = T(TA(vm[i]) + y[i]);
  const T r[3] = {vn[0] * dt, vn[1] * dt, vn[2] * dt};
  quat_exp_mul<T>(q, r, x_next);
  DPLL_UNROLL for (int i = 0; i < 3 + NJ; ++i) x_next[4 + i] = q[4 + i] + vn[3 + i] * dt;
  DPLL_UNROLL for (int i = 0; i < NV; ++i) x_next[NQ + i] = vn[i];
}


// ---------------------------------------------------------------------------------------------
// Adjoint of one simulation step with respect to the learnable parameters (state treated as data): what
// back-propagating a one-step prediction loss through dair_pll's forward_dynamics needs (experiment.py:292-320
// with the default t_prediction = 1; there the cone solve's backward is sappy's, unpinned -- here it is the
// implicit-function derivative of the stationarity condition G(y) = M y - sum_c J_c^T P_K(z_c) = 0).
// With s = d/d v+ (including the pull-back of d/d q+ through the Lie-group Euler update), H lambda = s at the
// solution, gamma_c = P_K(z_c), kappa_c = dP_c (J_c lambda) / eps:
//   dL = -B(S lambda, S y; d iota) + sum_c [gamma_c^T dJ_c lambda - kappa_c^T dJ_c v+ - kappa_c,n dphi_c / dt]
//        + (s - sum_c J_c^T kappa_c)^T dt da,          da = M^-1 (dF - dM a)
// ---------------------------------------------------------------------------------------------
template <typename T> DPLL_HD void quat_exp_mul_adjoint(const T* q, const T (&r)[3], const T* obar, T (&rbar)[3]) {
  const T n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const T n = tsqrt(n2);
  T sh, c;
  tsincos(n * T(0.5), sh, c);
  const bool small = !(n > T(1e-6));
  const T sfac = small ? T(0.5) : sh / n;                       // e_v = r * sfac
  const T dsfac = small ? T(-1.0 / 24.0) : (T(0.5) * c - sfac) / n2;  // d sfac / d r = dsfac * r
  const T qv[3] = {q[1], q[2], q[3]}, ov[3] = {obar[1], obar[2], obar[3]};
  const T e0bar = q[0] * obar[0] + dot3(qv, ov);
  T x[3];
  cross(ov, qv, x);
  T evbar[3];
  DPLL_UNROLL for (int i = 0; i < 3; ++i) evbar[i] = -qv[i] * obar[0] + q[0] * ov[i] + x[i];
  const T rdot = dot3(r, evbar);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) rbar[i] = -T(0.5) * sfac * r[i] * e0bar + sfac * evbar[i] + dsfac * rdot * r[i];
}

template <typename T, typename TA, int NJ, int KPL, class Lanes, int NG, class MD>
DPLL_HD void step_state_adjoint(const MD& md, const Derived<T, NJ, NG>& dp, const T* x, int first_contact,
                                const T* xbar_next, const TA (&y)[6 + NJ], const T (&vn)[6 + NJ], const T (&sv)[6 + NJ],
                                const T (&lam)[6 + NJ], T (&xbar)[13 + 2 * NJ], const T (*witness)[3] = nullptr,
                                const T (*pair_dir)[3] = nullptr, const T (*witness_a)[3] = nullptr);

template <typename T, typename TA, int NJ, int KPL, class Lanes, int NG, class MD, int GP>
DPLL_HD void step_item_backward(const MD& md, const Derived<T, NJ, NG>& dp, const SolverOpts& opt, const T* x,
                                int first_contact, const T* xbar_next, LossGrad<T, NJ, NG, GP>& grad,
                                const T (*witness)[3] = nullptr, T (*rbar_out)[3] = nullptr,
                                T (*xbar)[13 + 2 * NJ] = nullptr, const MeshPairIn<T, TA, KPL>* mesh_in = nullptr,
                                T (*rbar_a_out)[3] = nullptr) {
  constexpr int NB = NJ + 1, NV = 6 + NJ, NQ = 7 + NJ;
  const T dt = T(md.dt), eps = T(kDynamicsEps), ieps = fast_rcp(eps);
  const T* q = x;
  const T* v = x + NQ;
  // ---- forward (recomputed, nothing is stored between the passes) --------------------------------
  ItemStore<T, TA, NJ> own_store;
  ItemStore<T, TA, NJ>& store = Lanes::template item_store<ItemStore<T, TA, NJ>>(own_store);
  Terms<T, NJ>& t = store.t;
  Kin<TA, NJ>& kinA = store.kinA;
  compute_terms<T, TA, NJ>(md, dp, q, v, t, kinA);
  T vm[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) vm[i] = v[i] + dt * t.a[i];
  ContactGeom<T, NJ, MD::kGeneral> cg[KPL];
  CJac<T, NJ, MD::kGeneral> Jc[KPL];
  T mu[KPL], qc[KPL][3];
  const T idt = T(1) / dt;
  TA pdirs[kMaxPairs][3];
  const bool have_dirs = pair_find_directions<T, TA, Lanes, NJ>(md, dp, kinA, pdirs, mesh_in ? mesh_in->dirs : nullptr);
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    compute_contact<T, TA, NJ>(md, dp, t.kin, kinA, first_contact + c, cg[c], witness ? witness[c] : nullptr,
                               pair_dir_of<TA>(have_dirs, pdirs, first_contact + c), mesh_in ? mesh_in->wit_a[c] : nullptr);
    Jc[c] = cg[c].J;
    mu[c] = cg[c].mu;
    T jv[3];
    cjac_apply<T, T, NJ>(Jc[c], vm, jv);
    qc[c][0] = mu[c] * jv[0];
    qc[c][1] = mu[c] * jv[1];
    qc[c][2] = jv[2] + cg[c].phi * idt;
  }
  TA y[NV];
  T gam[KPL][3];
  sap_solve<T, TA, NJ, KPL, Lanes>(t.M, Jc, mu, qc, eps, opt, y, gam, vm, false);
  T yT[NV], vn[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) { yT[i] = T(y[i]); vn[i] = T(TA(vm[i]) + y[i]); }
  // ---- seed: d/d v+ plus the pull-back of d/d q+ through q+ = q (+) v+ dt ------------------------
  T sv[NV];
  {
    const T r[3] = {vn[0] * dt, vn[1] * dt, vn[2] * dt};
    T rbar[3];
    quat_exp_mul_adjoint<T>(q, r, xbar_next, rbar);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) sv[i] = xbar_next[NQ + i] + dt * rbar[i];
    DPLL_UNROLL for (int i = 3; i < NV; ++i) sv[i] = xbar_next[NQ + i] + dt * xbar_next[4 + (i - 3)];
  }
  // ---- H lambda = s at the solution ---------------------------------------------------------------
  Proj<T> pr[KPL];
  T H[NV][NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i)
    DPLL_UNROLL for (int j = 0; j <= i; ++j) H[i][j] = T(0);
  T dPc[KPL][6];
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    TA jy[3];
    cjac_apply<T, TA, NJ>(Jc[c], y, jy);
    const T z[3] = {-T(TA(mu[c]) * jy[0] + TA(qc[c][0])) * ieps, -T(TA(mu[c]) * jy[1] + TA(qc[c][1])) * ieps,
                    -T(jy[2] + TA(qc[c][2])) * ieps};
    lorentz_project(z, pr[c]);
    proj_jacobian(pr[c], dPc[c]);
    const T(&dP)[6] = dPc[c];
    const T m1 = mu[c] * ieps, m2 = mu[c] * m1;
    const T C[3][3] = {{dP[0] * m2, dP[3] * m2, dP[4] * m1}, {dP[3] * m2, dP[1] * m2, dP[5] * m1}, {dP[4] * m1, dP[5] * m1, dP[2] * ieps}};
    hessian_add<T, NJ>(Jc[c], C, H);
  }
  DPLL_UNROLL for (int i = 0; i < NV; ++i)
    DPLL_UNROLL for (int j = 0; j <= i; ++j) {
      H[i][j] = t.M[i][j] + Lanes::group_sum(H[i][j]);
      H[j][i] = H[i][j];
    }
  T L[NV][NV], invd[NV], lam[NV];
  cholesky<T, NV>(H, L, invd);
  chol_solve<T, NV>(L, invd, sv, lam);
  if (xbar) {
    // the body-body directions found above are constants of the dual passes (piecewise constant in the state)
    T pdir[kMaxPairs][3] = {};
    if constexpr (MD::kGeneral) {
      DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
        const int pp = first_contact + c - kQuery * kMaxGeoms;
        if (pp >= 0 && pp < kMaxPairs) {
          DPLL_UNROLL for (int i = 0; i < 3; ++i) pdir[pp][i] = cg[c].dir[i];
        }
      }
    }
    step_state_adjoint<T, TA, NJ, KPL, Lanes>(md, dp, x, first_contact, xbar_next, y, vn, sv, lam, *xbar, witness, pdir,
                                              mesh_in ? mesh_in->wit_a : nullptr);
  }
  // ---- per-contact pieces: kappa_c, friction and witness gradients; s' = s - sum_c J_c^T kappa_c ---
  T jtk[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) jtk[i] = T(0);
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    T pl[3], pv[3];
    cjac_apply<T, T, NJ>(Jc[c], lam, pl);  // Jp lambda
    cjac_apply<T, T, NJ>(Jc[c], vn, pv);   // Jp v+
    const T jl[3] = {mu[c] * pl[0], mu[c] * pl[1], pl[2]};
    const T(&dP)[6] = dPc[c];
    const T kap[3] = {ieps * (dP[0] * jl[0] + dP[3] * jl[1] + dP[4] * jl[2]),
                      ieps * (dP[3] * jl[0] + dP[1] * jl[1] + dP[5] * jl[2]),
                      ieps * (dP[4] * jl[0] + dP[5] * jl[1] + dP[2] * jl[2])};
    const T ak[3] = {mu[c] * kap[0], mu[c] * kap[1], kap[2]};
    cjac_apply_t_add<T, NJ>(Jc[c], ak, jtk);
    const T(&g)[3] = pr[c].g;
    const T gmu = g[0] * pl[0] + g[1] * pl[1] - kap[0] * pv[0] - kap[1] * pv[1];
    const T ag[3] = {mu[c] * g[0], mu[c] * g[1], g[2]};
    const T nak[3] = {-ak[0], -ak[1], -ak[2]};
    T rbar[3], rbar_a[3];
    witness_adjoint<T, NJ>(t.kin, cg[c], ag, lam, nak, vn, -kap[2] * idt, rbar, rbar_a);
    if (rbar_out) {
      DPLL_UNROLL for (int i = 0; i < 3; ++i) rbar_out[c][i] = rbar[i];
    }
    if (rbar_a_out) {
      DPLL_UNROLL for (int i = 0; i < 3; ++i) rbar_a_out[c][i] = rbar_a[i];
    }
    add_geometry_grad(cg[c], gmu, rbar, rbar_a, grad);
  }
  T abar[NV], bvec[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) abar[i] = dt * (sv[i] - Lanes::group_sum(jtk[i]));
  chol_solve<T, NV>(t.LM, t.invdM, abar, bvec);
  // ---- inertial part --------------------------------------------------------------------------------
  T Lw[NB][3], Lu[NB][3], Yw[NB][3], Yu[NB][3], Bw[NB][3], Bu[NB][3], Aw[NB][3], Au[NB][3];
  body_twists<T, NJ>(t.kin, lam, Lw, Lu);
  body_twists<T, NJ>(t.kin, yT, Yw, Yu);
  body_twists<T, NJ>(t.kin, bvec, Bw, Bu);
  body_twists<T, NJ>(t.kin, t.a, Aw, Au);
  DPLL_UNROLL for (int b = 0; b < NB; ++b) {
    inertia_bilinear_grad<T>(T(-1), Lw[b], Lu[b], Yw[b], Yu[b], grad.g_iota[b]);
    T accw[3], accu[3];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { accw[i] = Aw[b][i] + t.AGw[b][i]; accu[i] = Au[b][i] + t.AGu[b][i]; }
    inertia_bilinear_grad<T>(T(-1), Bw[b], Bu[b], accw, accu, grad.g_iota[b]);
    T cw[3], c1[3], c2[3], cu[3];
    cross(t.Vw[b], Bw[b], cw);
    cross(t.Vw[b], Bu[b], c1);
    cross(t.Vu[b], Bw[b], c2);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) cu[i] = c1[i] + c2[i];
    inertia_bilinear_grad<T>(T(1), cw, cu, t.Vw[b], t.Vu[b], grad.g_iota[b]);
  }
}

// ---------------------------------------------------------------------------------------------
// Adjoint of one simulation step with respect to the STATE (back-propagation through time over several steps).
// With the optimum y*, the seed s = d/d v+ and H lambda = s of step_item_backward held fixed, d(total)/dx is the
// partial derivative of the scalar
//     Phi(x) = xbar+_q . q+(q, v+ fixed) + s . v-(x) - lambda . G(x, y*),     G = M(q) y* - sum_c J_c(q)^T D_mu P_K(z_c(x, y*)),
// (implicit-function theorem on G = 0: dy*/dx = -H^-1 dG/dx).  Phi is pushed through the same templates as the
// forward pass with forward-mode duals, one state component per pass: n_x passes of (terms + contact geometry) in
// the accumulation type.  The witness is piecewise constant in q in both geometries: a box corner, or the support
// point of a LeakyReLU network (piecewise linear support function => piecewise constant gradient), passed in.
// ---------------------------------------------------------------------------------------------
template <typename T, typename TA, int NJ, int KPL, class Lanes, int NG, class MD>
DPLL_HD void step_state_adjoint(const MD& md, const Derived<T, NJ, NG>& dp, const T* x, int first_contact,
                                const T* xbar_next, const TA (&y)[6 + NJ], const T (&vn)[6 + NJ], const T (&sv)[6 + NJ],
                                const T (&lam)[6 + NJ], T (&xbar)[13 + 2 * NJ], const T (*witness)[3],
                                const T (*pair_dir)[3], const T (*witness_a)[3]) {
  constexpr int NB = NJ + 1, NV = 6 + NJ, NQ = 7 + NJ, NX = NQ + NV;
  using S = DualT<TA>;
  Derived<S, NJ, NG> dps;
  DPLL_UNROLL for (int b = 0; b < NB; ++b)
    DPLL_UNROLL for (int i = 0; i < kIota; ++i) dps.iota[b][i] = S(TA(dp.iota[b][i]));
  DPLL_UNROLL for (int g = 0; g < NG; ++g) {
    dps.mu[g] = S(TA(dp.mu[g]));
    DPLL_UNROLL for (int i = 0; i < 3; ++i) dps.habs[g][i] = S(TA(dp.habs[g][i]));
  }
  DPLL_UNROLL for (int p = 0; p < kMaxPairs; ++p) dps.mu_pair[p] = S(TA(dp.mu_pair[p]));
  DPLL_UNROLL for (int j = 0; j < (NJ > 0 ? NJ : 1); ++j) dps.tau[j] = S(TA(dp.tau[j]));
  dps.geo = nullptr;
  S geo_s[MD::kGeneral ? NG * MD::kGeoStride : 1];  // a polygon's vertices as constants of the dual passes
  if constexpr (MD::kGeneral) {
    if (dp.geo) {
      DPLL_UNROLL for (int i = 0; i < NG * MD::kGeoStride; ++i) geo_s[i] = S(TA(dp.geo[i]));
      dps.geo = geo_s;
    }
  }
  const S dt = S(TA(md.dt)), idt = S(TA(1) / TA(md.dt)), mieps = S(TA(-1) / TA(kDynamicsEps));
  for (int k = 0; k < NX; ++k) {  // deliberately not unrolled: one copy of the dual forward pass
    S xs[NX];
    DPLL_UNROLL for (int i = 0; i < NX; ++i) xs[i] = S(TA(x[i]), i == k ? TA(1) : TA(0));
    Terms<S, NJ> t;
    Kin<S, NJ> kin;
    compute_terms<S, S, NJ>(md, dps, xs, xs + NQ, t, kin);
    S vm[NV], ys[NV], ls[NV];
    DPLL_UNROLL for (int i = 0; i < NV; ++i) { vm[i] = xs[NQ + i] + dt * t.a[i]; ys[i] = S(y[i]); ls[i] = S(TA(lam[i])); }
    // q+ with the rotation vector v+ dt held fixed (linear in q), s . v-, -lambda . M y*
    const S r[3] = {S(TA(vn[0]) * TA(md.dt)), S(TA(vn[1]) * TA(md.dt)), S(TA(vn[2]) * TA(md.dt))};
    S qn[4];
    quat_exp_mul<S>(xs, r, qn);
    S phi = S(TA(0));
    DPLL_UNROLL for (int i = 0; i < 4; ++i) phi += S(TA(xbar_next[i])) * qn[i];
    DPLL_UNROLL for (int i = 0; i < 3 + NJ; ++i) phi += S(TA(xbar_next[4 + i])) * xs[4 + i];
    DPLL_UNROLL for (int i = 0; i < NV; ++i) phi += S(TA(sv[i])) * vm[i];
    DPLL_UNROLL for (int i = 0; i < NV; ++i) {
      S my = S(TA(0));
      DPLL_UNROLL for (int j = 0; j < NV; ++j) my += t.M[i][j] * ys[j];
      phi -= ls[i] * my;
    }
    // + sum_c (J_c lambda) . D_mu P_K(z_c)
    S phic = S(TA(0));
    DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
      ContactGeom<S, NJ, MD::kGeneral> cg;
      S wit[3] = {S(TA(0)), S(TA(0)), S(TA(0))}, wit_a[3] = {S(TA(0)), S(TA(0)), S(TA(0))};
      if (witness) {
        DPLL_UNROLL for (int i = 0; i < 3; ++i) wit[i] = S(TA(witness[c][i]));
      }
      if (witness_a) {
        DPLL_UNROLL for (int i = 0; i < 3; ++i) wit_a[i] = S(TA(witness_a[c][i]));
      }
      S pd[3] = {S(TA(0)), S(TA(0)), S(TA(1))};
      const int pp = first_contact + c - kQuery * kMaxGeoms;
      const bool is_pair = MD::kGeneral && pair_dir && pp >= 0 && pp < kMaxPairs;
      if (is_pair) {
        DPLL_UNROLL for (int i = 0; i < 3; ++i) pd[i] = S(TA(pair_dir[pp][i]));
      }
      compute_contact<S, S, NJ>(md, dps, t.kin, kin, first_contact + c, cg, witness ? wit : nullptr, is_pair ? pd : nullptr,
                                witness_a ? wit_a : nullptr);
      S jy[3], jv[3], jl[3];
      cjac_apply<S, S, NJ>(cg.J, ys, jy);
      cjac_apply<S, S, NJ>(cg.J, vm, jv);
      cjac_apply<S, S, NJ>(cg.J, ls, jl);
      const S z[3] = {(cg.mu * jy[0] + cg.mu * jv[0]) * mieps, (cg.mu * jy[1] + cg.mu * jv[1]) * mieps,
                      (jy[2] + jv[2] + cg.phi * idt) * mieps};
      // projection of a dual: value through lorentz_project, derivative through its generalised Jacobian
      const TA zv[3] = {z[0].v, z[1].v, z[2].v};
      Proj<TA> pr;
      lorentz_project(zv, pr);
      TA dP[6];
      proj_jacobian(pr, dP);
      const S f[3] = {S(pr.g[0], dP[0] * z[0].d + dP[3] * z[1].d + dP[4] * z[2].d),
                      S(pr.g[1], dP[3] * z[0].d + dP[1] * z[1].d + dP[5] * z[2].d),
                      S(pr.g[2], dP[4] * z[0].d + dP[5] * z[1].d + dP[2] * z[2].d)};
      phic += cg.mu * (f[0] * jl[0] + f[1] * jl[1]) + f[2] * jl[2];
    }
    const TA total = phi.d + Lanes::group_sum(phic.d);
    DPLL_UNROLL for (int i = 0; i < NX; ++i) xbar[i] = (i == k) ? T(total) : xbar[i];
  }
}

// ---------------------------------------------------------------------------------------------
// chain from the batch-reduced gradients (d/d iota, d/d mu_pair, d/d |length|) to the learnable
// parameters (theta, friction_params, length_params); double precision, a handful of flops.
// ---------------------------------------------------------------------------------------------
using Dual = DualT<double>;

// d(sum_i g_iota[i] iota_i(theta)) / d theta_k for one body
DPLL_HD double theta_grad_component(int inertia_mode, const double* theta, const double* g_iota, int k,
                                    const double (*body_rot)[3][3] = nullptr) {
  Dual th[10], io[kIota];
  DPLL_UNROLL for (int i = 0; i < 10; ++i) th[i] = Dual(theta[i], i == k ? 1.0 : 0.0);
  theta_to_iota<Dual>(th, inertia_mode, io);
  if (body_rot) rotate_iota<Dual>(*body_rot, io);
  double s = 0.0;
  DPLL_UNROLL for (int i = 0; i < kIota; ++i) s += g_iota[i] * io[i].d;
  return s;
}
// friction_params gradient: entry 0 is the ground, entry 1 + b geometry b; slot b of g_mu combines the ground with
// geometry b; (general build) g_mu_pair[p] belongs to the coefficient of the two geometries of body-body candidate p
DPLL_HD double friction_grad_component(int n_slots, const double* friction, const double* g_mu, int k,
                                       const ModelDesc* gd = nullptr, const double* g_mu_pair = nullptr) {
  double s = 0.0;
  auto add = [&](int ia, int ib, double g) {
    const double m0 = fabs(friction[ia]), mb = fabs(friction[ib]);
    const double den = (m0 + mb) * (m0 + mb);
    if (k == ia) s += g * 2.0 * mb * mb / den;
    if (k == ib) s += g * 2.0 * m0 * m0 / den;
  };
  for (int b = 0; b < n_slots; ++b)
    if (!gd || b < kMaxGeoms) add(0, 1 + b, g_mu[b]);
  if (gd && g_mu_pair)
    for (int p = 0; p < gd->n_pairs && p < kMaxPairs; ++p) add(1 + gd->pair_a[p], 1 + gd->pair_b[p], g_mu_pair[p]);
  const double p = friction[k];
  return s * (p > 0.0 ? 1.0 : (p < 0.0 ? -1.0 : 0.0));
}
DPLL_HD double length_grad_component(const double* lengths, const double* g_len, int k) {
  const double p = lengths[k];
  return g_len[k] * (p > 0.0 ? 1.0 : (p < 0.0 ? -1.0 : 0.0));
}


}  // namespace dpll
