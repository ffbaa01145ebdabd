// dpll_loss.hpp -- ContactNets loss of one item, forward + hand-derived adjoint (multibody_learnable_system.py:104-197)
// (part of the per-item math of the contact-dynamics hot path: see dpll_core.hpp for the overview and the reference citations)
#pragma once

#include "dpll_terms.hpp"
#include "dpll_solver.hpp"
#include "dpll_contact.hpp"

namespace dpll {
// ---------------------------------------------------------------------------------------------
// ContactNets loss for one item, forward + adjoint (multibody_learnable_system.py:104-197).
// Gradient accumulators (all scaled by `weight`, the upstream d(total)/d(loss_item)):
//   g_iota[b][10]  d/d iota_b          -- identical in every lane of the group
//   g_mu[b]        d/d (pair friction) -- this lane's contacts only (sum over lanes = item total)
//   g_len[b][3]    d/d |length_params| -- this lane's contacts only
// ---------------------------------------------------------------------------------------------
// GP = numbers per geometry in g_len: 3 (a box's |length_params|; a sphere uses [0]) or, in the general build, the
// geometry block stride 3 kMaxPolyVerts (a polygon's vertices)
template <typename T, int NJ, int NG = NJ + 1, int GP = 3> struct LossGrad {
  static constexpr int NB = NJ + 1;
  T g_iota[NB][kIota];
  T g_mu[NG];
  T g_len[NG][GP];
};

template <typename T, int NJ, int NG, int GP> DPLL_HD void zero_grad(LossGrad<T, NJ, NG, GP>& g) {
  DPLL_UNROLL for (int b = 0; b <= NJ; ++b)
    DPLL_UNROLL for (int i = 0; i < kIota; ++i) g.g_iota[b][i] = T(0);
  DPLL_UNROLL for (int gg = 0; gg < NG; ++gg) {
    g.g_mu[gg] = T(0);
    DPLL_UNROLL for (int i = 0; i < GP; ++i) g.g_len[gg][i] = T(0);
  }
}

// this contact's share of d/d(mu_pair, geometry lengths): gmu and the witness adjoint r_bar
// one witness point's adjoint into the parameters of geometry `gpar`
template <typename T, int NJ, int NG, int GP>
DPLL_HD void add_witness_grad(int gpar, const T (&sgn)[3], const T (&drad)[3], int vidx, const T (&rbar)[3],
                              LossGrad<T, NJ, NG, GP>& grad) {
  const T grad_r = drad[0] * rbar[0] + drad[1] * rbar[1] + drad[2] * rbar[2];
  DPLL_UNROLL for (int gg = 0; gg < NG; ++gg) {
    const bool mine = (gpar == gg);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) grad.g_len[gg][i] += mine ? sgn[i] * rbar[i] + (i == 0 ? grad_r : T(0)) : T(0);
    if constexpr (GP >= 3 * kMaxPolyVerts) {  // (a polygon has sgn = drad = 0: the line above adds nothing for it)
      DPLL_UNROLL for (int u = 0; u < kMaxPolyVerts; ++u)
        DPLL_UNROLL for (int i = 0; i < 3; ++i) grad.g_len[gg][3 * u + i] += (mine && vidx == u) ? rbar[i] : T(0);
    }
  }
}
template <typename T, int NJ, bool GEN, int NG, int GP>
DPLL_HD void add_geometry_grad(const ContactGeom<T, NJ, GEN>& cg, T gmu, const T (&rbar)[3], const T (&rbar_a)[3],
                               LossGrad<T, NJ, NG, GP>& grad) {
  if constexpr (GEN && NG > kMaxGeoms && GP >= kMaxPairs) {
    // a body-body contact's coefficient has a column of its own: entry p of the (parameterless) block behind the geometries
    DPLL_UNROLL for (int gg = 0; gg < NG; ++gg) grad.g_mu[gg] += (!cg.pair && cg.geom == gg) ? gmu : T(0);
    DPLL_UNROLL for (int p = 0; p < kMaxPairs; ++p) grad.g_len[NG > kMaxGeoms ? kMaxGeoms : 0][p] += (cg.pair && cg.pidx == p) ? gmu : T(0);
  } else {
    DPLL_UNROLL for (int gg = 0; gg < NG; ++gg) grad.g_mu[gg] += (cg.geom == gg) ? gmu : T(0);
  }
  if constexpr (GEN) {
    add_witness_grad<T, NJ>(cg.gpar, cg.sgn, cg.drad, cg.vidx, rbar, grad);
    add_witness_grad<T, NJ>(cg.pair ? cg.gpar_a : -1, cg.sgn_a, cg.drad_a, cg.vidx_a, rbar_a, grad);
  } else {
    add_witness_grad<T, NJ>(cg.geom, cg.sgn, cg.drad, cg.vidx, rbar, grad);
  }
}

// General build with learned shapes: what a lane needs beyond `witness` (the support point of its contact's geometry; for
// a body-body contact the one of B along -d)
template <typename T, typename TA, int KPL> struct MeshPairIn {
  T wit_a[KPL][3];        // body-body contact of two learned shapes: A's support point along d
  TA dirs[kMaxPairs][3];  // the candidates' directions in the frame of A, found by the GJK / EPA kernel (csrc/dpll_gjk.hpp)
};

constexpr double kLossEps = 1e-3;       // multibody_learnable_system.py:130
constexpr double kDynamicsEps = 1e-4;   // multibody_learnable_system.py:283, 298
constexpr double kInvalidForce = 1e3;   // multibody_learnable_system.py:187

template <typename T, typename TA, int NJ, int KPL, class Lanes, int NG, class MD, int GP>
DPLL_HD T loss_item(const MD& md, const Derived<T, NJ, NG>& dp, const SolverOpts& opt, const T* x, const T* xp,
                    int first_contact, T weight, bool want_grad, LossGrad<T, NJ, NG, GP>& grad, T (&force)[KPL][3],
                    int& iters, const T (*witness)[3] = nullptr, T (*rbar_out)[3] = nullptr,
                    const MeshPairIn<T, TA, KPL>* mesh_in = nullptr, T (*rbar_a_out)[3] = nullptr, bool* winner_out = nullptr) {
  constexpr int NB = NJ + 1, NV = 6 + NJ, NQ = 7 + NJ;
  const T dt = T(md.dt), eps = T(kLossEps);
  const T* v = x + NQ;
  const T* qp = xp;
  const T* vp = xp + NQ;
  ItemStore<T, TA, NJ> own_store;
  ItemStore<T, TA, NJ>& store = Lanes::template item_store<ItemStore<T, TA, NJ>>(own_store);
  Terms<T, NJ>& t = store.t;
  Kin<TA, NJ>& kinA = store.kinA;
  compute_terms<T, TA, NJ>(md, dp, qp, vp, t, kinA);  // terms at the NEXT state (quirk Q6)
  T dv[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) dv[i] = vp[i] - (v[i] + t.a[i] * dt);
  // contacts of this lane
  ContactGeom<T, NJ, MD::kGeneral> cg[KPL];
  CJac<T, NJ, MD::kGeneral> Jc[KPL];
  T mu[KPL], qc[KPL][3], slide[KPL][2], speed[KPL], jpv[KPL][3];
  T pen = T(0);
  TA pdirs[kMaxPairs][3];
  const bool have_dirs = pair_find_directions<T, TA, Lanes, NJ>(md, dp, kinA, pdirs, mesh_in ? mesh_in->dirs : nullptr);
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    compute_contact<T, TA, NJ>(md, dp, t.kin, kinA, first_contact + c, cg[c], witness ? witness[c] : nullptr,
                               pair_dir_of<TA>(have_dirs, pdirs, first_contact + c), mesh_in ? mesh_in->wit_a[c] : nullptr);
    Jc[c] = cg[c].J;
    mu[c] = cg[c].mu;
    T jdv[3];
    cjac_apply<T, T, NJ>(Jc[c], dv, jdv);
    cjac_apply<T, T, NJ>(Jc[c], vp, jpv[c]);
    slide[c][0] = mu[c] * jpv[c][0];
    slide[c][1] = mu[c] * jpv[c][1];
    speed[c] = tsqrt(slide[c][0] * slide[c][0] + slide[c][1] * slide[c][1]);
    qc[c][0] = -mu[c] * jdv[0] + dt * slide[c][0];
    qc[c][1] = -mu[c] * jdv[1] + dt * slide[c][1];
    qc[c][2] = -jdv[2] + tabs(cg[c].phi) + dt * speed[c];
    const T neg = tmax(-cg[c].phi, T(0));
    pen += neg * neg;
  }
  pen = Lanes::group_sum(pen);
  TA y[NV];
  DPLL_CORE_STAMP(4);
  SolverOpts loss_opt = opt;
  if (opt.loss_n_stages > 0) {
    loss_opt.n_stages = opt.loss_n_stages;
    loss_opt.stage_factor = opt.loss_stage_factor;
  }
  bool winner = true;  // racing copies of the item (SolverOpts::portfolio): the copy whose result counts
  iters = sap_solve<T, TA, NJ, KPL, Lanes>(t.M, Jc, mu, qc, eps, loss_opt, y, force, dv, opt.warm_start != 0, true, &winner);
  if (winner_out) *winner_out = winner;
  DPLL_CORE_STAMP(5);
  // invalid-solve mask (multibody_learnable_system.py:186-192)
  bool bad = false;
  DPLL_UNROLL for (int c = 0; c < KPL; ++c)
    DPLL_UNROLL for (int r = 0; r < 3; ++r) bad = bad || bad_number(force[c][r]) || tabs(force[c][r]) > T(kInvalidForce);
  bad = Lanes::group_any(bad);
  // (a copy that lost the race contributes nothing: its gradient terms are masked like an invalid solve's; the kernel
  // takes loss, forces and iteration count from the winner's lanes)
  if constexpr (Lanes::kVariants > 1) bad = bad || !winner;
  DPLL_UNROLL for (int c = 0; c < KPL; ++c)
    DPLL_UNROLL for (int r = 0; r < 3; ++r) force[c][r] = bad ? T(0) : force[c][r];
  // g = J^T f, w = M^-1 g
  T g[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) g[i] = T(0);
  // The loss  1/2 g^T M^-1 g + f . qc + 1/2 dv^T M dv  with  qc = -D_mu J dv + r,  r = (dt slide_x, dt slide_y, |phi| + dt speed)
  // per contact, is evaluated as  1/2 u . (g - M dv) + f . r  with  u = M^-1 g - dv  (M u = g - M dv): written the first way
  // it subtracts numbers of the size of dv^T M dv to get a loss that is often a hundredth of it -- float32 lost 1e-6 .. 8e-6 of
  // the loss on the general models that way (round 4: 1e-8 .. 1e-7) -- and the vectors of the second form are the adjoint's own.
  T fr = T(0), ff = T(0);
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    const T a[3] = {mu[c] * force[c][0], mu[c] * force[c][1], force[c][2]};
    cjac_apply_t_add<T, NJ>(Jc[c], a, g);
    fr += dt * (force[c][0] * slide[c][0] + force[c][1] * slide[c][1]) + force[c][2] * (tabs(cg[c].phi) + dt * speed[c]);
    ff += force[c][0] * force[c][0] + force[c][1] * force[c][1] + force[c][2] * force[c][2];
  }
  DPLL_UNROLL for (int i = 0; i < NV; ++i) g[i] = Lanes::group_sum(g[i]);
  fr = Lanes::group_sum(fr);
  ff = Lanes::group_sum(ff);
  T w[NV], Mdv[NV];
  chol_solve<T, NV>(t.LM, t.invdM, g, w);
  symv<T, NV>(t.M, dv, Mdv);
  T quad = T(0);
  DPLL_UNROLL for (int i = 0; i < NV; ++i) quad += (w[i] - dv[i]) * (g[i] - Mdv[i]);
  const T loss = bad ? T(0) : T(0.5) * (quad + eps * ff) + fr + pen;  // (a masked solve: forces zero and loss zero, as the reference)
  if (!want_grad) return loss;

  // ---- adjoint ------------------------------------------------------------------------------
  // The value above uses w = M^-1 J^T f (variationally consistent: the loss is stationary in f, so its
  // error is second order in the solver error).  The adjoint instead takes w = y*, the primal optimum,
  // which equals M^-1 J^T f at convergence but carries far less rounding error than re-solving with the
  // projected force (that route amplifies it by |J|^2 / (eps M)).
  DPLL_UNROLL for (int i = 0; i < NV; ++i) w[i] = bad ? T(0) : T(y[i]);
  const T wt = bad ? T(0) : weight;
  T u[NV], abar[NV], bvec[NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i) { u[i] = w[i] - dv[i]; abar[i] = -dt * (Mdv[i] - g[i]); }
  chol_solve<T, NV>(t.LM, t.invdM, abar, bvec);
  // inertial part: sum of bilinear forms in the body twists
  T Ww[NB][3], Wu[NB][3], Dw[NB][3], Du[NB][3], Bw[NB][3], Bu[NB][3], Aw[NB][3], Au[NB][3];
  body_twists<T, NJ>(t.kin, w, Ww, Wu);
  body_twists<T, NJ>(t.kin, dv, Dw, Du);
  body_twists<T, NJ>(t.kin, bvec, Bw, Bu);
  body_twists<T, NJ>(t.kin, t.a, Aw, Au);
  DPLL_UNROLL for (int b = 0; b < NB; ++b) {
    inertia_bilinear_grad<T>(T(-0.5) * wt, Ww[b], Wu[b], Ww[b], Wu[b], grad.g_iota[b]);
    inertia_bilinear_grad<T>(T(0.5) * wt, Dw[b], Du[b], Dw[b], Du[b], grad.g_iota[b]);
    T accw[3], accu[3];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { accw[i] = Aw[b][i] + t.AGw[b][i]; accu[i] = Au[b][i] + t.AGu[b][i]; }
    inertia_bilinear_grad<T>(-wt, Bw[b], Bu[b], accw, accu, grad.g_iota[b]);
    // (V x_m B): (Vw x Bw, Vw x Bu + Vu x Bw)
    T cw[3], c1[3], c2[3], cu[3];
    cross(t.Vw[b], Bw[b], cw);
    cross(t.Vw[b], Bu[b], c1);
    cross(t.Vu[b], Bw[b], c2);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) cu[i] = c1[i] + c2[i];
    inertia_bilinear_grad<T>(wt, cw, cu, t.Vw[b], t.Vu[b], grad.g_iota[b]);
  }
  // contact part
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    const T ftx = force[c][0], fty = force[c][1], fn = force[c][2];
    const T isp = speed[c] > T(0) ? T(1) / speed[c] : T(0);
    const T shx = slide[c][0] * isp, shy = slide[c][1] * isp;
    T jpu[3];
    cjac_apply<T, T, NJ>(Jc[c], u, jpu);
    const T bx = dt * (fn * shx + ftx), by = dt * (fn * shy + fty);
    const T gmu = ftx * jpu[0] + fty * jpu[1] + bx * jpv[c][0] + by * jpv[c][1];
    const T phibar = fn * (cg[c].phi > T(0) ? T(1) : (cg[c].phi < T(0) ? T(-1) : T(0))) - T(2) * tmax(-cg[c].phi, T(0));
    const T alpha[3] = {mu[c] * ftx, mu[c] * fty, fn};
    const T beta[3] = {mu[c] * bx, mu[c] * by, T(0)};
    T rbar[3], rbar_a[3];
    witness_adjoint<T, NJ>(t.kin, cg[c], alpha, u, beta, vp, phibar, rbar, rbar_a);  // r_bar = R_b^T rho_bar
    if (rbar_out) {
      DPLL_UNROLL for (int i = 0; i < 3; ++i) rbar_out[c][i] = wt * rbar[i];
    }
    if (rbar_a_out) {
      DPLL_UNROLL for (int i = 0; i < 3; ++i) rbar_a_out[c][i] = wt * rbar_a[i];
    }
    const T wrbar[3] = {wt * rbar[0], wt * rbar[1], wt * rbar[2]};
    const T wrbar_a[3] = {wt * rbar_a[0], wt * rbar_a[1], wt * rbar_a[2]};
    add_geometry_grad(cg[c], wt * gmu, wrbar, wrbar_a, grad);
  }
  return loss;
}

}  // namespace dpll
