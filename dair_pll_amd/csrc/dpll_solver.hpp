// dpll_solver.hpp -- the cone QP of the contact solve: semi-smooth Newton on the unconstrained primal (sap_newton), its line search, continuation and racing copies
// (part of the per-item math of the contact-dynamics hot path: see dpll_core.hpp for the overview and the reference citations)
#pragma once

#include "dpll_terms.hpp"

namespace dpll {
// ---------------------------------------------------------------------------------------------
// The cone QP:  argmin_{f in K^k} 1/2 f^T (J M^-1 J^T + eps 1) f + q^T f
// (what dair_pll asks of sappy.SAPSolver.apply(J_M, P^T q, eps), multibody_learnable_system.py:181-184,
// 295-298, with J_M J_M^T = J M^-1 J^T).  Solved on the equivalent unconstrained primal in the
// generalized velocity y:
//      l(y) = 1/2 y^T M y + eps/2 sum_c |P_K(-(J_c y + q_c)/eps)|^2,    f_c = P_K(-(J_c y* + q_c)/eps),
// whose minimiser satisfies M y* = J^T f, i.e. y* = M^-1 J^T f is exactly the velocity change the
// loss and the dynamics need.  Semi-smooth Newton with a derivative-based, safeguarded line search;
// Newton is affine invariant so the iterates coincide with those of the whitened problem the reference
// hands to its solver.  Contact rows are J_c = D_mu [A | 1 | j] (D_mu = diag(mu, mu, 1)): every product
// with J exploits the identity block.  TA is the accumulation type of the cone residual J y + q (its O(1)
// terms cancel to O(eps |f|), so float kernels carry y and that residual in double).
// ---------------------------------------------------------------------------------------------
struct SolverOpts {
  int max_iter;
  int max_ls;
  double tol;        // on the Newton decrement relative to 1 + |y|_M
  double stall_tol;  // a decrement that stopped halving ends the solve only below this (rounding floor)
  double ls_tol;     // on |l'(alpha)| relative to |l'(0)|
  // continuation in the regularisation: the solve starts at eps * stage_factor^(n_stages - 1) and divides eps by
  // stage_factor whenever a stage has converged to stage_tol (or used stage_max_iter iterations), warm starting
  // the next stage; only the last stage (the reference's eps) runs to `tol`.  Softer cones have fewer kinks, so the
  // active set is found along a smooth path: worst-case Newton iterations drop by ~30 %.
  int n_stages;
  int stage_max_iter;
  double stage_factor;
  double stage_tol;
  double stage_ls_tol;  // line-search tolerance of the non-final stages (their iterates are only warm starts)
  int stage_max_ls;     // line-search probes per iteration in the non-final stages
  // probes per iteration while the decrement keeps falling (0: always max_ls / stage_max_ls).  Exact line searches
  // rarely change the Newton path but a wave pays for its slowest item's probes on every iteration, so an iteration
  // only gets the full search once the decrement has failed to drop by 4x twice in a row (the safeguard that keeps
  // the method globally convergent).
  int fast_ls;
  // loss solve starts from y0 = dv, the observed velocity jump beyond free flight (on data the model explains,
  // y* = M^-1 J^T f is close to it), instead of from zero; the dynamics solve has no such observation and ignores it
  int warm_start;
  // build of the loss kernel: -1 picked from the batch size, 0 always one lane per contact, 1 always one lane per item
  int wide;
  // continuation schedule of the LOSS solve when it should differ from the dynamics solve's (n_stages / stage_factor above):
  // the loss regularises with eps = 1e-3, the dynamics with 1e-4, and the worst case of one is not the worst case of the
  // other (elbow, 4096 pairs: loss 21 -> 18 iterations with 5 stages of 2.5, where the dynamics solve goes 19 -> 36).
  // loss_n_stages = 0: same schedule for both.
  double loss_stage_factor;
  int loss_n_stages;
  // double-precision solves: 1 = float iterations refined in double (sap_solve), 0 = every iteration in double
  int f64_refine;
  // float mesh pipeline: form of the ICNN GEMM kernels -- 0 = v_mfma_f32_32x32x2_f32 (exact f32; default), 2 / 3 = the bf16
  // matrix cores on operands split into 2 / 3 bf16 planes (csrc/dpll_mesh_bf16.hpp)
  int mesh_gemm;
  // racing schedules (dpll.h): copies of an item's lane group with other continuation schedules; 0 = by batch size, 1 = off
  int portfolio;
  int race_stages[3];
  int race_flags[3];  // 1 = warm start, 2 = full Newton steps only (no line search)
  double race_factor[3];
};

// in-place-free Cholesky that only keeps what the solves need: strictly-lower L and 1 / diag
template <typename T, int N> DPLL_HD void cholesky_fast(const T (&A)[N][N], T (&L)[N][N], T (&invd)[N]) {
  DPLL_UNROLL for (int j = 0; j < N; ++j) {
    T s = A[j][j];
    DPLL_UNROLL for (int p = 0; p < j; ++p) s -= L[j][p] * L[j][p];
    const T id = fast_rsqrt(s);
    invd[j] = id;
    DPLL_UNROLL for (int i = j + 1; i < N; ++i) {
      T t = A[i][j];
      DPLL_UNROLL for (int p = 0; p < j; ++p) t -= L[i][p] * L[j][p];
      L[i][j] = t * id;
    }
  }
}

// State of the solve at one iterate y, kept free of the regularisation: zs = -(J y + q) = eps z and gs = P_K(zs) =
// eps P_K(z) (the cone is scale invariant, and so are the coefficients cp, a, b, that of the projection's Jacobian),
// so a change of eps between continuation stages needs no re-evaluation.
template <typename T, int NV, int KPL> struct SapPoint {
  T zs[KPL][3];
  Proj<T> pr[KPL];  // pr.g = gs
  T yT[NV], My[NV];
  T jtg[NV];        // sum_c J_c^T D_mu gs_c, summed over the lane group; [5] = sum of the normal components (identity block)
  T nsum;           // sum of the normal components of gs (= jtg[5] when every Jacobian has the identity block)
};
template <class JT> struct JacIsDense { static constexpr bool value = false; };
template <typename T, int NJ> struct JacIsDense<CJac<T, NJ, true>> { static constexpr bool value = true; };

template <typename T, typename TA, int NJ, int KPL, class Lanes, class JT>
DPLL_HD void sap_evaluate(const T (&M)[6 + NJ][6 + NJ], const JT (&Jc)[KPL], const T (&mu)[KPL],
                          const T (&qc)[KPL][3], const TA (&y)[6 + NJ], SapPoint<T, 6 + NJ, KPL>& p) {
  constexpr int NV = 6 + NJ;
  DPLL_UNROLL for (int i = 0; i < NV; ++i) { p.yT[i] = T(y[i]); p.jtg[i] = T(0); }
  T ns = T(0);
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    TA jy[3];
    cjac_apply<T, TA, NJ>(Jc[c], y, jy);
    p.zs[c][0] = -T(TA(mu[c]) * jy[0] + TA(qc[c][0]));
    p.zs[c][1] = -T(TA(mu[c]) * jy[1] + TA(qc[c][1]));
    p.zs[c][2] = -T(jy[2] + TA(qc[c][2]));
    lorentz_project(p.zs[c], p.pr[c]);
    const T a[3] = {mu[c] * p.pr[c].g[0], mu[c] * p.pr[c].g[1], p.pr[c].g[2]};
    cjac_apply_t_add<T, NJ>(Jc[c], a, p.jtg);
    ns += p.pr[c].g[2];
  }
  symv<T, NV>(M, p.yT, p.My);
  DPLL_UNROLL for (int i = 0; i < NV; ++i) p.jtg[i] = Lanes::group_sum(p.jtg[i]);
  if constexpr (JacIsDense<JT>::value) p.nsum = Lanes::group_sum(ns);
  else p.nsum = p.jtg[5];
}

// The state at y + a d from the state at y WITHOUT going back to y: zs moves by -a (J d) and M y by a (M d).  The float
// kernels iterate this way: forming J y + q afresh cancels O(1) terms down to O(eps |f|) and needs double, while the
// increment J d is itself small -- an update in float loses 1 ulp of zs per iteration and nothing to cancellation -- so
// the iteration carries no double arithmetic at all (the residual starts exactly: y = 0 gives zs = -q).
template <typename T, int NJ, int KPL, class Lanes, class JT>
DPLL_HD void sap_advance(const JT (&Jc)[KPL], const T (&mu)[KPL], const SapPoint<T, 6 + NJ, KPL>& cur,
                         const T (&jd)[KPL][3], const T (&Md)[6 + NJ], const T (&d)[6 + NJ], T a,
                         SapPoint<T, 6 + NJ, KPL>& p) {
  constexpr int NV = 6 + NJ;
  DPLL_UNROLL for (int i = 0; i < NV; ++i) { p.yT[i] = cur.yT[i] + a * d[i]; p.My[i] = cur.My[i] + a * Md[i]; p.jtg[i] = T(0); }
  T ns = T(0);
  DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
    DPLL_UNROLL for (int r = 0; r < 3; ++r) p.zs[c][r] = cur.zs[c][r] - a * jd[c][r];
    lorentz_project(p.zs[c], p.pr[c]);
    const T g[3] = {mu[c] * p.pr[c].g[0], mu[c] * p.pr[c].g[1], p.pr[c].g[2]};
    cjac_apply_t_add<T, NJ>(Jc[c], g, p.jtg);
    ns += p.pr[c].g[2];
  }
  DPLL_UNROLL for (int i = 0; i < NV; ++i) p.jtg[i] = Lanes::group_sum(p.jtg[i]);
  if constexpr (JacIsDense<JT>::value) p.nsum = Lanes::group_sum(ns);
  else p.nsum = p.jtg[5];
}

// One iteration: Newton direction d from the state at y, then the state at y + d is evaluated -- it is both the line
// search's probe of alpha = 1 (l'(1) = grad(y + d) . d) and, when that step is accepted (99 % of the item-iterations on
// the benchmark batch), the next iteration's starting state, so an accepted iteration costs one evaluation, one
// Hessian and one Cholesky and nothing else.  Only when some item of the wave rejects alpha = 1 does the wave run the
// derivative-based safeguarded search (re-projections of zs - alpha J d) and evaluate the state again at y + alpha d.
template <typename T, typename TA, int NJ, int KPL, class Lanes, class JT>
DPLL_HD int sap_newton(const T (&M)[6 + NJ][6 + NJ], const JT (&Jc)[KPL], const T (&mu)[KPL],
                       const T (&qc)[KPL][3], T eps, const SolverOpts& opt, TA (&y)[6 + NJ], T (&f)[KPL][3],
                       const T (&y0)[6 + NJ], bool use_y0, bool race = false, bool* winner = nullptr, bool participate = true) {
  constexpr int NV = 6 + NJ;
  // float storage with a wider accumulation type: iterate on increments, all in float (sap_advance)
  constexpr bool kIncremental = DPLL_INCREMENTAL && sizeof(T) < sizeof(TA);
  const T tol2 = T(opt.tol * opt.tol), stol2 = T(opt.stall_tol * opt.stall_tol), ls_tol = T(opt.ls_tol);
  const T stage_tol2 = T(opt.stage_tol * opt.stage_tol);
  // Racing schedules (SolverOpts::portfolio): the lane group of an item exists Lanes::kVariants times in the wave; copy 0
  // runs the caller's continuation schedule, the others the schedules of opt.race_*.  The copies solve the same problem
  // to the same tolerance in lock step (no extra time per iteration); the item is finished when any copy is.
  int n_stages = opt.n_stages;
  T factor = T(opt.stage_factor), inv_factor = T(1.0 / opt.stage_factor);
  // a racing copy may run without the line search (cheaper: a rejected step makes the whole wave run the fall-back code): it
  // is then not globally convergent, which costs nothing -- copy 0 is, and a copy only counts once it has met the
  // stopping rule with finite numbers (the objective is strictly convex: whoever meets it is at the same minimiser)
  bool full_steps = false;
  if constexpr (Lanes::kVariants > 1) {
    if (race) {
      const int vr = Lanes::variant();
      DPLL_UNROLL for (int k = 1; k < Lanes::kVariants && k <= 3; ++k) {
        n_stages = vr == k ? opt.race_stages[k - 1] : n_stages;
        factor = vr == k ? T(opt.race_factor[k - 1]) : factor;
        inv_factor = vr == k ? T(1) / T(opt.race_factor[k - 1]) : inv_factor;
        use_y0 = vr == k ? (opt.race_flags[k - 1] & 1) != 0 : use_y0;
        full_steps = vr == k ? (opt.race_flags[k - 1] & 2) != 0 : full_steps;
      }
    }
  }
  const int last_stage = n_stages - 1;
  T eps_c = eps;  // regularisation of the current stage (per item: items advance independently)
  if constexpr (Lanes::kVariants > 1) {
    DPLL_UNROLL for (int s = 0; s < kRaceMaxStages; ++s) eps_c *= s < last_stage ? factor : T(1);  // (per lane: no divergent loop)
  } else {
    for (int s = 0; s < last_stage; ++s) eps_c *= factor;
  }
  int stage = 0, it_stage = 0;
  bool active = participate;  // (racing copies: the refinement phase of a double solve runs the float phase's winner only)
  bool finished = false;  // this copy met the stopping rule itself (racing: it may also end because another copy did)
  int iters = 0;
  DPLL_UNROLL for (int i = 0; i < NV; ++i) y[i] = use_y0 ? TA(y0[i]) : TA(0);
  // every lane of the group starts its share of the Hessian from M / group size (a power of two, so exact): the
  // group sum then returns M + sum_c ... without a separate addition per entry
  T Mshare[NV][NV];
  DPLL_UNROLL for (int i = 0; i < NV; ++i)
    DPLL_UNROLL for (int j = 0; j <= i; ++j) Mshare[i][j] = M[i][j] * T(1.0 / Lanes::kGroup);
  T best = T(3.0e38);
  int stall = 0;
  // two point states that swap roles every iteration (the loop below is unrolled by two so that "the state at y + d
  // becomes the current state" costs no register moves)
  SapPoint<T, NV, KPL> pa, pb;
  sap_evaluate<T, TA, NJ, KPL, Lanes>(M, Jc, mu, qc, y, pa);
  DPLL_PHASE_BEGIN();
  // one iteration from the state `cur` at y; leaves the state at the new y in `trial`; false = every item of the wave is done
  auto iterate = [&](int it, const SapPoint<T, NV, KPL>& cur, SapPoint<T, NV, KPL>& trial) -> bool {
    DPLL_PHASE(5);
    const T ieps = fast_rcp(eps_c);
    const bool final_stage = stage >= last_stage;
    T grad[NV];
    DPLL_UNROLL for (int i = 0; i < NV; ++i) grad[i] = cur.My[i] - ieps * cur.jtg[i];
    // the sum of the normal forces is zero exactly when every contact sits in the polar region
    const bool any_force = cur.nsum > T(0);
    // Hessian H = M + sum_c [A 1 j]^T C [A 1 j],  C = D_mu dP D_mu / eps
    T H[NV][NV];
    DPLL_UNROLL for (int i = 0; i < NV; ++i)
      DPLL_UNROLL for (int j = 0; j <= i; ++j) H[i][j] = Mshare[i][j];
    DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
      T dP[6];
      proj_jacobian(cur.pr[c], dP);
      const T m1 = mu[c] * ieps, m2 = mu[c] * m1;
      const T C[3][3] = {{dP[0] * m2, dP[3] * m2, dP[4] * m1}, {dP[3] * m2, dP[1] * m2, dP[5] * m1}, {dP[4] * m1, dP[5] * m1, dP[2] * ieps}};
      hessian_add<T, NJ>(Jc[c], C, H);
    }
    DPLL_UNROLL for (int i = 0; i < NV; ++i)
      DPLL_UNROLL for (int j = 0; j <= i; ++j) {
        H[i][j] = Lanes::group_sum(H[i][j]);
        H[j][i] = H[i][j];
      }
    DPLL_PHASE(0);
    T L[NV][NV], invd[NV], d[NV];
    cholesky_fast<T, NV>(H, L, invd);
    chol_solve<T, NV>(L, invd, grad, d);
    DPLL_UNROLL for (int i = 0; i < NV; ++i) d[i] = -d[i];
    DPLL_PHASE(1);
    // Newton decrement and stopping rule (the step below is still taken: it only improves y)
    const T dec2 = -dotn<T, NV>(grad, d);
    const T ynorm2 = dotn<T, NV>(cur.yT, cur.My);
    const T scale = T(1) + fast_sqrt(tmax(ynorm2, T(0)));
    const T scale2 = scale * scale;
    const bool converged = !(dec2 > (final_stage ? tol2 : stage_tol2) * scale2);
    const bool improved = dec2 < T(0.25) * best;  // decrement still halving?
    stall = improved ? 0 : stall + 1;
    best = tmin(best, dec2);
    const bool stalled = stall >= 3 && !(dec2 > (final_stage ? stol2 : stage_tol2) * scale2);
    // no force at all and y stationary: the answer (y = 0 region-wise) does not depend on eps, skip the other stages
    const bool force_free = !(dec2 > T(0)) && !any_force;
    const bool moving = active && (dec2 > T(0));
    const T slope_tol = (final_stage ? ls_tol : T(opt.stage_ls_tol)) * dec2;  // |l'(0)| = dec2
    const int ls_full = final_stage ? opt.max_ls : opt.stage_max_ls;
    const int ls_cap = (opt.fast_ls > 0 && stall < 2) ? (opt.fast_ls < ls_full ? opt.fast_ls : ls_full) : ls_full;
    DPLL_PHASE(2);
    // The state at y + d: l'(1) = grad(y + d) . d.
    TA yt[NV];
    T alpha = T(1);
    T jd[KPL][3], Md[NV];
    if constexpr (kIncremental) {
      DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
        T t[3];
        cjac_apply<T, T, NJ>(Jc[c], d, t);
        jd[c][0] = mu[c] * t[0];
        jd[c][1] = mu[c] * t[1];
        jd[c][2] = t[2];
      }
      symv<T, NV>(M, d, Md);
      sap_advance<T, NJ, KPL, Lanes>(Jc, mu, cur, jd, Md, d, moving ? T(1) : T(0), trial);
    } else {
      DPLL_UNROLL for (int i = 0; i < NV; ++i) yt[i] = y[i] + (moving ? TA(d[i]) : TA(0));
      sap_evaluate<T, TA, NJ, KPL, Lanes>(M, Jc, mu, qc, yt, trial);
    }
    T first1 = T(0);
    DPLL_UNROLL for (int i = 0; i < NV; ++i) first1 += (trial.My[i] - ieps * trial.jtg[i]) * d[i];
    // alpha = 1 stands when the slope there is small enough, and also -- while the search is capped at one probe --
    // when l is still descending at 1 (the capped search would stop at its lower bracket, alpha = 1)
    const bool reject = moving && !full_steps && !(tabs(first1) <= slope_tol) && (first1 > T(0) || ls_cap > 1);
    DPLL_PHASE(3);
    const bool fell_back = Lanes::wave_any(reject);
    if (fell_back) {
      DPLL_PHASE_COUNT(6);
      // l'(alpha) = y.Md + alpha d.Md - sum_c gamma_c(alpha) . (J_c d); from H d = -grad:
      //   d.Md = dec2 - (1/eps) sum_c (J_c d)^T dP_c (J_c d),      l''(alpha) = d.Md + (1/eps) sum_c (J_c d)^T dP_c(alpha) (J_c d)
      T curv = T(0), curv1 = T(0);
      DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
        if constexpr (!kIncremental) {
          T t[3];
          cjac_apply<T, T, NJ>(Jc[c], d, t);
          jd[c][0] = mu[c] * t[0];
          jd[c][1] = mu[c] * t[1];
          jd[c][2] = t[2];
          curv += proj_quadratic(cur.pr[c], jd[c]);
        }
        curv1 += proj_quadratic(trial.pr[c], jd[c]);
      }
      T dMd;
      if constexpr (kIncremental) dMd = tmax(dotn<T, NV>(d, Md), T(0));
      else dMd = tmax(dec2 - ieps * Lanes::group_sum(curv), T(0));
      // capped search, l'(1) > 0 (overshoot): one safeguarded Newton step on l' back from alpha = 1, whose projections
      // are the trial state's
      {
        const T second1 = dMd + ieps * Lanes::group_sum(curv1);
        const T newton = T(1) - first1 * fast_rcp(second1);
        const T capped = (newton > T(0) && newton < T(1)) ? newton : T(0.5);
        alpha = reject ? capped : alpha;
      }
      const bool full = reject && ls_cap > 1;
      if (Lanes::wave_any(full)) {
        // stalled items: the derivative-based bracketing search, re-projecting the cone residuals at every probe
        const T yMd = dotn<T, NV>(cur.My, d);
        T lo = T(0), hi = T(-1);  // hi < 0: no upper bracket yet
        bool searching = full;
        T a_s = T(1);
        // one probe: l'(a) and l''(a) by re-projecting the cone residuals, then a safeguarded Newton step on l'
        for (int ls = 0; ls < opt.max_ls; ++ls) {
          if (!Lanes::wave_any(searching)) break;
          DPLL_PHASE_COUNT(6);
          T part1 = T(0), part2 = T(0);
          DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
            const T za[3] = {cur.zs[c][0] - a_s * jd[c][0], cur.zs[c][1] - a_s * jd[c][1], cur.zs[c][2] - a_s * jd[c][2]};
            Proj<T> pa;
            lorentz_project(za, pa);
            part1 += pa.g[0] * jd[c][0] + pa.g[1] * jd[c][1] + pa.g[2] * jd[c][2];
            part2 += proj_quadratic(pa, jd[c]);
          }
          const T first = yMd + a_s * dMd - ieps * Lanes::group_sum(part1);
          const T second = dMd + ieps * Lanes::group_sum(part2);
          bool ok = tabs(first) <= slope_tol;
          const T lo_n = first < T(0) ? a_s : lo;
          const T hi_n = first >= T(0) ? a_s : hi;
          const T newton = a_s - first * fast_rcp(second);
          const T mid = hi_n < T(0) ? T(2) * a_s : T(0.5) * (lo_n + hi_n);
          const bool bad = !((newton > lo_n) && (hi_n < T(0) || newton < hi_n));
          const T nxt = bad ? mid : newton;
          ok = ok || (hi_n >= T(0) && (hi_n - lo_n) <= T(4) * (sizeof(T) == 4 ? T(1.2e-7) : T(2.3e-16)) * hi_n);
          // out of probes: fall back to the largest step known to decrease l (l' < 0 on [0, lo])
          const bool out = searching && !ok && (ls + 1 >= ls_cap);
          lo = searching ? lo_n : lo;
          hi = searching ? hi_n : hi;
          a_s = (searching && !ok) ? (out ? (lo_n > T(0) ? lo_n : nxt) : nxt) : a_s;
          searching = searching && !ok && !out;
        }
        alpha = full ? a_s : alpha;
      }
      // The state at y + alpha d.  Items that kept alpha = 1 recompute the state they already have (same expressions,
      // same values): an item's result does not depend on which other items share its wave.
      if constexpr (kIncremental) {
        sap_advance<T, NJ, KPL, Lanes>(Jc, mu, cur, jd, Md, d, moving ? alpha : T(0), trial);
      } else {
        DPLL_UNROLL for (int i = 0; i < NV; ++i) yt[i] = y[i] + (moving ? TA(alpha) * TA(d[i]) : TA(0));
        sap_evaluate<T, TA, NJ, KPL, Lanes>(M, Jc, mu, qc, yt, trial);
      }
    }
    DPLL_PHASE(4);
    DPLL_PHASE_EVENT(7, fell_back);
#if defined(DPLL_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
    if (active) {
      printf("  it %2d stage %d eps %.2e dec2 %.3e scale %.3e alpha %.4f conv %d stall %d regions", it, stage, double(eps_c), double(dec2), double(scale), double(alpha), int(converged), stall);
      for (int c = 0; c < KPL; ++c) printf(" %c", cur.pr[c].inside ? 'I' : (cur.pr[c].polar ? '0' : 'M'));
      printf("\n");
    }
#endif
    DPLL_ITER_HOOK(it, moving, alpha);
    if constexpr (!kIncremental) {
      DPLL_UNROLL for (int i = 0; i < NV; ++i) y[i] = yt[i];
    }
    iters = active ? it + 1 : iters;
    const bool stage_done = converged || stalled || (!final_stage && it_stage + 1 >= opt.stage_max_iter);
    const bool advance = active && !final_stage && stage_done && !force_free;
    const bool ends = (final_stage && stage_done) || force_free;
    bool sound = true;
    if constexpr (Lanes::kVariants > 1) {
      // racing copies other than the first: only a decrement that is a number and under the tolerance counts (copy 0
      // keeps the rule above, which also ends a solve that stalled or went to NaN -- the invalid-solve mask's business)
      const T limit = tol2 * scale2;
      sound = Lanes::variant() == 0 || (final_stage && dec2 <= limit) || (force_free && dec2 <= T(0));
    }
    finished = finished || (active && ends && sound);
    active = active && !ends;
    if constexpr (Lanes::kVariants > 1) {
      // (evaluated by every lane, outside any short-circuit: a cross-lane read of a lane that skipped it returns zero)
      const int any_finished = Lanes::item_or(finished ? 1 : 0);
      active = active && any_finished == 0;
    }
    stage = advance ? stage + 1 : stage;
    eps_c = advance ? eps_c * inv_factor : eps_c;
    it_stage = advance ? 0 : it_stage + 1;
    best = advance ? T(3.0e38) : best;
    stall = advance ? 0 : stall;
    return Lanes::wave_any(active);
  };
  bool in_b = false;  // which of the two holds the final state (wave uniform)
  for (int it = 0; it < opt.max_iter; it += 2) {
    in_b = true;
    if (!iterate(it, pa, pb)) break;
    if (it + 1 >= opt.max_iter) break;
    in_b = false;
    if (!iterate(it + 1, pb, pa)) break;
  }
  DPLL_PHASE_END();
  if constexpr (kIncremental) {
    DPLL_UNROLL for (int i = 0; i < NV; ++i) y[i] = TA(in_b ? pb.yT[i] : pa.yT[i]);
  }
  // forces at the final iterate, with the reference's eps: f = P_K(zs / eps) = gs / eps
  const T ieps = fast_rcp(eps);
  DPLL_UNROLL for (int c = 0; c < KPL; ++c)
    DPLL_UNROLL for (int r = 0; r < 3; ++r) f[c][r] = (in_b ? pb.pr[c].g[r] : pa.pr[c].g[r]) * ieps;
  if constexpr (Lanes::kVariants > 1) {
    // the finished copy with the lowest index supplies the item's outputs (none finished: max_iter ran out, copy 0 does)
    const int done_bits = Lanes::item_or(finished ? (1 << Lanes::variant()) : 0);
    const int part_bits = Lanes::item_or(participate ? (1 << Lanes::variant()) : 0);
    const int first = __builtin_ctz((done_bits != 0 ? done_bits : part_bits) | (1 << Lanes::kVariants));
    if (winner) *winner = Lanes::variant() == first;
  } else {
    if (winner) *winner = true;
  }
  return iters;
}

// Double-precision solves by mixed-precision refinement: the continuation stages and the hunt for the active set run in
// float (the float build's iteration: ~1 us against ~2.5 us per iteration in double), then the double solver starts from
// that point at the reference's eps with its own stopping rule (decrement <= 1e-13 relative): Newton converges
// quadratically from a 1e-6-accurate start inside the right active set, so two to three double iterations remain of
// fourteen.  The result satisfies the same criterion as an all-double solve; `iters` counts both phases.
#ifndef DPLL_MIXED_F64
#define DPLL_MIXED_F64 1
#endif
template <typename T, int NJ> DPLL_HD void cjac_to_float(const CJac<T, NJ, false>& a, CJac<float, NJ, false>& b) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) DPLL_UNROLL for (int c = 0; c < 3; ++c) b.A[r][c] = float(a.A[r][c]);
  DPLL_UNROLL for (int j = 0; j < (NJ > 0 ? NJ : 1); ++j) DPLL_UNROLL for (int r = 0; r < 3; ++r) b.j[j][r] = float(a.j[j][r]);
}
template <typename T, int NJ> DPLL_HD void cjac_to_float(const CJac<T, NJ, true>& a, CJac<float, NJ, true>& b) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) DPLL_UNROLL for (int c = 0; c < 6 + NJ; ++c) b.m[r][c] = float(a.m[r][c]);
}
template <typename T, typename TA, int NJ, int KPL, class Lanes, bool DENSE>
DPLL_HD int sap_solve(const T (&M)[6 + NJ][6 + NJ], const CJac<T, NJ, DENSE> (&Jc)[KPL], const T (&mu)[KPL],
                      const T (&qc)[KPL][3], T eps, const SolverOpts& opt, TA (&y)[6 + NJ], T (&f)[KPL][3],
                      const T (&y0)[6 + NJ], bool use_y0, bool race = false, bool* winner = nullptr) {
  constexpr int NV = 6 + NJ;
  if constexpr (DPLL_MIXED_F64 && sizeof(T) == 8 && sizeof(TA) == 8) {
    if (opt.f64_refine == 0) return sap_newton<T, TA, NJ, KPL, Lanes>(M, Jc, mu, qc, eps, opt, y, f, y0, use_y0, race, winner);
    float Mf[NV][NV], muf[KPL], qcf[KPL][3], ff[KPL][3], y0f[NV];
    CJac<float, NJ, DENSE> Jf[KPL];
    DPLL_UNROLL for (int i = 0; i < NV; ++i) {
      y0f[i] = float(y0[i]);
      DPLL_UNROLL for (int j = 0; j < NV; ++j) Mf[i][j] = float(M[i][j]);
    }
    DPLL_UNROLL for (int c = 0; c < KPL; ++c) {
      cjac_to_float<T, NJ>(Jc[c], Jf[c]);
      muf[c] = float(mu[c]);
      DPLL_UNROLL for (int r = 0; r < 3; ++r) qcf[c][r] = float(qc[c][r]);
    }
    SolverOpts coarse = opt;
    coarse.tol = 1e-6;
    coarse.stall_tol = 1e-5;
    double yc[NV];
    // (racing copies: the float phase races the schedules, its winner alone is refined in double -- the other copies hold
    // points that are not converged and would only send the wave through the line search's fall-back code)
    bool coarse_winner = true;
    const int it_coarse = sap_newton<float, double, NJ, KPL, Lanes>(Mf, Jf, muf, qcf, float(eps), coarse, yc, ff, y0f, use_y0, race, &coarse_winner);
    SolverOpts fine = opt;
    fine.n_stages = 1;
    T start[NV];
    DPLL_UNROLL for (int i = 0; i < NV; ++i) start[i] = T(yc[i]);
    return it_coarse + sap_newton<T, TA, NJ, KPL, Lanes>(M, Jc, mu, qc, eps, fine, y, f, start, true, false, winner, coarse_winner);
  } else {
    return sap_newton<T, TA, NJ, KPL, Lanes>(M, Jc, mu, qc, eps, opt, y, f, y0, use_y0, race, winner);
  }
}

}  // namespace dpll
