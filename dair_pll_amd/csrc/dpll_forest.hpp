// dpll_forest.hpp -- the FOREST build of the contact-dynamics path: per-item math for systems beyond the register-resident
// builds of dpll_core.hpp -- several floating (or fixed) bases in one system, any number of joints, geometries and body-body
// candidates up to the limits below.
//
// What the reference handles generically in Python: a MultibodyPlant with several models (init_urdfs: Dict[str, str],
// multibody_learnable_system.py:51-54; one FloatingBaseSpace / FixedBaseSpace per model in a ProductSpace,
// drake_utils.py:309-335, state_space.py:650-730), any tree, any number of collision geometries and candidates
// (multibody_terms.py:286-382, 428-521).  The specialised builds keep an item's dense blocks in the registers of 4-16
// lanes, fully unrolled per (joints, geometries) instantiation: that does not scale past an 8 x 8 Newton system.  Here ONE
// WAVE works on ONE item at a time, everything an item needs lives in LDS with run-time sizes (one kernel per dtype, no
// template ranges), lanes take the bodies / contacts / matrix entries of a phase in strides, phases are separated by a
// wave-level barrier, and the dense linear algebra (Cholesky, triangular solves) is done column by column by the wave.
// The same source runs on the host with a team of ONE (tests/hostsim/forestsim.cpp), where it is held against the oracle and
// the reference-run fixtures in the CPU container.
//
// Math (file:line under /root/reference/dair_pll as in dpll_core.hpp): composite-rigid-body mass matrix and recursive
// Newton-Euler bias forces on the forest, contact Jacobians dense in the generalized velocity [per model: omega_body, v_world,
// joint rates] (state_space.py:412-424, 650-730), the cone solve of dpll_core.hpp's sap_newton (same continuation, same
// stopping rule, iterating on increments), contactnets_loss with its hand-derived adjoint, forward_dynamics + the Lie-group
// Euler update.
#pragma once

#include "dpll_core.hpp"

// run-time loops over LDS-resident blocks: four trips in flight at once (a trip alone waits out an LDS round trip per load)
#if defined(__clang__)
#define DPLL_PIPE _Pragma("unroll 4")
#else
#define DPLL_PIPE
#endif

// diagnostic builds (tools/diag/forest_stamps.py) time the phases of an item with the shader clock; nothing in the shipped build
#ifndef DPLL_FSTAMP
#define DPLL_FSTAMP(slot) do {} while (0)
#endif

namespace dpll_forest {

using namespace dpll;

constexpr int kMaxBodies = 16;
constexpr int kMaxGeoms = 12;
constexpr int kMaxPairs = 16;
constexpr int kMaxContacts = 64;   // 4 per box / polygon, 1 per sphere, 1 per candidate
constexpr int kMaxV = 32;          // generalized velocities
constexpr int kMaxX = 2 * kMaxV + kMaxBodies;  // an upper bound of n_q + n_v
constexpr int kJointFloating = 2;  // a model's root: free in the world (quaternion + position; omega_body, v_world)
constexpr int kJointFixed = 3;     // a model's root welded to the world (FixedBaseSpace, state_space.py:556-647): no coordinates
constexpr int kGeoStride = 3 * kMaxPolyVerts;

// Plain-old-data description (include/dpll.h: dpll_forest_desc_t); lives in device memory, read through uniform loads
struct ForestDesc {
  int32_t n_bodies, n_geoms, n_pairs, n_contacts, n_q, n_v, inertia_mode, rotated, max_depth, n_u;
  double dt, gravity_z;
  int32_t parent[kMaxBodies];       // -1: the world
  int32_t joint_kind[kMaxBodies];   // kJointRevolute | kJointPrismatic | kJointFloating | kJointFixed
  int32_t q_index[kMaxBodies];      // first coordinate of the body's joint in q (floating: quaternion wxyz, then position)
  int32_t v_index[kMaxBodies];      // first velocity of the body's joint in v (floating: omega_body, then v_world)
  int32_t depth[kMaxBodies];        // 0 for roots
  double joint_origin[kMaxBodies][3];  // in the parent's frame (a fixed root: in the world)
  double joint_axis[kMaxBodies][3];
  double body_rot[kMaxBodies][3][3];
  int32_t dof_body[kMaxV];          // velocity i belongs to the joint of this body
  int32_t geom_body[kMaxGeoms], geom_kind[kMaxGeoms], geom_nverts[kMaxGeoms];
  double geom_origin[kMaxGeoms][3];
  double geom_rot[kMaxGeoms][3][3];
  int32_t pair_a[kMaxPairs], pair_b[kMaxPairs];
  // contact c: witness contact_slot[c] of geometry contact_geom[c] against the ground, or (contact_geom[c] < 0) the contact of
  // candidate contact_slot[c]; in the reference's order (geometries in order, then the candidates)
  int32_t contact_geom[kMaxContacts], contact_slot[kMaxContacts];
  int32_t act_body[kMaxV];          // actuator k drives the joint of this body (B u of lagrangian_forces, multibody_terms.py:142-146)
};

// ---- the team that works on an item -----------------------------------------------------------------------------------------
struct HostTeam {
  static constexpr int kSize = 1;
  static DPLL_HD int rank() { return 0; }
  static DPLL_HD void sync() {}
  template <typename T> static DPLL_HD T sum(T x) { return x; }
  static DPLL_HD bool any(bool x) { return x; }
  static DPLL_HD bool wave_any(bool x) { return x; }  // over every team that shares the wavefront (the device: 1 or 4 teams)
  using Lanes = OneLane;  // the lane-group policy dpll_core.hpp's direction search takes
  static constexpr bool kLaneRows = false;  // (the device teams factor dense matrices with a row per lane in registers)
  static constexpr int kRowsMax = 0;
};

// ---- an item's storage (LDS on the device) ---------------------------------------------------------------------------------
template <typename S> struct ContactRec {
  S phi, mu;
  S R[3][3];       // rotation of the contact's (B side) geometry frame
  S sgn[3], drad[3];
  S Ra[3][3], F[3][3], dir[3], sgn_a[3], drad_a[3];  // candidates: A's geometry frame, contact frame (rows), direction in A
  S qc[3], slide[2], speed, jpv[3];                   // the loss's cone offset and what its adjoint reuses
  S rbar[3], rbar_a[3], gmu;                          // adjoint: witness adjoints (B side, A side) and d/d mu
  int32_t body, geom, vidx, body_a, geom_a, vidx_a, pair;  // pair: candidate index or -1
};
// one point of the cone solve (dpll_core.hpp SapPoint), per contact
template <typename S> struct ConePoint {
  S zs[3], g[3], that[2], cp, a, b;
  int32_t polar;
};

struct Bump {
  char* base;
  size_t off;
  template <class X> DPLL_HD X* take(size_t n) {
    off = (off + 7) & ~(size_t)7;
    X* p = base ? reinterpret_cast<X*>(base + off) : nullptr;
    off += n * sizeof(X);
    return p;
  }
};

template <typename S, typename SA> struct Arena {
  int nb, nv, nq, K, ng, np;
  SA *q;                                // coordinates at which the terms are evaluated
  S *v;
  SA *R, *o, *Rpc, *pj, *axw;           // kinematics [nb][9|3]
  S *iota, *mu_g, *mu_p, *habs;
  S *Vw, *Vu, *AGw, *AGu, *Wn, *Wf, *comp;
  S *M, *LM, *invdM, *a, *F;
  S *tau;                               // [nv] actuation B u of the call (zeros for an unactuated system)
  S *J;                                 // [K][3][nv]
  ContactRec<S>* ct;
  SA* dirs;                             // [np][3]: candidate directions in the frame of A
  SA *setA, *setB;                      // vertex sets of the direction search [8][3]
  // cone solve
  S *H, *invd, *CJ, *Cc;                // Newton matrix / factor, C J, the contacts' 3 x 3 blocks (6)
  S *y0, *My0, *jtg0, *y1, *My1, *jtg1; // two points that swap
  ConePoint<S> *p0, *p1;
  S *grad, *d, *Md, *jd, *tmp, *tmp2;   // [nv] (jd: [K][3])
  S *force;                             // [K][3]
  // loss / adjoint scratch
  S *dv, *gv, *w, *Mdv, *u, *abar, *bvec, *vp;
  S *tw;                                // body twists of up to four vectors [4][nb][6]
  S *scal;                              // a few scalars shared by the team
  S *cphi, *cmu;                        // the contacts' signed distances and friction coefficients (also in the records)
  short *tri_i, *tri_j;                 // entry e of a lower triangle in row-major order -> (row, column); filled by derive()
  // lite: an arena for terms + contact geometry only (the dual passes of the state adjoint): no contact records, no solver
  // or adjoint blocks
  DPLL_HD size_t carve(char* base, int nb_, int nv_, int nq_, int K_, int ng_, int np_, bool lite = false) {
    nb = nb_; nv = nv_; nq = nq_; K = K_; ng = ng_; np = np_;
    Bump m{base, 0};
    if (lite) {
      q = m.take<SA>(nq); v = m.take<S>(nv);
      R = m.take<SA>(9 * nb); o = m.take<SA>(3 * nb); Rpc = m.take<SA>(9 * nb); pj = m.take<SA>(3 * nb); axw = m.take<SA>(3 * nb);
      iota = m.take<S>(kIota * nb); mu_g = m.take<S>(ng); mu_p = m.take<S>(np > 0 ? np : 1); habs = m.take<S>(3 * ng);
      Vw = m.take<S>(3 * nb); Vu = m.take<S>(3 * nb); AGw = m.take<S>(3 * nb); AGu = m.take<S>(3 * nb);
      Wn = m.take<S>(3 * nb); Wf = m.take<S>(3 * nb); comp = m.take<S>(kIota * nb);
      M = m.take<S>(nv * nv); LM = m.take<S>(nv * nv); invdM = m.take<S>(nv); a = m.take<S>(nv); F = m.take<S>(nv); tau = m.take<S>(nv);
      J = m.take<S>((size_t)K * 3 * nv);
      ct = nullptr;
      dirs = m.take<SA>(3 * (np > 0 ? np : 1)); setA = setB = nullptr;
      H = invd = CJ = Cc = y0 = My0 = jtg0 = y1 = My1 = jtg1 = nullptr;
      p0 = p1 = nullptr;
      grad = d = Md = jd = tmp2 = force = dv = gv = w = Mdv = u = abar = bvec = scal = nullptr;
      tmp = m.take<S>(nv); vp = m.take<S>(nv);
      tw = m.take<S>((size_t)nb * 6);
      cphi = m.take<S>(K); cmu = m.take<S>(K);
      tri_i = m.take<short>(nv * (nv + 1) / 2); tri_j = m.take<short>(nv * (nv + 1) / 2);
      return (m.off + 15) & ~(size_t)15;
    }
    q = m.take<SA>(nq); v = m.take<S>(nv);
    R = m.take<SA>(9 * nb); o = m.take<SA>(3 * nb); Rpc = m.take<SA>(9 * nb); pj = m.take<SA>(3 * nb); axw = m.take<SA>(3 * nb);
    iota = m.take<S>(kIota * nb); mu_g = m.take<S>(ng); mu_p = m.take<S>(np > 0 ? np : 1); habs = m.take<S>(3 * ng);
    Vw = m.take<S>(3 * nb); Vu = m.take<S>(3 * nb); AGw = m.take<S>(3 * nb); AGu = m.take<S>(3 * nb);
    Wn = m.take<S>(3 * nb); Wf = m.take<S>(3 * nb); comp = m.take<S>(kIota * nb);
    M = m.take<S>(nv * nv); LM = m.take<S>(nv * nv); invdM = m.take<S>(nv); a = m.take<S>(nv); F = m.take<S>(nv); tau = m.take<S>(nv);
    J = m.take<S>((size_t)K * 3 * nv);
    ct = m.take<ContactRec<S>>(K);
    dirs = m.take<SA>(3 * (np > 0 ? np : 1)); setA = m.take<SA>(3 * kMaxPolyVerts); setB = m.take<SA>(3 * kMaxPolyVerts);
    H = m.take<S>(nv * nv); invd = m.take<S>(nv); CJ = m.take<S>((size_t)K * 3 * nv); Cc = m.take<S>(6 * K);
    y0 = m.take<S>(nv); My0 = m.take<S>(nv); jtg0 = m.take<S>(nv); y1 = m.take<S>(nv); My1 = m.take<S>(nv); jtg1 = m.take<S>(nv);
    p0 = m.take<ConePoint<S>>(K); p1 = m.take<ConePoint<S>>(K);
    grad = m.take<S>(nv); d = m.take<S>(nv); Md = m.take<S>(nv); jd = m.take<S>(3 * K); tmp = m.take<S>(nv); tmp2 = m.take<S>(nv);
    force = m.take<S>(3 * K);
    dv = m.take<S>(nv); gv = m.take<S>(nv); w = m.take<S>(nv); Mdv = m.take<S>(nv); u = m.take<S>(nv); abar = m.take<S>(nv);
    bvec = m.take<S>(nv); vp = m.take<S>(nv);
    tw = m.take<S>((size_t)4 * nb * 6);
    scal = m.take<S>(16);
    cphi = m.take<S>(K); cmu = m.take<S>(K);
    tri_i = m.take<short>(nv * (nv + 1) / 2); tri_j = m.take<short>(nv * (nv + 1) / 2);
    return (m.off + 15) & ~(size_t)15;
  }
};

template <typename S, typename SA> DPLL_HD size_t arena_bytes(const ForestDesc& fd, bool lite = false) {
  Arena<S, SA> a;
  return a.carve(nullptr, fd.n_bodies, fd.n_v, fd.n_q, fd.n_contacts, fd.n_geoms, fd.n_pairs, lite);
}

// [loss | d/d iota (nb, 10) | d/d mu of the ground pairs (ng) | d/d mu of the candidates (np) | d/d geometry blocks (ng, 24)]
DPLL_HD int row_width(const ForestDesc& fd) { return 1 + kIota * fd.n_bodies + fd.n_geoms + fd.n_pairs + kGeoStride * fd.n_geoms; }
DPLL_HD int param_count(const ForestDesc& fd) { return kIota * fd.n_bodies + 1 + fd.n_geoms + kGeoStride * fd.n_geoms; }

template <typename X, typename Y> DPLL_HD void load3(const X* p, Y (&out)[3]) { out[0] = Y(p[0]); out[1] = Y(p[1]); out[2] = Y(p[2]); }
template <typename X, typename Y> DPLL_HD void load33(const X* p, Y (&out)[3][3]) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) DPLL_UNROLL for (int c = 0; c < 3; ++c) out[r][c] = Y(p[3 * r + c]);
}
template <typename X, typename Y> DPLL_HD void store3(const X (&v)[3], Y* p) { p[0] = Y(v[0]); p[1] = Y(v[1]); p[2] = Y(v[2]); }
template <typename X, typename Y> DPLL_HD void store33(const X (&m)[3][3], Y* p) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) DPLL_UNROLL for (int c = 0; c < 3; ++c) p[3 * r + c] = Y(m[r][c]);
}
DPLL_HD int dofs_of(int kind) { return kind == kJointFloating ? 6 : (kind == kJointFixed ? 0 : 1); }

// constants of the description as scalars of type S (plain numbers, or duals with zero derivative)
template <typename S> DPLL_HD S cst(double x) {
  if constexpr (IsDual<S>::value) return S(decltype(S().v)(x));
  else return S(x);
}
template <typename S> struct ValueOf { using type = S; };
template <typename V> struct ValueOf<DualT<V>> { using type = V; };
template <typename S> DPLL_HD typename ValueOf<S>::type value_of(const S& x) {
  if constexpr (IsDual<S>::value) return x.v;
  else return x;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the per-item program
// ---------------------------------------------------------------------------------------------------------------------------
template <typename S, typename SA, class Team> struct Forest {
  const ForestDesc& fd;
  Arena<S, SA>& A;
  DPLL_HD Forest(const ForestDesc& fd_, Arena<S, SA>& A_) : fd(fd_), A(A_) {}

  // ---- parameters ------------------------------------------------------------------------------------------------------------
  // P: the parameter dtype in memory
  template <typename P> DPLL_HD void derive(const P* theta, const P* friction, const P* lengths) {
    for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
      S th[10], io[kIota];
      DPLL_UNROLL for (int i = 0; i < 10; ++i) th[i] = cst<S>(double(theta[10 * b + i]));
      theta_to_iota<S>(th, fd.inertia_mode, io);
      if (fd.rotated & 1) rotate_iota<S>(fd.body_rot[b], io);
      DPLL_UNROLL for (int i = 0; i < kIota; ++i) A.iota[kIota * b + i] = io[i];
    }
    const S mu0 = cst<S>(fabs(double(friction[0])));
    for (int g = Team::rank(); g < A.ng; g += Team::kSize) {
      const S mug = cst<S>(fabs(double(friction[1 + g])));
      A.mu_g[g] = S(2) * mu0 * mug / (mu0 + mug);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) A.habs[3 * g + i] = cst<S>(fabs(double(lengths[kGeoStride * g + i])));
    }
    for (int p = Team::rank(); p < A.np; p += Team::kSize) {
      const S ma = cst<S>(fabs(double(friction[1 + fd.pair_a[p]]))), mb = cst<S>(fabs(double(friction[1 + fd.pair_b[p]])));
      A.mu_p[p] = S(2) * ma * mb / (ma + mb);
    }
    for (int i = Team::rank(); i < A.nv; i += Team::kSize)
      for (int j = 0; j <= i; ++j) { A.tri_i[i * (i + 1) / 2 + j] = (short)i; A.tri_j[i * (i + 1) / 2 + j] = (short)j; }
    Team::sync();
  }

  // ---- kinematics (level by level; SA: double in the float build) -----------------------------------------------------------------
  DPLL_HD void kinematics() {
    for (int lvl = 0; lvl <= fd.max_depth; ++lvl) {
      for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
        if (fd.depth[b] != lvl) continue;
        const int kind = fd.joint_kind[b], qi = fd.q_index[b];
        SA Rb[3][3], ob[3], Rpc[3][3], pj[3], ax[3];
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { ax[i] = cst<SA>(fd.joint_axis[b][i]); pj[i] = cst<SA>(fd.joint_origin[b][i]); }
        DPLL_UNROLL for (int r = 0; r < 3; ++r) DPLL_UNROLL for (int c = 0; c < 3; ++c) Rpc[r][c] = cst<SA>(r == c ? 1.0 : 0.0);
        if (kind == kJointFloating) {
          quat_to_rot(A.q + qi, Rb);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) ob[i] = A.q[qi + 4 + i];
        } else if (kind == kJointFixed) {
          DPLL_UNROLL for (int r = 0; r < 3; ++r) { DPLL_UNROLL for (int c = 0; c < 3; ++c) Rb[r][c] = Rpc[r][c]; ob[r] = pj[r]; }
        } else {
          const int p = fd.parent[b];
          SA Rp[3][3], op[3], t[3];
          load33(A.R + 9 * p, Rp);
          load3(A.o + 3 * p, op);
          if (kind == kJointPrismatic) {
            DPLL_UNROLL for (int i = 0; i < 3; ++i) pj[i] += ax[i] * A.q[qi];
          } else {
            axis_rot(ax, A.q[qi], Rpc);
          }
          mat3_mul(Rp, Rpc, Rb);
          mat3_vec(Rp, pj, t);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) ob[i] = op[i] + t[i];
        }
        SA axw[3];
        mat3_vec(Rb, ax, axw);
        store33(Rb, A.R + 9 * b); store3(ob, A.o + 3 * b); store33(Rpc, A.Rpc + 9 * b); store3(pj, A.pj + 3 * b); store3(axw, A.axw + 3 * b);
      }
      Team::sync();
    }
  }

  // body-frame twists Y_b = S_b y of `count` generalized velocities at once: out[k][b] = (w (3), u (3))
  DPLL_HD void twists(const S* const* ys, int count, S* out) {
    for (int lvl = 0; lvl <= fd.max_depth; ++lvl) {
      for (int e = Team::rank(); e < count * A.nb; e += Team::kSize) {
        const int k = e / A.nb, b = e % A.nb;
        if (fd.depth[b] != lvl) continue;
        const S* y = ys[k];
        S* dst = out + ((size_t)k * A.nb + b) * 6;
        const int kind = fd.joint_kind[b], vi = fd.v_index[b];
        S yw[3], yu[3];
        if (kind == kJointFloating) {
          S Rb[3][3];
          load33(A.R + 9 * b, Rb);
          const S vl[3] = {y[vi + 3], y[vi + 4], y[vi + 5]};
          DPLL_UNROLL for (int i = 0; i < 3; ++i) yw[i] = y[vi + i];
          mat3t_vec(Rb, vl, yu);
        } else if (kind == kJointFixed) {
          DPLL_UNROLL for (int i = 0; i < 3; ++i) { yw[i] = S(0); yu[i] = S(0); }
        } else {
          const S* src = out + ((size_t)k * A.nb + fd.parent[b]) * 6;
          S pw[3] = {src[0], src[1], src[2]}, pu[3] = {src[3], src[4], src[5]}, Rpc[3][3], pj[3], wxp[3], t[3];
          load33(A.Rpc + 9 * b, Rpc);
          load3(A.pj + 3 * b, pj);
          cross(pw, pj, wxp);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) t[i] = pu[i] + wxp[i];
          mat3t_vec(Rpc, t, yu);
          mat3t_vec(Rpc, pw, yw);
          const S rate = y[vi];
          DPLL_UNROLL for (int i = 0; i < 3; ++i) {
            const S sr = cst<S>(fd.joint_axis[b][i]) * rate;
            if (kind == kJointPrismatic) yu[i] += sr;
            else yw[i] += sr;
          }
        }
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { dst[i] = yw[i]; dst[3 + i] = yu[i]; }
      }
      Team::sync();
    }
  }

  // F(q, v) by recursive Newton-Euler at zero generalized acceleration; leaves V (Vw, Vu) and AG = A_bias - G
  DPLL_HD void bias_forces() {
    const S* ys[1] = {A.v};
    twists(ys, 1, A.tw);
    for (int e = Team::rank(); e < 3 * A.nb; e += Team::kSize) { A.Vw[e] = A.tw[(e / 3) * 6 + e % 3]; A.Vu[e] = A.tw[(e / 3) * 6 + 3 + e % 3]; }
    Team::sync();
    // bias accelerations, level by level (kept in AGw / AGu)
    for (int lvl = 0; lvl <= fd.max_depth; ++lvl) {
      for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
        if (fd.depth[b] != lvl) continue;
        const int kind = fd.joint_kind[b];
        S Vw[3], Vu[3], aw[3], au[3];
        load3(A.Vw + 3 * b, Vw); load3(A.Vu + 3 * b, Vu);
        if (kind == kJointFloating) {
          S wxu[3];
          cross(Vw, Vu, wxu);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) { aw[i] = S(0); au[i] = -wxu[i]; }
        } else if (kind == kJointFixed) {
          DPLL_UNROLL for (int i = 0; i < 3; ++i) { aw[i] = S(0); au[i] = S(0); }
        } else {
          const int p = fd.parent[b];
          S pw[3], pu[3], Rpc[3][3], pj[3], wxp[3], t[3], r1[3], r2[3], sr[3], c1[3], c2[3];
          load3(A.AGw + 3 * p, pw); load3(A.AGu + 3 * p, pu);
          load33(A.Rpc + 9 * b, Rpc); load3(A.pj + 3 * b, pj);
          cross(pw, pj, wxp);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) t[i] = pu[i] + wxp[i];
          mat3t_vec(Rpc, t, r2);
          mat3t_vec(Rpc, pw, r1);
          const S rate = A.v[fd.v_index[b]];
          DPLL_UNROLL for (int i = 0; i < 3; ++i) sr[i] = cst<S>(fd.joint_axis[b][i]) * rate;
          cross(Vw, sr, c1);
          cross(Vu, sr, c2);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) {
            aw[i] = r1[i] + (kind == kJointPrismatic ? S(0) : c1[i]);
            au[i] = r2[i] + (kind == kJointPrismatic ? c1[i] : c2[i]);
          }
        }
        store3(aw, A.AGw + 3 * b); store3(au, A.AGu + 3 * b);
      }
      Team::sync();
    }
    // AG = A - G and the bodies' own wrenches
    for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
      S Rb[3][3], gb[3], Vw[3], Vu[3], agw[3], agu[3], io[kIota];
      load33(A.R + 9 * b, Rb); load3(A.Vw + 3 * b, Vw); load3(A.Vu + 3 * b, Vu); load3(A.AGw + 3 * b, agw); load3(A.AGu + 3 * b, agu);
      const S gw[3] = {S(0), S(0), cst<S>(fd.gravity_z)};
      mat3t_vec(Rb, gw, gb);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) agu[i] -= gb[i];
      DPLL_UNROLL for (int i = 0; i < kIota; ++i) io[i] = A.iota[kIota * b + i];
      S n1[3], f1[3], hn[3], hf[3], a1[3], a2[3], a3[3], wn[3], wf[3];
      inertia_apply(io, agw, agu, n1, f1);
      inertia_apply(io, Vw, Vu, hn, hf);
      cross(Vw, hn, a1); cross(Vu, hf, a2); cross(Vw, hf, a3);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) { wn[i] = n1[i] + a1[i] + a2[i]; wf[i] = f1[i] + a3[i]; }
      store3(agu, A.AGu + 3 * b); store3(wn, A.Wn + 3 * b); store3(wf, A.Wf + 3 * b);
    }
    Team::sync();
    // composite wrenches: every parent gathers its children (fixed order), deepest level first
    for (int lvl = fd.max_depth - 1; lvl >= 0; --lvl) {
      for (int p = Team::rank(); p < A.nb; p += Team::kSize) {
        if (fd.depth[p] != lvl) continue;
        S wn[3], wf[3];
        load3(A.Wn + 3 * p, wn); load3(A.Wf + 3 * p, wf);
        for (int c = p + 1; c < A.nb; ++c) {
          if (fd.parent[c] != p) continue;
          S Rpc[3][3], pj[3], cn[3], cf[3], rn[3], rf[3], pxf[3];
          load33(A.Rpc + 9 * c, Rpc); load3(A.pj + 3 * c, pj); load3(A.Wn + 3 * c, cn); load3(A.Wf + 3 * c, cf);
          mat3_vec(Rpc, cn, rn); mat3_vec(Rpc, cf, rf);
          cross(pj, rf, pxf);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) { wn[i] += rn[i] + pxf[i]; wf[i] += rf[i]; }
        }
        store3(wn, A.Wn + 3 * p); store3(wf, A.Wf + 3 * p);
      }
      Team::sync();
    }
    for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
      const int kind = fd.joint_kind[b], vi = fd.v_index[b];
      S wn[3], wf[3];
      load3(A.Wn + 3 * b, wn); load3(A.Wf + 3 * b, wf);
      if (kind == kJointFloating) {
        S Rb[3][3], t[3];
        load33(A.R + 9 * b, Rb);
        mat3_vec(Rb, wf, t);
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { A.F[vi + i] = -wn[i]; A.F[vi + 3 + i] = -t[i]; }
      } else if (kind != kJointFixed) {
        S ax[3];
        DPLL_UNROLL for (int i = 0; i < 3; ++i) ax[i] = cst<S>(fd.joint_axis[b][i]);
        A.F[vi] = (kind == kJointPrismatic ? -dot3(ax, wf) : -dot3(ax, wn)) + A.tau[vi];  // (+ B u)
      }
    }
    Team::sync();
  }

  // motion subspace of velocity i in the frame of its body
  DPLL_HD void dof_subspace(int i, S (&sw)[3], S (&su)[3]) const {
    const int b = fd.dof_body[i], kind = fd.joint_kind[b], local = i - fd.v_index[b];
    DPLL_UNROLL for (int k = 0; k < 3; ++k) { sw[k] = S(0); su[k] = S(0); }
    if (kind == kJointFloating) {
      if (local < 3) {
        DPLL_UNROLL for (int k = 0; k < 3; ++k) sw[k] = (k == local) ? S(1) : S(0);
      } else {  // v_world component: (0, R^T e) in the body frame
        DPLL_UNROLL for (int k = 0; k < 3; ++k) su[k] = S(A.R[9 * b + 3 * (local - 3) + k]);
      }
    } else {
      DPLL_UNROLL for (int k = 0; k < 3; ++k) {
        const S ax = cst<S>(fd.joint_axis[b][k]);
        if (kind == kJointPrismatic) su[k] = ax;
        else sw[k] = ax;
      }
    }
  }

  // composite-rigid-body mass matrix on the 10-vector representation (dpll_core.hpp mass_matrix, any forest)
  DPLL_HD void mass_matrix() {
    for (int e = Team::rank(); e < kIota * A.nb; e += Team::kSize) A.comp[e] = A.iota[e];
    for (int e = Team::rank(); e < A.nv * A.nv; e += Team::kSize) A.M[e] = S(0);
    Team::sync();
    for (int lvl = fd.max_depth - 1; lvl >= 0; --lvl) {
      for (int p = Team::rank(); p < A.nb; p += Team::kSize) {
        if (fd.depth[p] != lvl) continue;
        S acc[kIota];
        DPLL_UNROLL for (int i = 0; i < kIota; ++i) acc[i] = A.comp[kIota * p + i];
        for (int c = p + 1; c < A.nb; ++c) {
          if (fd.parent[c] != p) continue;
          // body c's composite inertia in the parent's frame: rotate, then shift the origin by pj
          S Rpc[3][3], d[3], cc[kIota];
          load33(A.Rpc + 9 * c, Rpc); load3(A.pj + 3 * c, d);
          DPLL_UNROLL for (int i = 0; i < kIota; ++i) cc[i] = A.comp[kIota * c + i];
          const S hc_child[3] = {cc[1], cc[2], cc[3]};
          S hc[3];
          mat3_vec(Rpc, hc_child, hc);
          const S I[3][3] = {{cc[4], cc[7], cc[8]}, {cc[7], cc[5], cc[9]}, {cc[8], cc[9], cc[6]}};
          S RI[3][3], RIRt[3][3];
          mat3_mul(Rpc, I, RI);
          DPLL_UNROLL for (int r = 0; r < 3; ++r)
            DPLL_UNROLL for (int k = 0; k < 3; ++k) RIRt[r][k] = RI[r][0] * Rpc[k][0] + RI[r][1] * Rpc[k][1] + RI[r][2] * Rpc[k][2];
          const S m = cc[0];
          const S dd = dot3(d, d), dh = dot3(d, hc);
          S Ip[3][3];
          DPLL_UNROLL for (int r = 0; r < 3; ++r)
            DPLL_UNROLL for (int k = 0; k < 3; ++k) {
              const S delta = (r == k) ? S(1) : S(0);
              Ip[r][k] = RIRt[r][k] - m * (d[r] * d[k] - dd * delta) - (hc[r] * d[k] + d[r] * hc[k] - S(2) * dh * delta);
            }
          const S add[kIota] = {m, hc[0] + m * d[0], hc[1] + m * d[1], hc[2] + m * d[2], Ip[0][0], Ip[1][1], Ip[2][2], Ip[0][1], Ip[0][2], Ip[1][2]};
          DPLL_UNROLL for (int i = 0; i < kIota; ++i) acc[i] += add[i];
        }
        DPLL_UNROLL for (int i = 0; i < kIota; ++i) A.comp[kIota * p + i] = acc[i];
      }
      Team::sync();
    }
    // column j: the wrench of the composite body under the joint's motion, walked up to the root; velocity j writes the entries
    // (i, j) and (j, i) of the velocities i <= j on its path (every entry has one writer)
    for (int j = Team::rank(); j < A.nv; j += Team::kSize) {
      S sw[3], su[3], n[3], f[3], io[kIota];
      dof_subspace(j, sw, su);
      int a = fd.dof_body[j];
      DPLL_UNROLL for (int i = 0; i < kIota; ++i) io[i] = A.comp[kIota * a + i];
      inertia_apply(io, sw, su, n, f);
      while (true) {
        const int vi = fd.v_index[a], nd = dofs_of(fd.joint_kind[a]);
        for (int i = vi; i < vi + nd && i <= j; ++i) {
          S tw[3], tu[3];
          dof_subspace(i, tw, tu);
          const S val = dot3(tw, n) + dot3(tu, f);
          A.M[i * A.nv + j] = val;
          A.M[j * A.nv + i] = val;
        }
        const int p = fd.parent[a];
        if (p < 0) break;
        S Rpc[3][3], pj[3], rn[3], rf[3], pxf[3];
        load33(A.Rpc + 9 * a, Rpc); load3(A.pj + 3 * a, pj);
        mat3_vec(Rpc, n, rn); mat3_vec(Rpc, f, rf);
        cross(pj, rf, pxf);
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { n[i] = rn[i] + pxf[i]; f[i] = rf[i]; }
        a = p;
      }
    }
    Team::sync();
  }

  // ---- dense symmetric positive definite n x n (row-major, lower triangle used): the wave factors column by column ----------------
  // L overwrites the strict lower triangle of Aio, invd = 1 / diag(L); `fast`: the 1-ulp reciprocal square root of the float build.
  // Then the factor is INVERTED into the strict upper triangle (entry (j, i), j < i, holds (L^-1)[i][j]; its diagonal is invd):
  // column j of L^-1 is one lane's forward substitution, no barrier between the columns -- and every solve with the factor
  // is two matrix-vector products (two barriers) instead of 2 n dependent column steps (2 n barriers), which is what a wave
  // that waits at every barrier pays for.
  DPLL_HD void cholesky(S* Aio, S* invd, int n, bool fast) {
    if constexpr (Team::kLaneRows && !IsDual<S>::value) {  // (a factor made there is solved there: chol_solve tests the same)
      if (n <= Team::kRowsMax) { Team::template factor_rows<S>(Aio, invd, n, fast); return; }
    }
    for (int j = 0; j < n; ++j) {
      Team::sync();
      const S djj = Aio[j * n + j];
      S id;
      if constexpr (IsDual<S>::value) id = S(1) / tsqrt(djj);
      else id = fast ? fast_rsqrt(djj) : S(1) / tsqrt(djj);
      for (int i = j + 1 + Team::rank(); i < n; i += Team::kSize) Aio[i * n + j] = Aio[i * n + j] * id;
      if (Team::rank() == 0) invd[j] = id;
      Team::sync();
      const int m = n - j - 1;
      for (int e = Team::rank(); e < m * (m + 1) / 2; e += Team::kSize) {
        const int i = j + 1 + A.tri_i[e], k = j + 1 + A.tri_j[e];
        Aio[i * n + k] -= Aio[i * n + j] * Aio[k * n + j];
      }
    }
    Team::sync();
    for (int j = Team::rank(); j < n; j += Team::kSize) {
      for (int i = j + 1; i < n; ++i) {
        S acc = Aio[i * n + j] * invd[j];
        DPLL_PIPE for (int p = j + 1; p < i; ++p) acc += Aio[i * n + p] * Aio[j * n + p];
        Aio[j * n + i] = -acc * invd[i];
      }
    }
    Team::sync();
  }
  // x = (L L^T)^-1 b with the inverted factor of `cholesky`; work: n numbers of scratch (x may alias neither b nor work)
  DPLL_HD void chol_solve(const S* L, const S* invd, const S* b, S* x, S* work, int n) {
    if constexpr (Team::kLaneRows && !IsDual<S>::value) {
      if (n <= Team::kRowsMax) { Team::template solve_rows<S>(L, invd, b, x, n); return; }
    }
    for (int i = Team::rank(); i < n; i += Team::kSize) {  // y = L^-1 b
      S acc = invd[i] * b[i];
      DPLL_PIPE for (int j = 0; j < i; ++j) acc += L[j * n + i] * b[j];
      work[i] = acc;
    }
    Team::sync();
    for (int i = Team::rank(); i < n; i += Team::kSize) {  // x = L^-T y
      S acc = invd[i] * work[i];
      DPLL_PIPE for (int k = i + 1; k < n; ++k) acc += L[i * n + k] * work[k];
      x[i] = acc;
    }
    Team::sync();
  }
  DPLL_HD void symv(const S* Mat, const S* x, S* y, int n) {
    for (int i = Team::rank(); i < n; i += Team::kSize) {
      S s = S(0);
      DPLL_PIPE for (int j = 0; j < n; ++j) s += Mat[i * n + j] * x[j];
      y[i] = s;
    }
    Team::sync();
  }
  DPLL_HD S dot(const S* a, const S* b, int n) {
    S s = S(0);
    for (int i = Team::rank(); i < n; i += Team::kSize) s += a[i] * b[i];
    return Team::sum(s);
  }

  // M, its factor, a = M^-1 F at (q, v) of the arena
  DPLL_HD void terms() {
    kinematics();
    DPLL_FSTAMP(1);
    mass_matrix();
    DPLL_FSTAMP(2);
    bias_forces();
    DPLL_FSTAMP(3);
    for (int e = Team::rank(); e < A.nv * A.nv; e += Team::kSize) A.LM[e] = A.M[e];
    Team::sync();
    cholesky(A.LM, A.invdM, A.nv, false);
    chol_solve(A.LM, A.invdM, A.F, A.a, A.tmp, A.nv);
    DPLL_FSTAMP(4);
  }

  // ---- contacts ------------------------------------------------------------------------------------------------------------------
  // rows of the point Jacobian of `pt` (world) rigidly attached to body b, times `sign`, added into Jrow (3 x nv, dense)
  // through the contact frame Fr (rows = axes in the world)
  DPLL_HD void add_point_jacobian(int b, const S (&pt)[3], const S (&Fr)[3][3], S sign, S* Jrow) const {
    int a = b;
    while (true) {
      const int kind = fd.joint_kind[a], vi = fd.v_index[a];
      if (kind == kJointFloating) {
        S Ra[3][3], oa[3], d0[3];
        load33(A.R + 9 * a, Ra); load3(A.o + 3 * a, oa);
        DPLL_UNROLL for (int i = 0; i < 3; ++i) d0[i] = pt[i] - oa[i];
        DPLL_UNROLL for (int c = 0; c < 3; ++c) {
          const S col[3] = {Ra[0][c], Ra[1][c], Ra[2][c]};
          S x[3];
          cross(col, d0, x);
          DPLL_UNROLL for (int r = 0; r < 3; ++r) {
            Jrow[r * A.nv + vi + c] += sign * (Fr[r][0] * x[0] + Fr[r][1] * x[1] + Fr[r][2] * x[2]);
            Jrow[r * A.nv + vi + 3 + c] += sign * Fr[r][c];
          }
        }
      } else if (kind != kJointFixed) {
        S axw[3], oa[3], dj[3], x[3];
        load3(A.axw + 3 * a, axw); load3(A.o + 3 * a, oa);
        DPLL_UNROLL for (int i = 0; i < 3; ++i) dj[i] = pt[i] - oa[i];
        cross(axw, dj, x);
        DPLL_UNROLL for (int r = 0; r < 3; ++r) {
          const S col = kind == kJointPrismatic ? (Fr[r][0] * axw[0] + Fr[r][1] * axw[1] + Fr[r][2] * axw[2])
                                                 : (Fr[r][0] * x[0] + Fr[r][1] * x[1] + Fr[r][2] * x[2]);
          Jrow[r * A.nv + vi] += sign * col;
        }
      }
      a = fd.parent[a];
      if (a < 0) break;
    }
  }
  // world angular velocity of body b under generalized velocity y
  DPLL_HD void world_omega(int b, const S* y, S (&w)[3]) const {
    DPLL_UNROLL for (int i = 0; i < 3; ++i) w[i] = S(0);
    int a = b;
    while (a >= 0) {
      const int kind = fd.joint_kind[a], vi = fd.v_index[a];
      if (kind == kJointFloating) {
        S Ra[3][3], t[3];
        load33(A.R + 9 * a, Ra);
        const S yb[3] = {y[vi], y[vi + 1], y[vi + 2]};
        mat3_vec(Ra, yb, t);
        DPLL_UNROLL for (int i = 0; i < 3; ++i) w[i] += t[i];
      } else if (kind == kJointRevolute) {
        DPLL_UNROLL for (int i = 0; i < 3; ++i) w[i] += S(A.axw[3 * a + i]) * y[vi];
      }
      a = fd.parent[a];
    }
  }
  // geometry g's frame in the world (SA) and its origin's offset in that frame
  DPLL_HD void geometry_frame(int g, SA (&Rg)[3][3], SA (&ob)[3], SA (&gorg)[3]) const {
    const int b = fd.geom_body[g];
    load33(A.R + 9 * b, Rg);
    load3(A.o + 3 * b, ob);
    if (fd.rotated & 2) mat3_mul_const<SA>(Rg, fd.geom_rot[g]);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) gorg[i] = cst<SA>(fd.geom_origin[g][i]);
  }
  // vertex set of geometry g in its own frame (+ a sphere's radius as a margin): the count, and ONE vertex at a time (an array
  // of them indexed by a run-time number would live in scratch memory)
  DPLL_HD int geometry_nverts(int g, SA& margin) const {
    const int kind = fd.geom_kind[g];
    margin = kind == kGeomSphere ? SA(A.habs[3 * g]) : SA(0);
    return kind == kGeomSphere ? 1 : (kind == kGeomPolygon ? fd.geom_nverts[g] : 8);
  }
  template <typename P> DPLL_HD void geometry_vertex(int g, const P* lengths, int u, SA (&v)[3]) const {
    const int kind = fd.geom_kind[g];
    if (kind == kGeomSphere) {
      DPLL_UNROLL for (int i = 0; i < 3; ++i) v[i] = SA(0);
    } else if (kind == kGeomPolygon) {
      DPLL_UNROLL for (int i = 0; i < 3; ++i) v[i] = cst<SA>(double(lengths[kGeoStride * g + 3 * u + i]));
    } else {
      DPLL_UNROLL for (int i = 0; i < 3; ++i) v[i] = (((u >> (2 - i)) & 1) ? SA(1) : SA(-1)) * SA(A.habs[3 * g + i]);
    }
  }
  // directions of the body-body candidates (frame of A), searched by the team together (dpll_core.hpp pair_direction: fcl's role)
  template <typename P> DPLL_HD void pair_directions(const P* lengths) {
    if constexpr (!IsDual<SA>::value) {
      for (int p = 0; p < A.np; ++p) {
        const int ga = fd.pair_a[p], gb = fd.pair_b[p];
        SA RA[3][3], RB[3][3], oA[3], oB[3], gA[3], gB[3], cA[3], cB[3], mA, mB;
        geometry_frame(ga, RA, oA, gA);
        geometry_frame(gb, RB, oB, gB);
        mat3_vec(RA, gA, cA); mat3_vec(RB, gB, cB);
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { cA[i] += oA[i]; cB[i] += oB[i]; }
        const int na = geometry_nverts(ga, mA), nb = geometry_nverts(gb, mB);
        SA (*sa)[3] = reinterpret_cast<SA (*)[3]>(A.setA);
        SA (*sb)[3] = reinterpret_cast<SA (*)[3]>(A.setB);
        Team::sync();
        // (both sets padded to kMaxPolyVerts with repeats of vertex 0: fixed-length loops in the search)
        for (int u = Team::rank(); u < kMaxPolyVerts; u += Team::kSize) {
          SA vu[3];
          geometry_vertex(ga, lengths, u < na ? u : 0, vu);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) sa[u][i] = vu[i];
        }
        for (int u = Team::rank(); u < kMaxPolyVerts; u += Team::kSize) {
          SA vu[3], wv[3], rel[3], out[3];
          geometry_vertex(gb, lengths, u < nb ? u : 0, vu);
          mat3_vec(RB, vu, wv);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) rel[i] = wv[i] + cB[i] - cA[i];
          mat3t_vec(RA, rel, out);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) sb[u][i] = out[i];
        }
        Team::sync();
        SA dA[3];
        pair_direction<SA, typename Team::Lanes, true>(sa, na, fd.geom_kind[ga], sb, nb, fd.geom_kind[gb], dA);
        if (Team::rank() == 0) { DPLL_UNROLL for (int i = 0; i < 3; ++i) A.dirs[3 * p + i] = dA[i]; }
      }
      Team::sync();
    }
  }

  // contact c: geometry, signed distance, dense Jacobian rows (contact frame: t_x, t_y, n)
  template <typename P> DPLL_HD void contact(int c, const P* lengths) {
    ContactRec<S> local;  // (a lite arena keeps no records: signed distance and coefficient go to cphi / cmu, the rest is dropped)
    ContactRec<S>& ct = A.ct ? A.ct[c] : local;
    contact_into(c, lengths, ct);
    A.cphi[c] = ct.phi;
    A.cmu[c] = ct.mu;
  }
  template <typename P> DPLL_HD void contact_into(int c, const P* lengths, ContactRec<S>& ct) {
    S* Jrow = A.J + (size_t)c * 3 * A.nv;
    for (int e = 0; e < 3 * A.nv; ++e) Jrow[e] = S(0);
    const int g = fd.contact_geom[c], slot = fd.contact_slot[c];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { ct.sgn[i] = S(0); ct.drad[i] = S(0); ct.sgn_a[i] = S(0); ct.drad_a[i] = S(0); ct.dir[i] = S(0); }
    ct.vidx = -1; ct.vidx_a = -1; ct.body_a = 0; ct.geom_a = -1;
    if (g >= 0) {  // witness `slot` of geometry g against the ground (geometry.py:511-582)
      const int b = fd.geom_body[g], kind = fd.geom_kind[g];
      SA RgA[3][3], obA[3], gorgA[3];
      geometry_frame(g, RgA, obA, gorgA);
      S Rg[3][3];
      DPLL_UNROLL for (int r = 0; r < 3; ++r) DPLL_UNROLL for (int k = 0; k < 3; ++k) Rg[r][k] = S(RgA[r][k]);
      const S d[3] = {-Rg[2][0], -Rg[2][1], -Rg[2][2]};
      const S habs[3] = {A.habs[3 * g], A.habs[3 * g + 1], A.habs[3 * g + 2]};
      S wit[3];
      if (kind == kGeomSphere) {
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { wit[i] = d[i] * habs[0]; ct.drad[i] = d[i]; }
      } else if (kind == kGeomPolygon) {
        const int nv = fd.geom_nverts[g];
        S best[3] = {S(0), S(0), S(0)};
        for (int u = 0; u < nv; ++u) {
          S vu[3];
          DPLL_UNROLL for (int i = 0; i < 3; ++i) vu[i] = cst<S>(double(lengths[kGeoStride * g + 3 * u + i]));
          const S du = dot3(d, vu);
          int rank = 0;
          for (int o2 = 0; o2 < nv; ++o2) {
            if (o2 == u) continue;
            S vo[3];
            DPLL_UNROLL for (int i = 0; i < 3; ++i) vo[i] = cst<S>(double(lengths[kGeoStride * g + 3 * o2 + i]));
            const S dov = dot3(d, vo);
            rank += (o2 < u ? dov >= du : dov > du) ? 1 : 0;  // ties: lower index first
          }
          if (rank == slot) { ct.vidx = u; DPLL_UNROLL for (int i = 0; i < 3; ++i) best[i] = vu[i]; }
        }
        DPLL_UNROLL for (int i = 0; i < 3; ++i) wit[i] = best[i];
      } else {
        box_corner_signs(d, habs, slot, ct.sgn);
        DPLL_UNROLL for (int i = 0; i < 3; ++i) wit[i] = ct.sgn[i] * habs[i];
      }
      S r_b[3], rho[3], pt[3];
      DPLL_UNROLL for (int i = 0; i < 3; ++i) r_b[i] = S(gorgA[i]) + wit[i];
      mat3_vec(Rg, r_b, rho);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) pt[i] = S(obA[i]) + rho[i];
      SA phiA = obA[2];
      DPLL_UNROLL for (int i = 0; i < 3; ++i) phiA += RgA[2][i] * (gorgA[i] + SA(wit[i]));
      ct.phi = S(phiA);
      ct.mu = A.mu_g[g];
      ct.body = b; ct.geom = g; ct.pair = -1;
      DPLL_UNROLL for (int r = 0; r < 3; ++r) DPLL_UNROLL for (int k = 0; k < 3; ++k) { ct.R[r][k] = Rg[r][k]; ct.F[r][k] = (r == k) ? S(1) : S(0); ct.Ra[r][k] = ct.F[r][k]; }
      add_point_jacobian(b, pt, ct.F, S(1), Jrow);
      return;
    }
    // candidate `slot`: ONE contact along the direction found up front (geometry.py:585-643, multibody_terms.py:464-513)
    const int p = slot, ga = fd.pair_a[p], gb = fd.pair_b[p];
    const int kindA = fd.geom_kind[ga], kindB = fd.geom_kind[gb];
    SA RA[3][3], RB[3][3], oA[3], oB[3], gorgA[3], gorgB[3], marginA, marginB;
    geometry_frame(ga, RA, oA, gorgA);
    geometry_frame(gb, RB, oB, gorgB);
    const int na = geometry_nverts(ga, marginA), nb = geometry_nverts(gb, marginB);
    SA dA[3], dW[3], dB[3], ndW[3];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) dA[i] = A.dirs[3 * p + i];
    mat3_vec(RA, dA, dW);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) ndW[i] = -dW[i];
    mat3t_vec(RB, ndW, dB);
    int ia = 0, ib = 0;
    SA besta = cst<SA>(-3.0e38), bestb = cst<SA>(-3.0e38);
    for (int u = 0; u < na; ++u) {
      SA vu[3];
      geometry_vertex(ga, lengths, u, vu);
      const SA t = dA[0] * vu[0] + dA[1] * vu[1] + dA[2] * vu[2];
      if (t > besta + cst<SA>(kPairTie)) { besta = t; ia = u; }
    }
    for (int u = 0; u < nb; ++u) {
      SA vu[3];
      geometry_vertex(gb, lengths, u, vu);
      const SA t = dB[0] * vu[0] + dB[1] * vu[1] + dB[2] * vu[2];
      if (t > bestb + cst<SA>(kPairTie)) { bestb = t; ib = u; }
    }
    SA witA[3], witB[3], vA[3], vB[3];
    geometry_vertex(ga, lengths, ia, vA);
    geometry_vertex(gb, lengths, ib, vB);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { witA[i] = vA[i] + marginA * dA[i]; witB[i] = vB[i] + marginB * dB[i]; }
    DPLL_UNROLL for (int i = 0; i < 3; ++i) {
      ct.sgn_a[i] = kindA == kGeomBox ? (((ia >> (2 - i)) & 1) ? S(1) : S(-1)) : S(0);
      ct.sgn[i] = kindB == kGeomBox ? (((ib >> (2 - i)) & 1) ? S(1) : S(-1)) : S(0);
      ct.drad_a[i] = kindA == kGeomSphere ? S(dA[i]) : S(0);
      ct.drad[i] = kindB == kGeomSphere ? S(dB[i]) : S(0);
      ct.dir[i] = S(dA[i]);
    }
    ct.vidx_a = kindA == kGeomPolygon ? ia : -1;
    ct.vidx = kindB == kGeomPolygon ? ib : -1;
    SA rA[3], rB[3], ptA[3], ptB[3], wA[3], wB[3];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { rA[i] = gorgA[i] + witA[i]; rB[i] = gorgB[i] + witB[i]; }
    mat3_vec(RA, rA, wA); mat3_vec(RB, rB, wB);
    SA phi = SA(0);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { ptA[i] = oA[i] + wA[i]; ptB[i] = oB[i] + wB[i]; phi += dW[i] * (ptB[i] - ptA[i]); }
    ct.phi = S(phi);
    ct.mu = A.mu_p[p];
    ct.body = fd.geom_body[gb]; ct.geom = gb; ct.body_a = fd.geom_body[ga]; ct.geom_a = ga; ct.pair = p;
    SA FA[3][3];
    frame_from_normal<SA>(dA, FA);
    DPLL_UNROLL for (int k = 0; k < 3; ++k) {
      SA axis[3];
      mat3_vec(RA, FA[k], axis);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) ct.F[k][i] = S(axis[i]);
    }
    DPLL_UNROLL for (int r = 0; r < 3; ++r) DPLL_UNROLL for (int k = 0; k < 3; ++k) { ct.Ra[r][k] = S(RA[r][k]); ct.R[r][k] = S(RB[r][k]); }
    S pa[3], pb[3];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { pa[i] = S(ptA[i]); pb[i] = S(ptB[i]); }
    add_point_jacobian(ct.body, pb, ct.F, S(1), Jrow);
    add_point_jacobian(ct.body_a, pa, ct.F, S(-1), Jrow);
  }
  template <typename P> DPLL_HD void contacts(const P* lengths) {
    pair_directions(lengths);
    DPLL_FSTAMP(5);
    for (int c = Team::rank(); c < A.K; c += Team::kSize) contact(c, lengths);
    Team::sync();
    DPLL_FSTAMP(6);
  }
  // out = J_c y (contact frame: t_x, t_y, n)
  DPLL_HD void jac_apply(int c, const S* y, S (&out)[3]) const {
    const S* Jrow = A.J + (size_t)c * 3 * A.nv;
    DPLL_UNROLL for (int r = 0; r < 3; ++r) {
      S s = S(0);
      DPLL_PIPE for (int i = 0; i < A.nv; ++i) s += Jrow[r * A.nv + i] * y[i];
      out[r] = s;
    }
  }

  // ---- the cone solve (dpll_core.hpp sap_newton on an item of its own: every decision is uniform over the team) ---------------------
  DPLL_HD void project_point(ConePoint<S>& cp) {
    Proj<S> pr;
    lorentz_project(cp.zs, pr);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) cp.g[i] = pr.g[i];
    cp.that[0] = pr.that[0]; cp.that[1] = pr.that[1];
    cp.cp = pr.cp; cp.a = pr.a; cp.b = pr.b;
    cp.polar = pr.polar ? 1 : 0;
  }
  // jtg = sum_c J_c^T D_mu g_c; returns the sum of the normal components
  DPLL_HD S gather_jtg(const ConePoint<S>* pts, S* jtg) {
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) {
      S s = S(0);
      DPLL_PIPE for (int c = 0; c < A.K; ++c) {  // (a contact in the polar region has g = 0: no branch needed)
        const S* Jrow = A.J + (size_t)c * 3 * A.nv;
        const S mu = A.cmu[c];
        s += Jrow[i] * (mu * pts[c].g[0]) + Jrow[A.nv + i] * (mu * pts[c].g[1]) + Jrow[2 * A.nv + i] * pts[c].g[2];
      }
      jtg[i] = s;
    }
    S ns = S(0);
    for (int c = Team::rank(); c < A.K; c += Team::kSize) ns += pts[c].g[2];
    return Team::sum(ns);
  }
  // the state at y + alpha d from the state at y (sap_advance); returns the sum of the normal components
  DPLL_HD S advance(const S* y, const S* My, const ConePoint<S>* cur, S alpha, S* yn, S* Myn, ConePoint<S>* nxt, S* jtgn) {
    const bool move = alpha != S(0);  // (a team that stands still copies its state: no 0 * inf of a direction it no longer needs)
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) {
      yn[i] = move ? y[i] + alpha * A.d[i] : y[i];
      Myn[i] = move ? My[i] + alpha * A.Md[i] : My[i];
    }
    for (int c = Team::rank(); c < A.K; c += Team::kSize) {
      DPLL_UNROLL for (int r = 0; r < 3; ++r) nxt[c].zs[r] = move ? cur[c].zs[r] - alpha * A.jd[3 * c + r] : cur[c].zs[r];
      project_point(nxt[c]);
    }
    Team::sync();
    const S ns = gather_jtg(nxt, jtgn);
    Team::sync();
    return ns;
  }
  DPLL_HD S quadratic(const ConePoint<S>& p, const S* w) const {
    const S a1 = p.that[0] * w[1] - p.that[1] * w[0];
    const S u = p.that[0] * w[0] + p.that[1] * w[1];
    return p.cp * a1 * a1 + p.a * (u * u + w[2] * w[2]) + (p.b + p.b) * u * w[2];
  }
  // argmin_{f in K} 1/2 f^T (J M^-1 J^T + eps) f + qc^T f through its primal in y; leaves y (A.y0), force; returns iterations
  DPLL_HD int solve(S eps, const SolverOpts& opt, int n_stages, S factor) {
    const bool fast = sizeof(S) == 4;
    const S tol2 = S(opt.tol * opt.tol), stol2 = S(opt.stall_tol * opt.stall_tol), ls_tol = S(opt.ls_tol);
    const S stage_tol2 = S(opt.stage_tol * opt.stage_tol);
    const S inv_factor = S(1) / factor;
    const int last_stage = n_stages - 1;
    S eps_c = eps;
    for (int s = 0; s < last_stage; ++s) eps_c *= factor;
    int stage = 0, it_stage = 0, stall = 0, iters = 0;
    S best = S(3.0e38);
    S *y = A.y0, *My = A.My0, *jtg = A.jtg0, *yt = A.y1, *Myt = A.My1, *jtgt = A.jtg1;
    ConePoint<S>*cur = A.p0, *trial = A.p1;
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) { y[i] = S(0); My[i] = S(0); }
    for (int c = Team::rank(); c < A.K; c += Team::kSize) {
      DPLL_UNROLL for (int r = 0; r < 3; ++r) cur[c].zs[r] = -A.ct[c].qc[r];
      project_point(cur[c]);
    }
    Team::sync();
    S nsum = gather_jtg(cur, jtg);
    Team::sync();
    // Several teams may share a wavefront (the device build for small systems: four items per wave, 16 lanes each): every
    // barrier below is then met by all of them, so loops and branches that contain one are decided for the WAVE
    // (Team::wave_any) and an item that has finished -- or does not take a branch -- goes through it with its state held.
    bool active = true;
    for (int it = 0; it < opt.max_iter; ++it) {
      if (!Team::wave_any(active)) break;
      DPLL_FSTAMP(8);
      const S ieps = fast ? fast_rcp(eps_c) : S(1) / eps_c;
      const bool final_stage = stage >= last_stage;
      for (int i = Team::rank(); i < A.nv; i += Team::kSize) A.grad[i] = My[i] - ieps * jtg[i];
      const bool any_force = nsum > S(0);
      // H = M + sum_c J_c^T C_c J_c,  C = D_mu dP D_mu / eps (contacts in the polar region add nothing).  With every contact of
      // the wave's items in the polar region -- airborne items, most of a toss -- H IS M, whose factor is there already.
      S n_act = S(0);
      for (int c = Team::rank(); c < A.K; c += Team::kSize) n_act += cur[c].polar ? S(0) : S(1);
      const bool any_contact = Team::wave_any(Team::sum(n_act) > S(0));
      if (any_contact) {
      for (int c = Team::rank(); c < A.K; c += Team::kSize) {
        const ConePoint<S>& p = cur[c];
        const S tx = p.that[0], ty = p.that[1];
        const S dP[6] = {p.cp * ty * ty + p.a * tx * tx, p.cp * tx * tx + p.a * ty * ty, p.a, (p.a - p.cp) * tx * ty, p.b * tx, p.b * ty};
        const S mu = A.ct[c].mu, m1 = mu * ieps, m2 = mu * m1;
        A.Cc[6 * c + 0] = dP[0] * m2; A.Cc[6 * c + 1] = dP[1] * m2; A.Cc[6 * c + 2] = dP[2] * ieps;
        A.Cc[6 * c + 3] = dP[3] * m2; A.Cc[6 * c + 4] = dP[4] * m1; A.Cc[6 * c + 5] = dP[5] * m1;
      }
      Team::sync();
      for (int e = Team::rank(); e < A.K * A.nv; e += Team::kSize) {
        const int c = e / A.nv, j = e % A.nv;
        const S* Jrow = A.J + (size_t)c * 3 * A.nv;
        const S* C = A.Cc + 6 * c;  // (all zero for a contact in the polar region: cp = a = b = 0)
        const S j0 = Jrow[j], j1 = Jrow[A.nv + j], j2 = Jrow[2 * A.nv + j];
        S* dst = A.CJ + (size_t)c * 3 * A.nv;
        dst[j] = C[0] * j0 + C[3] * j1 + C[4] * j2;
        dst[A.nv + j] = C[3] * j0 + C[1] * j1 + C[5] * j2;
        dst[2 * A.nv + j] = C[4] * j0 + C[5] * j1 + C[2] * j2;
      }
      Team::sync();
      for (int e = Team::rank(); e < A.nv * (A.nv + 1) / 2; e += Team::kSize) {
        const int i = A.tri_i[e], j = A.tri_j[e];
        S h = A.M[i * A.nv + j];
        DPLL_PIPE for (int c = 0; c < A.K; ++c) {  // (C J of a contact in the polar region is written as zeros below)
          const S* Jrow = A.J + (size_t)c * 3 * A.nv;
          const S* CJ = A.CJ + (size_t)c * 3 * A.nv;
          h += Jrow[i] * CJ[j] + Jrow[A.nv + i] * CJ[A.nv + j] + Jrow[2 * A.nv + i] * CJ[2 * A.nv + j];
        }
        A.H[i * A.nv + j] = h;
      }
      DPLL_FSTAMP(9);
      cholesky(A.H, A.invd, A.nv, fast);
      DPLL_FSTAMP(10);
      chol_solve(A.H, A.invd, A.grad, A.d, A.tmp, A.nv);
      } else {
        chol_solve(A.LM, A.invdM, A.grad, A.d, A.tmp, A.nv);
      }
      for (int i = Team::rank(); i < A.nv; i += Team::kSize) A.d[i] = -A.d[i];
      Team::sync();
      DPLL_FSTAMP(11);
      const S dec2 = -dot(A.grad, A.d, A.nv);
      const S ynorm2 = dot(y, My, A.nv);
      const S scale = S(1) + tsqrt(tmax(ynorm2, S(0)));
      const S scale2 = scale * scale;
      const bool converged = !(dec2 > (final_stage ? tol2 : stage_tol2) * scale2);
      const bool improved = dec2 < S(0.25) * best;
      stall = improved ? 0 : stall + 1;
      best = tmin(best, dec2);
      const bool stalled = stall >= 3 && !(dec2 > (final_stage ? stol2 : stage_tol2) * scale2);
      const bool force_free = !(dec2 > S(0)) && !any_force;
      const bool moving = active && dec2 > S(0);
      const S slope_tol = (final_stage ? ls_tol : S(opt.stage_ls_tol)) * dec2;
      const int ls_full = final_stage ? opt.max_ls : opt.stage_max_ls;
      const int ls_cap = (opt.fast_ls > 0 && stall < 2) ? (opt.fast_ls < ls_full ? opt.fast_ls : ls_full) : ls_full;
      // the state at y + d: its slope along d is l'(1)
      for (int c = Team::rank(); c < A.K; c += Team::kSize) {
        S t[3];
        jac_apply(c, A.d, t);
        const S mu = A.ct[c].mu;
        A.jd[3 * c] = mu * t[0]; A.jd[3 * c + 1] = mu * t[1]; A.jd[3 * c + 2] = t[2];
      }
      DPLL_FSTAMP(12);
      symv(A.M, A.d, A.Md, A.nv);
      S alpha = moving ? S(1) : S(0);
      DPLL_FSTAMP(13);
      S nsum_t = advance(y, My, cur, alpha, yt, Myt, trial, jtgt);
      DPLL_FSTAMP(14);
      S f1 = S(0);
      for (int i = Team::rank(); i < A.nv; i += Team::kSize) f1 += (Myt[i] - ieps * jtgt[i]) * A.d[i];
      const S first1 = Team::sum(f1);
      const bool reject = moving && !(tabs(first1) <= slope_tol) && (first1 > S(0) || ls_cap > 1);
      if (Team::wave_any(reject)) {
        S c1 = S(0);
        for (int c = Team::rank(); c < A.K; c += Team::kSize) c1 += quadratic(trial[c], A.jd + 3 * c);
        const S curv1 = Team::sum(c1);
        const S dMd = tmax(dot(A.d, A.Md, A.nv), S(0));
        const S second1 = dMd + ieps * curv1;
        const S newton = S(1) - first1 / second1;
        alpha = reject ? ((newton > S(0) && newton < S(1)) ? newton : S(0.5)) : alpha;
        const bool full = reject && ls_cap > 1;
        if (Team::wave_any(full)) {  // stalled: the derivative-based bracketing search, re-projecting the cone residuals at every probe
          const S yMd = dot(My, A.d, A.nv);
          S lo = S(0), hi = S(-1), a_s = S(1);
          bool searching = full;
          for (int ls = 0; ls < opt.max_ls; ++ls) {
            if (!Team::wave_any(searching)) break;
            S part1 = S(0), part2 = S(0);
            for (int c = Team::rank(); c < A.K; c += Team::kSize) {
              ConePoint<S> pa;
              DPLL_UNROLL for (int r = 0; r < 3; ++r) pa.zs[r] = cur[c].zs[r] - a_s * A.jd[3 * c + r];
              project_point(pa);
              part1 += pa.g[0] * A.jd[3 * c] + pa.g[1] * A.jd[3 * c + 1] + pa.g[2] * A.jd[3 * c + 2];
              part2 += quadratic(pa, A.jd + 3 * c);
            }
            const S first = yMd + a_s * dMd - ieps * Team::sum(part1);
            const S second = dMd + ieps * Team::sum(part2);
            bool ok = tabs(first) <= slope_tol;
            const S lo_n = first < S(0) ? a_s : lo;
            const S hi_n = first >= S(0) ? a_s : hi;
            const S nwt = a_s - first / second;
            const S mid = hi_n < S(0) ? S(2) * a_s : S(0.5) * (lo_n + hi_n);
            const bool bad = !((nwt > lo_n) && (hi_n < S(0) || nwt < hi_n));
            const S nxt = bad ? mid : nwt;
            ok = ok || (hi_n >= S(0) && (hi_n - lo_n) <= S(4) * (sizeof(S) == 4 ? S(1.2e-7) : S(2.3e-16)) * hi_n);
            const bool out = searching && !ok && (ls + 1 >= ls_cap);
            lo = searching ? lo_n : lo;
            hi = searching ? hi_n : hi;
            a_s = (searching && !ok) ? (out ? (lo_n > S(0) ? lo_n : nxt) : nxt) : a_s;
            searching = searching && !ok && !out;
          }
          alpha = full ? a_s : alpha;
        }
        // (a team that kept its step recomputes the state it has: same expressions, same values)
        nsum_t = advance(y, My, cur, alpha, yt, Myt, trial, jtgt);
      }
      DPLL_ITER_HOOK(it, moving, alpha);
      DPLL_FSTAMP(15);
      {  // the trial state is the current one from here on
        S* t;
        t = y; y = yt; yt = t;
        t = My; My = Myt; Myt = t;
        t = jtg; jtg = jtgt; jtgt = t;
        ConePoint<S>* tp = cur; cur = trial; trial = tp;
        nsum = nsum_t;
      }
      iters = active ? it + 1 : iters;
      const bool stage_done = converged || stalled || (!final_stage && it_stage + 1 >= opt.stage_max_iter);
      const bool next_stage = active && !final_stage && stage_done && !force_free;
      const bool ends = (final_stage && stage_done) || force_free;
      active = active && !ends;
      stage = next_stage ? stage + 1 : stage;
      eps_c = next_stage ? eps_c * inv_factor : eps_c;
      it_stage = next_stage ? 0 : it_stage + 1;
      best = next_stage ? S(3.0e38) : best;
      stall = next_stage ? 0 : stall;
    }
    const S ieps = fast ? fast_rcp(eps) : S(1) / eps;
    for (int c = Team::rank(); c < A.K; c += Team::kSize)
      DPLL_UNROLL for (int r = 0; r < 3; ++r) A.force[3 * c + r] = cur[c].g[r] * ieps;
    if (y != A.y0)
      for (int i = Team::rank(); i < A.nv; i += Team::kSize) A.y0[i] = y[i];
    Team::sync();
    return iters;
  }

  // ---- state in / out -----------------------------------------------------------------------------------------------------------
  // B u of the item: `u` = its actuation inputs (fd.n_u numbers) or nullptr (no inputs: zeros, what the reference's sim_step
  // passes, multibody_learnable_system.py:311); data, not a variable of any derivative
  template <typename X> DPLL_HD void load_actuation(const X* u) {
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) A.tau[i] = S(0);
    Team::sync();
    if (u != nullptr) {
      for (int k = Team::rank(); k < fd.n_u; k += Team::kSize) A.tau[fd.v_index[fd.act_body[k]]] = cst<S>(double(u[k]));
      Team::sync();
    }
  }
  template <typename X> DPLL_HD void load_state(const X* x) {
    for (int i = Team::rank(); i < A.nq; i += Team::kSize) A.q[i] = cst<SA>(double(x[i]));
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) A.v[i] = cst<S>(double(x[A.nq + i]));
    Team::sync();
  }

  // ---- ContactNets loss of one transition, forward + adjoint (multibody_learnable_system.py:104-197; dpll_core.hpp loss_item) --------
  // x, xp: the transition (memory dtype X).  Gradient terms are ADDED into `row` (row_width(fd) doubles: the team's partial sums,
  // every entry owned by one lane), scaled by `weight`.  Returns the loss (uniform over the team).
  template <typename X, typename P>
  DPLL_HD S loss(const X* x, const X* xp, const P* lengths, const SolverOpts& opt, S weight, bool want_grad, double* row, int& iters) {
    const S dt = S(fd.dt), eps = S(kLossEps);
    DPLL_FSTAMP(0);
    load_state(xp);  // terms at the NEXT state (quirk Q6)
    terms();
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) {
      A.vp[i] = A.v[i];
      A.dv[i] = A.v[i] - (S(double(x[A.nq + i])) + A.a[i] * dt);
    }
    Team::sync();
    contacts(lengths);
    S pen_part = S(0);
    for (int c = Team::rank(); c < A.K; c += Team::kSize) {
      ContactRec<S>& ct = A.ct[c];
      S jdv[3];
      jac_apply(c, A.dv, jdv);
      jac_apply(c, A.vp, ct.jpv);
      ct.slide[0] = ct.mu * ct.jpv[0];
      ct.slide[1] = ct.mu * ct.jpv[1];
      ct.speed = tsqrt(ct.slide[0] * ct.slide[0] + ct.slide[1] * ct.slide[1]);
      ct.qc[0] = -ct.mu * jdv[0] + dt * ct.slide[0];
      ct.qc[1] = -ct.mu * jdv[1] + dt * ct.slide[1];
      ct.qc[2] = -jdv[2] + tabs(ct.phi) + dt * ct.speed;
      const S neg = tmax(-ct.phi, S(0));
      pen_part += neg * neg;
    }
    const S pen = Team::sum(pen_part);
    Team::sync();
    DPLL_FSTAMP(7);
    const int n_stages = opt.loss_n_stages > 0 ? opt.loss_n_stages : opt.n_stages;
    const S factor = S(opt.loss_n_stages > 0 ? opt.loss_stage_factor : opt.stage_factor);
    iters = solve(eps, opt, n_stages, factor);
    DPLL_FSTAMP(16);
    // invalid-solve mask (multibody_learnable_system.py:186-192)
    bool bad_part = false;
    for (int e = Team::rank(); e < 3 * A.K; e += Team::kSize) bad_part = bad_part || bad_number(A.force[e]) || tabs(A.force[e]) > S(kInvalidForce);
    const bool bad = Team::any(bad_part);
    Team::sync();
    if (bad)
      for (int e = Team::rank(); e < 3 * A.K; e += Team::kSize) A.force[e] = S(0);
    Team::sync();
    // g = J^T D_mu f, w = M^-1 g
    // The loss  1/2 g^T M^-1 g + f . qc + 1/2 dv^T M dv  with  qc = -D_mu J dv + r  is evaluated as  1/2 u^T M u + f . r  with
    // u = M^-1 g - dv: the first form subtracts two numbers of the size of dv^T M dv to get a loss that is often a thousandth
    // of it (float32 lost up to 1.4e-4 of the loss on systems with dozens of contacts); in the second every term is of
    // the loss's own size.  r = (dt slide_x, dt slide_y, |phi| + dt speed) per contact.
    S fr_part = S(0), ff_part = S(0);
    for (int c = Team::rank(); c < A.K; c += Team::kSize) {
      const S* f = A.force + 3 * c;
      const ContactRec<S>& ct = A.ct[c];
      fr_part += dt * (f[0] * ct.slide[0] + f[1] * ct.slide[1]) + f[2] * (tabs(ct.phi) + dt * ct.speed);
      ff_part += f[0] * f[0] + f[1] * f[1] + f[2] * f[2];
    }
    const S fr = Team::sum(fr_part), ff = Team::sum(ff_part);
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) {
      S s = S(0);
      DPLL_PIPE for (int c = 0; c < A.K; ++c) {
        const S* Jrow = A.J + (size_t)c * 3 * A.nv;
        const S* f = A.force + 3 * c;
        const S mu = A.cmu[c];
        s += Jrow[i] * (mu * f[0]) + Jrow[A.nv + i] * (mu * f[1]) + Jrow[2 * A.nv + i] * f[2];
      }
      A.gv[i] = s;
    }
    Team::sync();
    chol_solve(A.LM, A.invdM, A.gv, A.w, A.tmp, A.nv);
    symv(A.M, A.dv, A.Mdv, A.nv);
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) A.u[i] = A.w[i] - A.dv[i];  // (w = 0 after a masked solve: u = -dv)
    Team::sync();
    symv(A.M, A.u, A.tmp, A.nv);
    const S value = bad ? S(0) : S(0.5) * (dot(A.u, A.tmp, A.nv) + eps * ff) + fr + pen;  // (a masked solve: loss 0, as the reference)
    DPLL_FSTAMP(17);
    if (!want_grad) return value;
    // ---- adjoint (dpll_core.hpp loss_item): w = y*, u = w - dv, abar = -dt (M dv - g), b = M^-1 abar -----------------------------
    const S wt = bad ? S(0) : weight;
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) {
      A.w[i] = bad ? S(0) : A.y0[i];
      A.u[i] = A.w[i] - A.dv[i];
      A.abar[i] = -dt * (A.Mdv[i] - A.gv[i]);
    }
    Team::sync();
    chol_solve(A.LM, A.invdM, A.abar, A.bvec, A.tmp, A.nv);
    const S* ys[4] = {A.w, A.dv, A.bvec, A.a};
    twists(ys, 4, A.tw);
    DPLL_FSTAMP(18);
    for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
      S g[kIota];
      DPLL_UNROLL for (int i = 0; i < kIota; ++i) g[i] = S(0);
      auto tw = [&](int k, S (&wv)[3], S (&uv)[3]) {
        const S* src = A.tw + ((size_t)k * A.nb + b) * 6;
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { wv[i] = src[i]; uv[i] = src[3 + i]; }
      };
      S Ww[3], Wu[3], Dw[3], Du[3], Bw[3], Bu[3], Aw[3], Au[3], Vw[3], Vu[3], agw[3], agu[3];
      tw(0, Ww, Wu); tw(1, Dw, Du); tw(2, Bw, Bu); tw(3, Aw, Au);
      load3(A.Vw + 3 * b, Vw); load3(A.Vu + 3 * b, Vu); load3(A.AGw + 3 * b, agw); load3(A.AGu + 3 * b, agu);
      inertia_bilinear_grad<S>(S(-0.5) * wt, Ww, Wu, Ww, Wu, g);
      inertia_bilinear_grad<S>(S(0.5) * wt, Dw, Du, Dw, Du, g);
      S accw[3], accu[3];
      DPLL_UNROLL for (int i = 0; i < 3; ++i) { accw[i] = Aw[i] + agw[i]; accu[i] = Au[i] + agu[i]; }
      inertia_bilinear_grad<S>(-wt, Bw, Bu, accw, accu, g);
      S cw[3], c1[3], c2[3], cu[3];
      cross(Vw, Bw, cw); cross(Vw, Bu, c1); cross(Vu, Bw, c2);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) cu[i] = c1[i] + c2[i];
      inertia_bilinear_grad<S>(wt, cw, cu, Vw, Vu, g);
      DPLL_UNROLL for (int i = 0; i < kIota; ++i) row[1 + kIota * b + i] += double(g[i]);
    }
    // contact part: per contact d/d mu and the witness adjoints, then every geometry entry gathers its contacts in order
    for (int c = Team::rank(); c < A.K; c += Team::kSize) {
      ContactRec<S>& ct = A.ct[c];
      const S* f = A.force + 3 * c;
      const S ftx = f[0], fty = f[1], fn = f[2];
      const S isp = ct.speed > S(0) ? S(1) / ct.speed : S(0);
      const S shx = ct.slide[0] * isp, shy = ct.slide[1] * isp;
      S jpu[3];
      jac_apply(c, A.u, jpu);
      const S bx = dt * (fn * shx + ftx), by = dt * (fn * shy + fty);
      ct.gmu = wt * (ftx * jpu[0] + fty * jpu[1] + bx * ct.jpv[0] + by * ct.jpv[1]);
      const S phibar = fn * (ct.phi > S(0) ? S(1) : (ct.phi < S(0) ? S(-1) : S(0))) - S(2) * tmax(-ct.phi, S(0));
      const S alpha[3] = {ct.mu * ftx, ct.mu * fty, fn}, beta[3] = {ct.mu * bx, ct.mu * by, S(0)};
      witness_adjoint(ct, alpha, A.u, beta, A.vp, phibar, wt);
    }
    Team::sync();
    DPLL_FSTAMP(19);
    gather_geometry_grads(row);
    if (Team::rank() == 0) row[0] += double(weight) * double(value);
    DPLL_FSTAMP(20);
    return value;
  }

  // d/d(witness points) of a1 . (J w1) + a2 . (J w2) + phibar phi (dpll_core.hpp witness_adjoint), times `scale`, left in the record
  DPLL_HD void witness_adjoint(ContactRec<S>& ct, const S (&a1)[3], const S* w1, const S (&a2)[3], const S* w2, S phibar, S scale) {
    S aw1[3], aw2[3], nW[3];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) {
      aw1[i] = a1[0] * ct.F[0][i] + a1[1] * ct.F[1][i] + a1[2] * ct.F[2][i];
      aw2[i] = a2[0] * ct.F[0][i] + a2[1] * ct.F[1][i] + a2[2] * ct.F[2][i];
      nW[i] = ct.F[2][i];
    }
    S o1[3], o2[3], c1[3], c2[3], rho[3], rb[3];
    world_omega(ct.body, w1, o1); world_omega(ct.body, w2, o2);
    cross(aw1, o1, c1); cross(aw2, o2, c2);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) rho[i] = c1[i] + c2[i] + phibar * nW[i];
    mat3t_vec(ct.R, rho, rb);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { ct.rbar[i] = scale * rb[i]; ct.rbar_a[i] = S(0); }
    if (ct.pair >= 0) {
      world_omega(ct.body_a, w1, o1); world_omega(ct.body_a, w2, o2);
      cross(aw1, o1, c1); cross(aw2, o2, c2);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) rho[i] = -(c1[i] + c2[i] + phibar * nW[i]);
      mat3t_vec(ct.Ra, rho, rb);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) ct.rbar_a[i] = scale * rb[i];
    }
  }
  // row += the contacts' d/d mu and witness adjoints: one lane per row entry, contacts visited in order
  DPLL_HD void gather_geometry_grads(double* row) {
    const int mu0 = 1 + kIota * A.nb, mup0 = mu0 + A.ng, len0 = mup0 + A.np;
    for (int g = Team::rank(); g < A.ng; g += Team::kSize) {
      double s = 0.0;
      for (int c = 0; c < A.K; ++c)
        if (A.ct[c].pair < 0 && A.ct[c].geom == g) s += double(A.ct[c].gmu);
      row[mu0 + g] += s;
    }
    for (int p = Team::rank(); p < A.np; p += Team::kSize) {
      double s = 0.0;
      for (int c = 0; c < A.K; ++c)
        if (A.ct[c].pair == p) s += double(A.ct[c].gmu);
      row[mup0 + p] += s;
    }
    for (int e = Team::rank(); e < kGeoStride * A.ng; e += Team::kSize) {
      const int g = e / kGeoStride, k = e % kGeoStride, kind = fd.geom_kind[g];
      const int used = kind == kGeomBox ? 3 : (kind == kGeomSphere ? 1 : 3 * fd.geom_nverts[g]);
      if (k >= used) continue;
      double s = 0.0;
      for (int c = 0; c < A.K; ++c) {
        const ContactRec<S>& ct = A.ct[c];
        if (ct.geom == g) s += double(witness_share(kind, k, ct.sgn, ct.drad, ct.vidx, ct.rbar));
        if (ct.pair >= 0 && ct.geom_a == g) s += double(witness_share(kind, k, ct.sgn_a, ct.drad_a, ct.vidx_a, ct.rbar_a));
      }
      row[len0 + e] += s;
    }
  }
  DPLL_HD S witness_share(int kind, int k, const S (&sgn)[3], const S (&drad)[3], int vidx, const S (&rbar)[3]) const {
    if (kind == kGeomBox) return sgn[k] * rbar[k];
    if (kind == kGeomSphere) return drad[0] * rbar[0] + drad[1] * rbar[1] + drad[2] * rbar[2];
    return (vidx == k / 3) ? rbar[k % 3] : S(0);
  }

  // ---- one simulation step (forward_dynamics + the Lie-group Euler update; dpll_core.hpp step_item) --------------------------------
  // leaves v+ in A.w and y* in A.y0; x_next (memory dtype X) may be null
  template <typename X, typename P>
  DPLL_HD int step(const X* x, const P* lengths, const SolverOpts& opt, X* x_next) {
    const S dt = S(fd.dt), eps = S(kDynamicsEps), idt = S(1) / dt;
    load_state(x);
    terms();
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) A.vp[i] = A.v[i] + dt * A.a[i];  // v-
    Team::sync();
    contacts(lengths);
    for (int c = Team::rank(); c < A.K; c += Team::kSize) {
      ContactRec<S>& ct = A.ct[c];
      S jv[3];
      jac_apply(c, A.vp, jv);
      ct.qc[0] = ct.mu * jv[0];
      ct.qc[1] = ct.mu * jv[1];
      ct.qc[2] = jv[2] + ct.phi * idt;
    }
    Team::sync();
    const int iters = solve(eps, opt, opt.n_stages, S(opt.stage_factor));
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) A.w[i] = A.vp[i] + A.y0[i];  // v+ = v- + y*
    Team::sync();
    if (x_next) {
      for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
        const int kind = fd.joint_kind[b], qi = fd.q_index[b], vi = fd.v_index[b];
        if (kind == kJointFloating) {
          S qq[4], out[4];
          DPLL_UNROLL for (int i = 0; i < 4; ++i) qq[i] = S(double(x[qi + i]));
          const S r[3] = {A.w[vi] * dt, A.w[vi + 1] * dt, A.w[vi + 2] * dt};
          quat_exp_mul<S>(qq, r, out);
          DPLL_UNROLL for (int i = 0; i < 4; ++i) x_next[qi + i] = X(out[i]);
          DPLL_UNROLL for (int i = 0; i < 3; ++i) x_next[qi + 4 + i] = X(S(double(x[qi + 4 + i])) + A.w[vi + 3 + i] * dt);
        } else if (kind != kJointFixed) {
          x_next[qi] = X(S(double(x[qi])) + A.w[vi] * dt);
        }
      }
      for (int i = Team::rank(); i < A.nv; i += Team::kSize) x_next[A.nq + i] = X(A.w[i]);
    }
    Team::sync();
    return iters;
  }
};

// ---------------------------------------------------------------------------------------------------------------------------
// Backward of one simulation step (dpll_core.hpp step_item_backward + step_state_adjoint): the gradient of
// sum(xbar_next . x_next) with respect to the parameters (added into `row`, iota space) and, when `xbar` is given, to the
// state -- implicit differentiation of the cone solve's stationarity condition, the state part by forward-mode duals through
// the same program (one pass per state component on a second arena).  Double arithmetic whatever the storage dtype.
// ---------------------------------------------------------------------------------------------------------------------------
template <class Team> struct ForestBackward {
  using D = double;
  using Du = DualT<double>;
  const ForestDesc& fd;
  Arena<D, D>& A;
  Arena<Du, Du>& B;
  DPLL_HD ForestBackward(const ForestDesc& fd_, Arena<D, D>& A_, Arena<Du, Du>& B_) : fd(fd_), A(A_), B(B_) {}

  // sv, lam live in A.tmp2 / A.gv after this call; returns nothing (row / xbar are the outputs)
  template <typename X, typename P>
  DPLL_HD void run(const X* x, const X* xbar_next, const P* theta, const P* friction, const P* lengths, const SolverOpts& opt, double* row,
                   X* xbar) {
    Forest<D, D, Team> prog(fd, A);
    const D dt = fd.dt, eps = kDynamicsEps, ieps = 1.0 / eps, idt = 1.0 / dt;
    prog.step(x, lengths, opt, (X*)nullptr);  // leaves v- (A.vp), y* (A.y0), v+ (A.w), the contacts, M and its factor, a, V, AG
    D* sv = A.tmp2;
    D* lam = A.gv;
    // seed: d/d v+ plus the pull-back of d/d q+ through q+ = q (+) v+ dt
    for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
      const int kind = fd.joint_kind[b], qi = fd.q_index[b], vi = fd.v_index[b];
      if (kind == kJointFloating) {
        D qq[4], ob[4], rbar[3];
        DPLL_UNROLL for (int i = 0; i < 4; ++i) { qq[i] = D(x[qi + i]); ob[i] = D(xbar_next[qi + i]); }
        const D r[3] = {A.w[vi] * dt, A.w[vi + 1] * dt, A.w[vi + 2] * dt};
        quat_exp_mul_adjoint<D>(qq, r, ob, rbar);
        DPLL_UNROLL for (int i = 0; i < 3; ++i) {
          sv[vi + i] = D(xbar_next[A.nq + vi + i]) + dt * rbar[i];
          sv[vi + 3 + i] = D(xbar_next[A.nq + vi + 3 + i]) + dt * D(xbar_next[qi + 4 + i]);
        }
      } else if (kind != kJointFixed) {
        sv[vi] = D(xbar_next[A.nq + vi]) + dt * D(xbar_next[qi]);
      }
    }
    // H lambda = s at the solution
    for (int c = Team::rank(); c < A.K; c += Team::kSize) {
      const ContactRec<D>& ct = A.ct[c];
      D jy[3];
      prog.jac_apply(c, A.y0, jy);
      ConePoint<D>& cp = A.p0[c];
      cp.zs[0] = -(ct.mu * jy[0] + ct.qc[0]) * ieps;
      cp.zs[1] = -(ct.mu * jy[1] + ct.qc[1]) * ieps;
      cp.zs[2] = -(jy[2] + ct.qc[2]) * ieps;
      prog.project_point(cp);
      const D tx = cp.that[0], ty = cp.that[1];
      const D dP[6] = {cp.cp * ty * ty + cp.a * tx * tx, cp.cp * tx * tx + cp.a * ty * ty, cp.a, (cp.a - cp.cp) * tx * ty, cp.b * tx, cp.b * ty};
      const D m1 = ct.mu * ieps, m2 = ct.mu * m1;
      D* C = A.Cc + 6 * c;
      C[0] = dP[0] * m2; C[1] = dP[1] * m2; C[2] = dP[2] * ieps; C[3] = dP[3] * m2; C[4] = dP[4] * m1; C[5] = dP[5] * m1;
      // (the projection's own Jacobian entries, for kappa below)
      D* keep = A.jd + 3 * c;  // not enough room for six: kappa is formed from C and the unscaled J lambda instead
      (void)keep;
    }
    Team::sync();
    for (int e = Team::rank(); e < A.K * A.nv; e += Team::kSize) {
      const int c = e / A.nv, j = e % A.nv;
      const D* Jrow = A.J + (size_t)c * 3 * A.nv;
      const D* C = A.Cc + 6 * c;
      const D j0 = Jrow[j], j1 = Jrow[A.nv + j], j2 = Jrow[2 * A.nv + j];
      D* dst = A.CJ + (size_t)c * 3 * A.nv;
      dst[j] = C[0] * j0 + C[3] * j1 + C[4] * j2;
      dst[A.nv + j] = C[3] * j0 + C[1] * j1 + C[5] * j2;
      dst[2 * A.nv + j] = C[4] * j0 + C[5] * j1 + C[2] * j2;
    }
    Team::sync();
    for (int e = Team::rank(); e < A.nv * (A.nv + 1) / 2; e += Team::kSize) {
      const int i = A.tri_i[e], j = A.tri_j[e];
      D h = A.M[i * A.nv + j];
      for (int c = 0; c < A.K; ++c) {
        const D* Jrow = A.J + (size_t)c * 3 * A.nv;
        const D* CJ = A.CJ + (size_t)c * 3 * A.nv;
        h += Jrow[i] * CJ[j] + Jrow[A.nv + i] * CJ[A.nv + j] + Jrow[2 * A.nv + i] * CJ[2 * A.nv + j];
      }
      A.H[i * A.nv + j] = h;
    }
    prog.cholesky(A.H, A.invd, A.nv, false);
    prog.chol_solve(A.H, A.invd, sv, lam, A.tmp, A.nv);
    if (xbar) state_adjoint(x, xbar_next, theta, friction, lengths, sv, lam, xbar);
    // per-contact pieces: kappa_c = C_c (J_c lambda) (C carries D_mu and 1 / eps), friction and witness gradients
    for (int c = Team::rank(); c < A.K; c += Team::kSize) {
      ContactRec<D>& ct = A.ct[c];
      D pl[3], pv[3];
      prog.jac_apply(c, lam, pl);
      prog.jac_apply(c, A.w, pv);
      const D* C = A.Cc + 6 * c;
      // ak = D_mu kappa = C (J lambda) with C = D_mu dP D_mu / eps
      const D ak[3] = {C[0] * pl[0] + C[3] * pl[1] + C[4] * pl[2], C[3] * pl[0] + C[1] * pl[1] + C[5] * pl[2],
                       C[4] * pl[0] + C[5] * pl[1] + C[2] * pl[2]};
      const D imu = ct.mu != 0.0 ? 1.0 / ct.mu : 0.0;
      const D kap[3] = {ak[0] * imu, ak[1] * imu, ak[2]};
      A.jd[3 * c] = ak[0]; A.jd[3 * c + 1] = ak[1]; A.jd[3 * c + 2] = ak[2];
      const D* g = A.p0[c].g;
      ct.gmu = g[0] * pl[0] + g[1] * pl[1] - kap[0] * pv[0] - kap[1] * pv[1];
      const D ag[3] = {ct.mu * g[0], ct.mu * g[1], g[2]}, nak[3] = {-ak[0], -ak[1], -ak[2]};
      prog.witness_adjoint(ct, ag, lam, nak, A.w, -kap[2] * idt, 1.0);
    }
    Team::sync();
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) {
      D s = 0.0;
      for (int c = 0; c < A.K; ++c) {
        const D* Jrow = A.J + (size_t)c * 3 * A.nv;
        s += Jrow[i] * A.jd[3 * c] + Jrow[A.nv + i] * A.jd[3 * c + 1] + Jrow[2 * A.nv + i] * A.jd[3 * c + 2];
      }
      A.abar[i] = dt * (sv[i] - s);
    }
    Team::sync();
    prog.chol_solve(A.LM, A.invdM, A.abar, A.bvec, A.tmp, A.nv);
    const D* ys[4] = {lam, A.y0, A.bvec, A.a};
    prog.twists(ys, 4, A.tw);
    for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
      D g[kIota];
      DPLL_UNROLL for (int i = 0; i < kIota; ++i) g[i] = 0.0;
      auto tw = [&](int k, D (&wv)[3], D (&uv)[3]) {
        const D* src = A.tw + ((size_t)k * A.nb + b) * 6;
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { wv[i] = src[i]; uv[i] = src[3 + i]; }
      };
      D Lw[3], Lu[3], Yw[3], Yu[3], Bw[3], Bu[3], Aw[3], Au[3], Vw[3], Vu[3], agw[3], agu[3];
      tw(0, Lw, Lu); tw(1, Yw, Yu); tw(2, Bw, Bu); tw(3, Aw, Au);
      load3(A.Vw + 3 * b, Vw); load3(A.Vu + 3 * b, Vu); load3(A.AGw + 3 * b, agw); load3(A.AGu + 3 * b, agu);
      inertia_bilinear_grad<D>(-1.0, Lw, Lu, Yw, Yu, g);
      D accw[3], accu[3];
      DPLL_UNROLL for (int i = 0; i < 3; ++i) { accw[i] = Aw[i] + agw[i]; accu[i] = Au[i] + agu[i]; }
      inertia_bilinear_grad<D>(-1.0, Bw, Bu, accw, accu, g);
      D cw[3], c1[3], c2[3], cu[3];
      cross(Vw, Bw, cw); cross(Vw, Bu, c1); cross(Vu, Bw, c2);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) cu[i] = c1[i] + c2[i];
      inertia_bilinear_grad<D>(1.0, cw, cu, Vw, Vu, g);
      DPLL_UNROLL for (int i = 0; i < kIota; ++i) row[1 + kIota * b + i] += g[i];
    }
    prog.gather_geometry_grads(row);
    Team::sync();
  }

  // d(total)/dx with y*, s and lambda held fixed: the partial derivative of
  //   Phi(x) = xbar+_q . q+(q, v+ fixed) + s . v-(x) - lambda . G(x, y*),   G = M(q) y* - sum_c J_c(q)^T D_mu P_K(z_c(x, y*)),
  // one forward-mode pass of (terms + contact geometry) per state component on the dual arena
  template <typename X, typename P>
  DPLL_HD void state_adjoint(const X* x, const X* xbar_next, const P* theta, const P* friction, const P* lengths, const D* sv, const D* lam,
                             X* xbar) {
    Forest<Du, Du, Team> dual(fd, B);
    dual.derive(theta, friction, lengths);
    for (int i = Team::rank(); i < A.nv; i += Team::kSize) B.tau[i] = Du(A.tau[i], 0.0);  // (B u of the item: data)
    Team::sync();
    for (int e = Team::rank(); e < 3 * A.np; e += Team::kSize) B.dirs[e] = Du(A.dirs[e]);  // the primal pass's directions: constants
    Team::sync();
    const D dt = fd.dt, idt = 1.0 / dt, mieps = -1.0 / kDynamicsEps;
    const int nx = A.nq + A.nv;
    for (int k = 0; k < nx; ++k) {
      for (int i = Team::rank(); i < A.nq; i += Team::kSize) B.q[i] = Du(D(x[i]), i == k ? 1.0 : 0.0);
      for (int i = Team::rank(); i < A.nv; i += Team::kSize) B.v[i] = Du(D(x[A.nq + i]), A.nq + i == k ? 1.0 : 0.0);
      Team::sync();
      dual.terms();
      for (int i = Team::rank(); i < A.nv; i += Team::kSize) B.vp[i] = B.v[i] + Du(dt) * B.a[i];  // v-
      Team::sync();
      dual.contacts(lengths);
      D part = 0.0;
      // q+ with the rotation vector v+ dt held fixed, s . v-, -lambda . M y*
      for (int b = Team::rank(); b < A.nb; b += Team::kSize) {
        const int kind = fd.joint_kind[b], qi = fd.q_index[b], vi = fd.v_index[b];
        if (kind == kJointFloating) {
          const Du r[3] = {Du(A.w[vi] * dt), Du(A.w[vi + 1] * dt), Du(A.w[vi + 2] * dt)};
          Du qq[4], qn[4];
          DPLL_UNROLL for (int i = 0; i < 4; ++i) qq[i] = B.q[qi + i];
          quat_exp_mul<Du>(qq, r, qn);
          DPLL_UNROLL for (int i = 0; i < 4; ++i) part += D(xbar_next[qi + i]) * qn[i].d;
          DPLL_UNROLL for (int i = 0; i < 3; ++i) part += D(xbar_next[qi + 4 + i]) * B.q[qi + 4 + i].d;
        } else if (kind != kJointFixed) {
          part += D(xbar_next[qi]) * B.q[qi].d;
        }
      }
      for (int i = Team::rank(); i < A.nv; i += Team::kSize) {
        Du my = Du(0.0);
        for (int j = 0; j < A.nv; ++j) my += B.M[i * A.nv + j] * Du(A.y0[j]);
        part += sv[i] * B.vp[i].d - lam[i] * my.d;
      }
      // + sum_c (J_c lambda) . D_mu P_K(z_c)
      for (int c = Team::rank(); c < A.K; c += Team::kSize) {
        const struct { Du mu, phi; } ct = {B.cmu[c], B.cphi[c]};
        const Du* Jrow = B.J + (size_t)c * 3 * A.nv;
        Du jy[3], jv[3], jl[3];
        DPLL_UNROLL for (int r = 0; r < 3; ++r) {
          Du a = Du(0.0), b2 = Du(0.0), c2 = Du(0.0);
          for (int i = 0; i < A.nv; ++i) {
            a += Jrow[r * A.nv + i] * Du(A.y0[i]);
            b2 += Jrow[r * A.nv + i] * B.vp[i];
            c2 += Jrow[r * A.nv + i] * Du(lam[i]);
          }
          jy[r] = a; jv[r] = b2; jl[r] = c2;
        }
        const Du z[3] = {(ct.mu * jy[0] + ct.mu * jv[0]) * Du(mieps), (ct.mu * jy[1] + ct.mu * jv[1]) * Du(mieps),
                         (jy[2] + jv[2] + ct.phi * Du(idt)) * Du(mieps)};
        const D zv[3] = {z[0].v, z[1].v, z[2].v};
        Proj<D> pr;
        lorentz_project(zv, pr);
        D dP[6];
        proj_jacobian(pr, dP);
        const Du f[3] = {Du(pr.g[0], dP[0] * z[0].d + dP[3] * z[1].d + dP[4] * z[2].d),
                         Du(pr.g[1], dP[3] * z[0].d + dP[1] * z[1].d + dP[5] * z[2].d),
                         Du(pr.g[2], dP[4] * z[0].d + dP[5] * z[1].d + dP[2] * z[2].d)};
        const Du phic = ct.mu * (f[0] * jl[0] + f[1] * jl[1]) + f[2] * jl[2];
        part += phic.d;
      }
      const D total = Team::sum(part);
      if (Team::rank() == 0) xbar[k] = X(total);
      Team::sync();
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------------
// chain from the batch-summed row (iota space) to learnable parameter k of [theta | friction | lengths] (double; a handful of
// flops per parameter, done by one thread per parameter in the finalize kernel)
// ---------------------------------------------------------------------------------------------------------------------------
template <typename P>
DPLL_HD double chain_param(const ForestDesc& fd, const P* theta, const P* friction, const P* lengths, const double* row, int k) {
  const int nb = fd.n_bodies, ng = fd.n_geoms, np = fd.n_pairs;
  const int mu0 = 1 + kIota * nb, mup0 = mu0 + ng, len0 = mup0 + np;
  if (k < 10 * nb) {
    const int b = k / 10;
    double th[10];
    for (int i = 0; i < 10; ++i) th[i] = double(theta[10 * b + i]);
    return theta_grad_component(fd.inertia_mode, th, row + 1 + kIota * b, k % 10, (fd.rotated & 1) ? &fd.body_rot[b] : nullptr);
  }
  if (k < 10 * nb + 1 + ng) {
    const int fk = k - 10 * nb;
    // d (2 ma mb / (ma + mb)) / d friction_fk for the coefficient that combines entries ia and ib, mu = |friction|
    auto factor = [&](int ia, int ib) {
      const double ma = fabs(double(friction[ia])), mb = fabs(double(friction[ib]));
      const double den = (ma + mb) * (ma + mb);
      double fac = 0.0;
      if (fk == ia) fac += 2.0 * mb * mb / den;
      if (fk == ib) fac += 2.0 * ma * ma / den;
      return fac;
    };
    double s = 0.0;
    for (int g = 0; g < ng; ++g) s += row[mu0 + g] * factor(0, 1 + g);
    for (int p = 0; p < np; ++p) s += row[mup0 + p] * factor(1 + fd.pair_a[p], 1 + fd.pair_b[p]);
    const double pk = double(friction[fk]);
    return s * (pk > 0.0 ? 1.0 : (pk < 0.0 ? -1.0 : 0.0));
  }
  const int e = k - (10 * nb + 1 + ng), g = e / kGeoStride;
  if (fd.geom_kind[g] == kGeomPolygon) return row[len0 + e];  // vertices are signed parameters
  const double pl = double(lengths[e]);
  return row[len0 + e] * (pl > 0.0 ? 1.0 : (pl < 0.0 ? -1.0 : 0.0));
}

// which entries of [theta | friction | lengths (n_geoms, 24)] belong to a parameter of the model (the tail of a geometry's block is
// padding, which an optimizer must leave alone)
DPLL_HD bool param_is_real(const ForestDesc& fd, int k) {
  const int head = 10 * fd.n_bodies + 1 + fd.n_geoms;
  if (k < head) return true;
  const int g = (k - head) / kGeoStride, e = (k - head) % kGeoStride, kind = fd.geom_kind[g];
  return kind == kGeomBox ? e < 3 : (kind == kGeomSphere ? e < 1 : e < 3 * fd.geom_nverts[g]);
}

}  // namespace dpll_forest
