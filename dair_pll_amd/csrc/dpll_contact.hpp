// dpll_contact.hpp -- learnable parameters as the item code sees them (Derived), MultibodyTerms of one item (compute_terms), contact geometry incl. body-body candidates
// (part of the per-item math of the contact-dynamics hot path: see dpll_core.hpp for the overview and the reference citations)
#pragma once

#include "dpll_terms.hpp"

namespace dpll {
// ---------------------------------------------------------------------------------------------
// learnable parameters as the item code sees them
// ---------------------------------------------------------------------------------------------
// NG = number of collision geometries (the two fast builds: one per body)
template <typename T, int NJ, int NG = NJ + 1> struct Derived {
  static constexpr int NB = NJ + 1;
  T iota[NB][kIota];
  T mu[NG];       // pair coefficient ground-vs-geometry g: 2 mu_0 mu_g / (mu_0 + mu_g), mu = |friction_params| (multibody_terms.py:321-324, :471)
  T habs[NG][3];  // |length_params| of a box (geometry.py:393-403); [0] = |length_param|, the radius of a sphere (:415-456)
  T mu_pair[kMaxPairs];  // general build: pair coefficient of the two geometries of body-body candidate p
  const T* geo;   // the raw geometry parameter blocks (stride MD::kGeoStride): a Polygon's vertices are read from here
  // general build, actuated models: the ITEM's generalized joint forces B u (multibody_terms.py:142-146) -- not a parameter, but
  // it rides here because every item function already takes the block; zero unless a kernel fills it per item (load_actuation)
  T tau[NJ > 0 ? NJ : 1];
};

// the item's actuation: tau[j] = sum of the inputs u[k] whose actuator drives joint j + 1 (u: the item's row, or nullptr)
template <typename T, int NJ, int NG, class MD>
DPLL_HD void load_actuation(const MD& md, const T* u, Derived<T, NJ, NG>& dp) {
  DPLL_UNROLL for (int j = 0; j < (NJ > 0 ? NJ : 1); ++j) dp.tau[j] = T(0);
  if constexpr (MD::kGeneral && NJ > 0) {
    if (u) {
      DPLL_UNROLL for (int k = 0; k < NJ; ++k)
        DPLL_UNROLL for (int j = 0; j < NJ; ++j) dp.tau[j] += (k < md.n_u && md.act_joint[k] == j) ? u[k] : T(0);
    }
  }
}

template <typename T, int NJ, int NG, class MD>
DPLL_HD void derive_params(const MD& md, const T* theta, const T* friction, const T* lengths, Derived<T, NJ, NG>& dp) {
  const T mu0 = tabs(friction[0]);
  DPLL_UNROLL for (int b = 0; b <= NJ; ++b) {
    T th[10];
    DPLL_UNROLL for (int i = 0; i < 10; ++i) th[i] = theta[10 * b + i];
    // (mode 2, rows that are composites of welded links, is the general / forest builds': masked out of the specialised ones so
    // that their code is what it was)
    theta_to_iota<T>(th, MD::kGeneral ? md.inertia_mode : (md.inertia_mode & 1), dp.iota[b]);
    if constexpr (MD::kGeneral) {
      if (DPLL_ROTATED(md) & 1) rotate_iota<T>(md.body_rot[b], dp.iota[b]);
    }
  }
  DPLL_UNROLL for (int g = 0; g < NG; ++g) {
    const T mug = tabs(friction[1 + g]);
    dp.mu[g] = T(2) * mu0 * mug / (mu0 + mug);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) dp.habs[g][i] = lengths ? tabs(lengths[MD::kGeoStride * g + i]) : T(0);
  }
  dp.geo = lengths;
  DPLL_UNROLL for (int j = 0; j < (NJ > 0 ? NJ : 1); ++j) dp.tau[j] = T(0);
  if constexpr (MD::kGeneral) {
    // pair coefficient of the two geometries of a body-body candidate (the group behind the geometries has none of its own)
    DPLL_UNROLL for (int p = 0; p < kMaxPairs; ++p) {
      const bool on = p < md.n_pairs;
      const T ma = tabs(friction[1 + (on ? md.pair_a[p] : 0)]), mb = tabs(friction[1 + (on ? md.pair_b[p] : 0)]);
      dp.mu_pair[p] = T(2) * ma * mb / (ma + mb);
    }
    if constexpr (NG > kMaxGeoms) dp.mu[NG > kMaxGeoms ? kMaxGeoms : 0] = T(1);
  }
}

// per-item terms shared by the loss and the dynamics
template <typename T, int NJ> struct Terms {
  static constexpr int NB = NJ + 1, NV = 6 + NJ;
  Kin<T, NJ> kin;
  T M[NV][NV];
  T LM[NV][NV], invdM[NV];
  T a[NV];  // M^-1 F
  T Vw[NB][3], Vu[NB][3], AGw[NB][3], AGu[NB][3];
};

// What an item's lanes all hold alike -- the rigid-body terms and the kinematics in the accumulation type: 350 numbers for
// a three-joint tree.  The specialised builds keep it in registers; the general build keeps ONE copy per item in LDS
// (`Lanes::item_store`, csrc/dpll_common.hpp): replicated in the registers of a lone wave it left the double-precision kernels
// living on kilobytes of scratch spills -- the regime in which the compiler produced kernels whose results depended on
// unrelated code.  Every lane of the group writes the same values to the same addresses and reads back
// what it wrote itself, so no lane depends on another lane's store.
template <typename T, typename TA, int NJ> struct ItemStore {
  Terms<T, NJ> t;
  Kin<TA, NJ> kinA;
};

template <typename T, typename TA, int NJ> DPLL_HD void convert_kin(const Kin<TA, NJ>& a, Kin<T, NJ>& k) {
  DPLL_UNROLL for (int b = 0; b <= NJ; ++b)
    DPLL_UNROLL for (int r = 0; r < 3; ++r) {
      DPLL_UNROLL for (int c = 0; c < 3; ++c) { k.R[b][r][c] = T(a.R[b][r][c]); k.Rpc[b][r][c] = T(a.Rpc[b][r][c]); }
      k.o[b][r] = T(a.o[b][r]); k.pj[b][r] = T(a.pj[b][r]); k.ax[b][r] = T(a.ax[b][r]); k.axw[b][r] = T(a.axw[b][r]);
    }
  DPLL_UNROLL for (int b = 0; b <= NJ; ++b) {
    k.par[b] = a.par[b];
    k.pris[b] = a.pris[b];
    DPLL_UNROLL for (int j = 0; j <= NJ; ++j) k.anc[b][j] = a.anc[b][j];
  }
}

// Kinematics run in the accumulation type TA (double in the float kernels): the signed distance
// phi = o_z + (R r)_z is a cancellation of O(0.1) terms down to O(1e-4) that the dynamics then divides by dt,
// and evaluating it in float perturbs the next velocity by up to 2e-4 in weakly constrained directions.
template <typename T, typename TA, int NJ, int NG, class MD>
DPLL_HD void compute_terms(const MD& md, const Derived<T, NJ, NG>& dp, const T* q, const T* v, Terms<T, NJ>& t,
                           Kin<TA, NJ>& kinA) {
  constexpr int NV = 6 + NJ;
  TA qA[7 + NJ];
  DPLL_UNROLL for (int i = 0; i < 7 + NJ; ++i) qA[i] = TA(q[i]);
  kinematics<TA, NJ>(md, qA, kinA);
  convert_kin<T, TA, NJ>(kinA, t.kin);
  mass_matrix<T, NJ>(t.kin, dp.iota, t.M);
  T F[NV];
  bias_forces<T, NJ>(md, t.kin, dp.iota, v, F, t.Vw, t.Vu, t.AGw, t.AGu);
  if constexpr (MD::kGeneral && NJ > 0) {  // + B u: the actuators' generalized forces on their joints' coordinates
    DPLL_UNROLL for (int j = 0; j < NJ; ++j) F[6 + j] += dp.tau[j];
  }
  cholesky<T, NV>(t.M, t.LM, t.invdM);
  chol_solve<T, NV>(t.LM, t.invdM, F, t.a);
}

// One contact of this lane: geometry g = contact / 4 (the fast builds: geometry g sits on body g), slot 0..3.
// GEN (the general build) adds what a body-body contact needs: the other member of the pair and the contact frame.
template <typename T, int NJ, bool GEN> struct ContactPair {};
template <typename T, int NJ> struct ContactPair<T, NJ, true> {
  bool pair;       // body-body contact: geometry A (fields below) against geometry B (the fields of ContactGeom)
  int pidx;        // ... of candidate pidx (-1: a ground contact)
  int gpar;        // geometry whose parameters the witness of ContactGeom belongs to (a ground contact: = geom)
  int body_a, gpar_a;
  T sgn_a[3], drad_a[3];
  int vidx_a;
  T Ra[3][3];
  T F[3][3];       // rows: the contact frame's axes (t_x, t_y, n) in the world; the identity for a ground contact
  T dir[3];        // the pair's direction in the frame of A (unit, from A to B): piecewise constant in the state
};
template <typename T, int NJ, bool GEN = false> struct ContactGeom : ContactPair<T, NJ, GEN> {
  int body;
  int geom;      // geometry slot: whose pair friction coefficient mu is
  T sgn[3];      // box: corner sign pattern = d witness_i / d |length_i|; sphere: zero
  T drad[3];     // sphere: d witness / d radius (the support direction); box: zero
  int vidx;      // polygon: index of the vertex this contact picked (d witness / d vertices[vidx] = 1); otherwise -1
  T phi;
  T mu;
  T R[3][3];     // rotation of the contact's body
  CJac<T, NJ, GEN> J;  // contact-frame Jacobian (fast builds: world-frame point Jacobian, the ground's frame is the world's)
};

constexpr double kMaskedPhi = 1e3;  // signed distance of a contact slot that does not exist: no force, no gradient
constexpr double kPairTie = 1e-12;   // support values (metres) closer than this are a tie

// ---------------------------------------------------------------------------------------------
// Body-body contact (general build).  The reference (GeometryCollider.collide_mesh_mesh, geometry.py:585-643) asks
// fcl for ONE direction per pair -- the difference of the nearest points when the shapes are apart, a contact normal
// when they overlap -- treats it as piecewise constant, and evaluates everything else from the shapes' support
// functions: witness points p_Ac = s_A(d), p_Bc = s_B(-d), phi = (p_Bc - p_Ac) . d, contact frame
// rotation_matrix_from_one_vector(d, 2).  fcl's role is taken by an exact search: the unit d maximising the
// separation  sep(d) = min_b d . b - max_a d . a  of the two vertex sets (apart: the nearest-points direction;
// overlapping: the direction of minimum penetration), found among the directions the closest features of two convex
// polytopes can define -- vertex-vertex differences, vertex-edge perpendiculars, face normals of either set, cross
// products of an edge of each (every unit d is a lower bound of the maximum, so candidates that are not real features
// -- a polygon's hull is not known: all its vertex pairs and triples are tried -- cost time, never correctness).  A sphere is its centre with the radius as a margin (the margin shifts
// sep by a constant: same maximiser).
// ---------------------------------------------------------------------------------------------
// host/one-lane implementation of the lane-group primitives
struct OneLane {
  static constexpr int kGroup = 1;  // lanes that share one item
  static constexpr int kVariants = 1;  // racing copies of an item (SolverOpts::portfolio): the device builds only
  static DPLL_HD int variant() { return 0; }
  static DPLL_HD int item_or(int x) { return x; }
  template <typename T> static DPLL_HD T item_pick(bool, T x) { return x; }
  template <typename T> static DPLL_HD T group_sum(T x) { return x; }
  static DPLL_HD bool group_any(bool x) { return x; }
  static DPLL_HD bool wave_any(bool x) { return x; }
  static DPLL_HD int lane_in_group() { return 0; }
  // where an item's shared terms live: the caller's own object (registers / stack)
  template <class Store> static DPLL_HD Store& item_store(Store& local) { return local; }
  // the best (largest value; ties: smallest index) candidate over the lanes of the group, left in every lane
  template <typename S> static DPLL_HD void group_best(S&, int&, S (&)[3]) {}
  // where the group keeps a vertex set of the direction search (the host: the caller's array)
  template <typename S> static DPLL_HD S (*pair_storage(int, S (*local)[3]))[3] { return local; }
};

template <typename S> struct IsDual { static constexpr bool value = false; };
template <typename S> struct IsDual<DualT<S>> { static constexpr bool value = true; };
template <typename S> struct PairBest {
  S sep;
  S d[3];
  int k;  // number of the candidate that set it (ties between lanes: the earliest candidate wins, as in one lane)
};
// PAD8: both sets are stored with kMaxPolyVerts entries, the ones past the count repeating vertex 0 (a repeated vertex changes
// no maximum or minimum): the loops have a fixed length, and the loads of a set are issued together instead of one
// round trip to memory per vertex.
template <typename S, bool PAD8 = false>
DPLL_HD void pair_try(const S (&n)[3], int k, const S (*a)[3], int na, const S (*b)[3], int nb, PairBest<S>& best) {
  const S n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
  if (!(n2 > S(0))) return;
  S amax = S(-3.0e38), amin = S(3.0e38), bmax = S(-3.0e38), bmin = S(3.0e38);
  if constexpr (PAD8) {
    DPLL_UNROLL for (int i = 0; i < kMaxPolyVerts; ++i) {
      const S t = n[0] * a[i][0] + n[1] * a[i][1] + n[2] * a[i][2];
      amax = tmax(amax, t);
      amin = tmin(amin, t);
    }
    DPLL_UNROLL for (int j = 0; j < kMaxPolyVerts; ++j) {
      const S t = n[0] * b[j][0] + n[1] * b[j][1] + n[2] * b[j][2];
      bmax = tmax(bmax, t);
      bmin = tmin(bmin, t);
    }
  } else {
    for (int i = 0; i < na; ++i) {
      const S t = n[0] * a[i][0] + n[1] * a[i][1] + n[2] * a[i][2];
      amax = tmax(amax, t);
      amin = tmin(amin, t);
    }
    for (int j = 0; j < nb; ++j) {
      const S t = n[0] * b[j][0] + n[1] * b[j][1] + n[2] * b[j][2];
      bmax = tmax(bmax, t);
      bmin = tmin(bmin, t);
    }
  }
  const S inv = S(1) / tsqrt(n2);
  const S sp = (bmin - amax) * inv, sm = (amin - bmax) * inv;
  // (a candidate replaces the best so far only when it separates by MORE than kPairTie more: separations that differ by
  // rounding -- two faces of overlapping boxes that are equally deep -- are a tie, and a tie goes to the lower number)
  if (sp > best.sep + S(kPairTie)) {
    best.sep = sp;
    best.k = k;
    DPLL_UNROLL for (int i = 0; i < 3; ++i) best.d[i] = n[i] * inv;
  }
  if (sm > best.sep + S(kPairTie)) {
    best.sep = sm;
    best.k = k;
    DPLL_UNROLL for (int i = 0; i < 3; ++i) best.d[i] = -n[i] * inv;
  }
}
// Features of a vertex set that can define the direction: its edges (vertex-edge candidates), one edge per edge
// DIRECTION (edge x edge candidates) and one vertex triple per face normal.  A box: its 12 edges, 3 axis edges and 3
// faces; a polygon's hull is not known, so every vertex pair and triple stands in (a superset: correct, slower).
// Feature number -> vertex numbers is ARITHMETIC (no per-lane tables: a table indexed by the lane lives in scratch
// memory, and the search paid more for reading it than for its dot products).
struct PairFeatureCounts { int n_edges, n_dirs, n_tris; };
DPLL_HD PairFeatureCounts pair_feature_counts(int kind, int nv) {
  if (kind == kGeomBox) return {12, 3, 3};
  const int pairs = nv * (nv - 1) / 2;
  return {pairs, pairs, pairs * (nv - 2) / 3};
}
// m-th vertex pair (i < j, ordered by i then j) of nv vertices
DPLL_HD void nth_pair(int nv, int m, int& i, int& j) {
  i = 0;
  while (m >= nv - 1 - i) { m -= nv - 1 - i; ++i; }
  j = i + 1 + m;
}
// edge m: a box's corners u = bits (x y z), reference geometry.py:39-41 -- edges (u, u | bit) for u ascending, bit = 1, 2, 4
DPLL_HD void pair_edge(int kind, int nv, int m, int& k, int& l) {
  if (kind == kGeomBox) {
    // k = 0 0 0 1 1 2 2 3 4 4 5 6,  l = 1 2 4 3 5 3 6 7 5 6 7 7 as 3-bit fields
    constexpr unsigned long long kLo = 0ull | (0ull << 3) | (0ull << 6) | (1ull << 9) | (1ull << 12) | (2ull << 15) | (2ull << 18) | (3ull << 21) |
                                       (4ull << 24) | (4ull << 27) | (5ull << 30) | (6ull << 33);
    constexpr unsigned long long kHi = 1ull | (2ull << 3) | (4ull << 6) | (3ull << 9) | (5ull << 12) | (3ull << 15) | (6ull << 18) | (7ull << 21) |
                                       (5ull << 24) | (6ull << 27) | (7ull << 30) | (7ull << 33);
    k = int((kLo >> (3 * m)) & 7ull);
    l = int((kHi >> (3 * m)) & 7ull);
    return;
  }
  nth_pair(nv, m, k, l);
}
DPLL_HD void pair_dir(int kind, int nv, int m, int& k, int& l) {
  if (kind == kGeomBox) { k = 0; l = 1 << m; return; }
  nth_pair(nv, m, k, l);
}
DPLL_HD void pair_tri(int kind, int nv, int r, int& i, int& j, int& k) {
  if (kind == kGeomBox) { i = 0; j = r < 2 ? 1 : 2; k = r == 0 ? 2 : 4; return; }  // faces (0 1 2), (0 1 4), (0 2 4)
  i = 0;
  while (true) {  // triples that start with i: pairs of the nv - 1 - i vertices after it
    const int rest = nv - 1 - i, count = rest * (rest - 1) / 2;
    if (r < count) break;
    r -= count; ++i;
  }
  int jj, kk;
  nth_pair(nv - 1 - i, r, jj, kk);
  j = i + 1 + jj; k = i + 1 + kk;
}
// vertex sets a (na) and b (nb) in one frame -> unit d from A towards B; runtime loops (the sets live in memory).
// The candidates are numbered -- vertex-vertex, vertex(A)-edge(B), vertex(B)-edge(A), faces of A, faces of B, edge x edge
// -- and the lanes of the item's group take 16 consecutive numbers at a time (all of them busy at once), then agree on
// the best with one butterfly; one lane alone walks the same numbers in order.
template <typename S, class Lanes, bool PAD8 = false>
DPLL_HD_CALL void pair_direction(const S (*a)[3], int na, int kind_a, const S (*b)[3], int nb, int kind_b, S (&d)[3]) {
  PairBest<S> best;
  best.sep = S(-3.0e38);
  best.d[0] = S(0); best.d[1] = S(0); best.d[2] = S(1);
  best.k = 0x7fffffff;
  const PairFeatureCounts fa = pair_feature_counts(kind_a, na), fb = pair_feature_counts(kind_b, nb);
  const int n_vv = na * nb, n_veb = na * fb.n_edges, n_vea = nb * fa.n_edges, n_ta = fa.n_tris, n_tb = fb.n_tris;
  const int total = n_vv + n_veb + n_vea + n_ta + n_tb + fa.n_dirs * fb.n_dirs;
  const int lane = Lanes::lane_in_group();
  for (int base = 0; base < total; base += Lanes::kGroup) {
    const int c = base + lane;
    if (c >= total) continue;
    int r = c;
    S n[3];
    if (r < n_vv) {  // vertex - vertex
      const int i = r / nb, j = r % nb;
      DPLL_UNROLL for (int t = 0; t < 3; ++t) n[t] = b[j][t] - a[i][t];
    } else if ((r -= n_vv) < n_veb + n_vea) {  // vertex of one set - edge of the other: the perpendicular part
      const bool first = r < n_veb;
      if (!first) r -= n_veb;
      const S (*p)[3] = first ? a : b;
      const S (*e)[3] = first ? b : a;
      const int ne = first ? fb.n_edges : fa.n_edges;
      const int i = r / ne, m = r % ne;
      int k, l;
      pair_edge(first ? kind_b : kind_a, first ? nb : na, m, k, l);
      const S ed[3] = {e[l][0] - e[k][0], e[l][1] - e[k][1], e[l][2] - e[k][2]};
      const S w[3] = {e[k][0] - p[i][0], e[k][1] - p[i][1], e[k][2] - p[i][2]};
      const S ee = ed[0] * ed[0] + ed[1] * ed[1] + ed[2] * ed[2];
      const S t = ee > S(0) ? (w[0] * ed[0] + w[1] * ed[1] + w[2] * ed[2]) / ee : S(0);
      DPLL_UNROLL for (int q = 0; q < 3; ++q) n[q] = ee > S(0) ? w[q] - t * ed[q] : S(0);
    } else if ((r -= n_veb + n_vea) < n_ta + n_tb) {  // face normals of either set
      const bool first = r < n_ta;
      if (!first) r -= n_ta;
      const S (*p)[3] = first ? a : b;
      int i, j, k;
      pair_tri(first ? kind_a : kind_b, first ? na : nb, r, i, j, k);
      const S u[3] = {p[j][0] - p[i][0], p[j][1] - p[i][1], p[j][2] - p[i][2]};
      const S v[3] = {p[k][0] - p[i][0], p[k][1] - p[i][1], p[k][2] - p[i][2]};
      cross(u, v, n);
    } else {  // edge direction x edge direction
      r -= n_ta + n_tb;
      const int m = r / fb.n_dirs, o = r % fb.n_dirs;
      int i, j, k, l;
      pair_dir(kind_a, na, m, i, j);
      pair_dir(kind_b, nb, o, k, l);
      const S u[3] = {a[j][0] - a[i][0], a[j][1] - a[i][1], a[j][2] - a[i][2]};
      const S v[3] = {b[l][0] - b[k][0], b[l][1] - b[k][1], b[l][2] - b[k][2]};
      cross(u, v, n);
    }
    pair_try<S, PAD8>(n, c, a, na, b, nb, best);
  }
  Lanes::group_best(best.sep, best.k, best.d);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) d[i] = best.d[i];
}

// rotation_matrix_from_one_vector(d, axis = 2) (tensor_utils.py:305-366, after Drake's MakeFromOneVector): columns
// (b, c, a = d); `rows` receives its TRANSPOSE (row k = axis k of the contact frame, in the frame d is given in)
template <typename S> DPLL_HD void frame_from_normal(const S (&a)[3], S (&rows)[3][3]) {
  const S m0 = tabs(a[0]), m1 = tabs(a[1]), m2 = tabs(a[2]);
  const int i = (m0 <= m1 && m0 <= m2) ? 0 : (m1 <= m2 ? 1 : 2);  // torch.min: the first of equal minima
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  S ai = a[0], aj = a[1], ak = a[2];
  ai = i == 1 ? a[1] : (i == 2 ? a[2] : a[0]);
  aj = j == 1 ? a[1] : (j == 2 ? a[2] : a[0]);
  ak = k == 1 ? a[1] : (k == 2 ? a[2] : a[0]);
  const S mag = tsqrt(S(1) - ai * ai);
  const S corr = -ai / mag;
  S colb[3] = {S(0), S(0), S(0)}, colc[3] = {S(0), S(0), S(0)};
  DPLL_UNROLL for (int r = 0; r < 3; ++r) {
    colb[r] = r == j ? -ak / mag : (r == k ? aj / mag : S(0));
    colc[r] = r == i ? mag : (r == j ? corr * aj : corr * ak);
  }
  DPLL_UNROLL for (int r = 0; r < 3; ++r) { rows[0][r] = colb[r]; rows[1][r] = colc[r]; rows[2][r] = a[r]; }
}

// vertex set of geometry g in its own frame (+ the margin a sphere's radius adds)
template <typename S, typename T, int NJ, int NG, class MD>
DPLL_HD void geometry_vertices(const MD& md, const Derived<T, NJ, NG>& dp, int g, S (*v)[3], int& nv, S& margin) {
  const int kind = md.geom_kind[g];
  margin = S(0);
  if (kind == kGeomMesh) {
    nv = 0;  // (its witnesses are support points of the network, handed in)
    v[0][0] = S(0); v[0][1] = S(0); v[0][2] = S(0);
  } else if (kind == kGeomSphere) {
    nv = 1;
    v[0][0] = S(0); v[0][1] = S(0); v[0][2] = S(0);
    margin = S(dp.habs[g][0]);
  } else if (kind == kGeomPolygon) {
    nv = md.geom_nverts[g];
    for (int u = 0; u < nv; ++u)
      for (int i = 0; i < 3; ++i) v[u][i] = S(dp.geo[MD::kGeoStride * g + 3 * u + i]);
  } else {
    nv = 8;  // corner order of the reference's _UNIT_BOX_VERTICES (geometry.py:39-41): x is the slowest bit
    for (int u = 0; u < 8; ++u)
      for (int i = 0; i < 3; ++i) v[u][i] = (((u >> (2 - i)) & 1) ? S(1) : S(-1)) * S(dp.habs[g][i]);
  }
}

// the two geometry frames of a pair in the world and their vertex sets
template <typename S> struct PairSetup {
  S RA[3][3], RB[3][3], oA[3], oB[3], gorgA[3], gorgB[3], cA[3], cB[3];
  S va[kMaxPolyVerts][3], vb[kMaxPolyVerts][3], marginA, marginB;
  int na, nb;
};
template <typename T, typename TA, int NJ, int NG, class MD>
DPLL_HD void pair_setup(const MD& md, const Derived<T, NJ, NG>& dp, const Kin<TA, NJ>& kinA, int ga, int gb, PairSetup<TA>& ps) {
  constexpr int NB = NJ + 1;
  const int ba = md.geom_body[ga], bb = md.geom_body[gb];
  // geometry frames in the world (TA): rotation of the body, origin = body origin + R * geometry origin
  DPLL_UNROLL for (int r = 0; r < 3; ++r) {
    DPLL_UNROLL for (int c = 0; c < 3; ++c) { ps.RA[r][c] = kinA.R[0][r][c]; ps.RB[r][c] = kinA.R[0][r][c]; }
    ps.oA[r] = kinA.o[0][r]; ps.oB[r] = kinA.o[0][r];
  }
  DPLL_UNROLL for (int j = 1; j < NB; ++j)
    DPLL_UNROLL for (int r = 0; r < 3; ++r) {
      DPLL_UNROLL for (int c = 0; c < 3; ++c) {
        ps.RA[r][c] = ba == j ? kinA.R[j][r][c] : ps.RA[r][c];
        ps.RB[r][c] = bb == j ? kinA.R[j][r][c] : ps.RB[r][c];
      }
      ps.oA[r] = ba == j ? kinA.o[j][r] : ps.oA[r];
      ps.oB[r] = bb == j ? kinA.o[j][r] : ps.oB[r];
    }
  if (DPLL_ROTATED(md) & 2) {  // geometries turned in their bodies: from here on "the frame of A / B" is the geometry's own
    mat3_mul_const<TA>(ps.RA, md.geom_rot[ga]);
    mat3_mul_const<TA>(ps.RB, md.geom_rot[gb]);
  }
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { ps.gorgA[i] = TA(md.geom_origin[ga][i]); ps.gorgB[i] = TA(md.geom_origin[gb][i]); }
  mat3_vec(ps.RA, ps.gorgA, ps.cA);
  mat3_vec(ps.RB, ps.gorgB, ps.cB);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { ps.cA[i] += ps.oA[i]; ps.cB[i] += ps.oB[i]; }
  geometry_vertices<TA>(md, dp, ga, ps.va, ps.na, ps.marginA);
  geometry_vertices<TA>(md, dp, gb, ps.vb, ps.nb, ps.marginB);
}
// the pair's direction in the frame of A, searched by the lanes of the item's group together
template <typename TA, class Lanes>
DPLL_HD void pair_search(const PairSetup<TA>& ps, int kind_a, int kind_b, TA (&dA)[3]) {
  // both sets where the candidate loops read them from: the group's on-chip storage on the device (every lane of the
  // group writes the same values), plain arrays on the host
  TA la[kMaxPolyVerts][3], lb[kMaxPolyVerts][3];
  TA (*sa)[3] = Lanes::template pair_storage<TA>(0, la);
  TA (*sb)[3] = Lanes::template pair_storage<TA>(1, lb);
  for (int u = 0; u < ps.na; ++u)
    DPLL_UNROLL for (int i = 0; i < 3; ++i) sa[u][i] = ps.va[u][i];
  // B's vertices in the frame of A: R_A^T (c_B + R_B v - c_A)
  for (int u = 0; u < ps.nb; ++u) {
    TA w[3], rel[3], out[3];
    mat3_vec(ps.RB, ps.vb[u], w);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) rel[i] = w[i] + ps.cB[i] - ps.cA[i];
    mat3t_vec(ps.RA, rel, out);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) sb[u][i] = out[i];
  }
  pair_direction<TA, Lanes>(sa, ps.na, kind_a, sb, ps.nb, kind_b, dA);
}
// every body-body candidate's direction, before the contacts are set up (all lanes of the group take part)
// `mesh_dirs` (general build with mesh geometry): the directions of the candidates between two learned shapes, found by the
// GJK / EPA kernel (csrc/dpll_gjk.hpp) before this kernel ran; the other candidates are searched here
template <typename T, typename TA, class Lanes, int NJ, int NG, class MD>
DPLL_HD bool pair_find_directions(const MD& md, const Derived<T, NJ, NG>& dp, const Kin<TA, NJ>& kinA, TA (&dirs)[kMaxPairs][3],
                                  const TA (*mesh_dirs)[3] = nullptr) {
  DPLL_UNROLL for (int p = 0; p < kMaxPairs; ++p) { dirs[p][0] = TA(0); dirs[p][1] = TA(0); dirs[p][2] = TA(1); }
  if constexpr (MD::kGeneral && !IsDual<TA>::value) {
    if (md.n_pairs <= 0) return false;
    for (int p = 0; p < kMaxPairs; ++p) {
      if (p >= md.n_pairs) break;
      if (md.geom_kind[md.pair_a[p]] == kGeomMesh || md.geom_kind[md.pair_b[p]] == kGeomMesh) {
        if (mesh_dirs) {
          DPLL_UNROLL for (int i = 0; i < 3; ++i) dirs[p][i] = mesh_dirs[p][i];
        }
        continue;
      }
      PairSetup<TA> ps;
      pair_setup<T, TA, NJ>(md, dp, kinA, md.pair_a[p], md.pair_b[p], ps);
      pair_search<TA, Lanes>(ps, md.geom_kind[md.pair_a[p]], md.geom_kind[md.pair_b[p]], dirs[p]);
    }
    return true;
  }
  return false;
}
// the found direction of the pair contact `contact` sits in, or nullptr (not a pair slot / nothing found up front)
template <typename TA> DPLL_HD const TA* pair_dir_of(bool have, const TA (&dirs)[kMaxPairs][3], int contact) {
  const int p = contact - kQuery * kMaxGeoms;
  return (have && p >= 0 && p < kMaxPairs) ? dirs[p] : nullptr;
}

// `witness` (mesh geometry, DeepSupportConvex): the support point of this contact in the geometry frame, already
// evaluated by the ICNN kernels (geometry.py:309-325); nullptr = box corner / sphere point chosen here.
template <typename T, typename TA, int NJ, int NG, class MD>
DPLL_HD void compute_pair_contact(const MD& md, const Derived<T, NJ, NG>& dp, const Kin<T, NJ>& kin,
                                  const Kin<TA, NJ>& kinA, int p, ContactGeom<T, NJ, true>& cg,
                                  const TA* dir_in, const T* wit_b = nullptr, const T* wit_a = nullptr) {
  const bool masked = p >= md.n_pairs;
  const int ga = masked ? 0 : md.pair_a[p], gb = masked ? 0 : md.pair_b[p];
  const int ba = md.geom_body[ga], bb = md.geom_body[gb];
  cg.pair = true;
  cg.geom = kMaxGeoms;
  cg.pidx = p;
  cg.mu = dp.mu_pair[0];
  DPLL_UNROLL for (int pp = 1; pp < kMaxPairs; ++pp) cg.mu = (p == pp) ? dp.mu_pair[pp] : cg.mu;
  cg.body = bb; cg.gpar = gb;
  cg.body_a = ba; cg.gpar_a = ga;
  cg.vidx = -1; cg.vidx_a = -1;
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { cg.sgn[i] = T(0); cg.drad[i] = T(0); cg.sgn_a[i] = T(0); cg.drad_a[i] = T(0); cg.dir[i] = T(i == 2 ? 1 : 0); }
  DPLL_UNROLL for (int r = 0; r < 3; ++r)
    DPLL_UNROLL for (int c = 0; c < 3; ++c) { cg.R[r][c] = kin.R[0][r][c]; cg.Ra[r][c] = kin.R[0][r][c]; cg.F[r][c] = T(r == c ? 1 : 0); }
  DPLL_UNROLL for (int r = 0; r < 3; ++r)
    DPLL_UNROLL for (int c = 0; c < 6 + NJ; ++c) cg.J.m[r][c] = T(0);
  cg.phi = T(kMaskedPhi);
  if (masked) return;
  PairSetup<TA> ps;
  pair_setup<T, TA, NJ>(md, dp, kinA, ga, gb, ps);
  TA (&RA)[3][3] = ps.RA; TA (&RB)[3][3] = ps.RB;
  TA (&oA)[3] = ps.oA; TA (&oB)[3] = ps.oB; TA (&gorgA)[3] = ps.gorgA; TA (&gorgB)[3] = ps.gorgB;
  TA (&va)[kMaxPolyVerts][3] = ps.va; TA (&vb)[kMaxPolyVerts][3] = ps.vb;
  const TA marginA = ps.marginA, marginB = ps.marginB;
  const int na = ps.na, nb = ps.nb;
  // direction, in the frame of A
  TA dA[3];
  if (dir_in) {
    DPLL_UNROLL for (int i = 0; i < 3; ++i) dA[i] = dir_in[i];
  } else if constexpr (IsDual<TA>::value) {
    // (the dual passes of the state adjoint always get the direction of the primal pass: no search code for them)
    DPLL_UNROLL for (int i = 0; i < 3; ++i) dA[i] = TA(i == 2 ? 1.0 : 0.0);
  } else {
    pair_search<TA, OneLane>(ps, md.geom_kind[ga], md.geom_kind[gb], dA);  // (callers with a lane group search up front)
  }
  TA dW[3], dB[3], ndW[3];
  mat3_vec(RA, dA, dW);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) ndW[i] = -dW[i];
  mat3t_vec(RB, ndW, dB);  // -d in the frame of B
  // witness points: the vertex furthest along the direction + the sphere margin along it.  When the direction is a face
  // normal or an edge normal of the shape itself several vertices are equally far up to rounding: the lowest index of
  // those within kPairTie wins (a support function network has no such ties: its gradient is one vertex).
  int ia = 0, ib = 0;
  TA besta = TA(-3.0e38), bestb = TA(-3.0e38);
  for (int u = 0; u < na; ++u) {
    const TA t = dA[0] * va[u][0] + dA[1] * va[u][1] + dA[2] * va[u][2];
    if (t > besta + TA(kPairTie)) { besta = t; ia = u; }
  }
  for (int u = 0; u < nb; ++u) {
    const TA t = dB[0] * vb[u][0] + dB[1] * vb[u][1] + dB[2] * vb[u][2];
    if (t > bestb + TA(kPairTie)) { bestb = t; ib = u; }
  }
  const int kindA = md.geom_kind[ga], kindB = md.geom_kind[gb];
  TA witA[3], witB[3];
  DPLL_UNROLL for (int i = 0; i < 3; ++i) {
    // learned shapes: geometry.network(d), geometry.network(-d) of collide_mesh_mesh (geometry.py:627-629), evaluated by the
    // ICNN kernels at the direction the GJK / EPA kernel found; otherwise the vertex furthest along the direction
    witA[i] = (kindA == kGeomMesh) ? (wit_a ? TA(wit_a[i]) : TA(0)) : va[na > 0 ? ia : 0][i] + marginA * dA[i];
    witB[i] = (kindB == kGeomMesh) ? (wit_b ? TA(wit_b[i]) : TA(0)) : vb[nb > 0 ? ib : 0][i] + marginB * dB[i];
  }
  // d witness / d parameters
  DPLL_UNROLL for (int i = 0; i < 3; ++i) {
    cg.sgn_a[i] = kindA == kGeomBox ? (((ia >> (2 - i)) & 1) ? T(1) : T(-1)) : T(0);
    cg.sgn[i] = kindB == kGeomBox ? (((ib >> (2 - i)) & 1) ? T(1) : T(-1)) : T(0);
    cg.drad_a[i] = kindA == kGeomSphere ? T(dA[i]) : T(0);
    cg.drad[i] = kindB == kGeomSphere ? T(dB[i]) : T(0);
    cg.dir[i] = T(dA[i]);
  }
  cg.vidx_a = kindA == kGeomPolygon ? ia : -1;
  cg.vidx = kindB == kGeomPolygon ? ib : -1;
  // world points, signed distance along d
  TA rA[3], rB[3], ptA[3], ptB[3], wA[3], wB[3];
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { rA[i] = gorgA[i] + witA[i]; rB[i] = gorgB[i] + witB[i]; }
  mat3_vec(RA, rA, wA);
  mat3_vec(RB, rB, wB);
  TA phi = TA(0);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { ptA[i] = oA[i] + wA[i]; ptB[i] = oB[i] + wB[i]; phi += dW[i] * (ptB[i] - ptA[i]); }
  cg.phi = T(phi);
  // contact frame: rows of R_AF^T in the frame of A, taken to the world by R_A
  TA FA[3][3];
  frame_from_normal<TA>(dA, FA);
  T F[3][3];
  DPLL_UNROLL for (int k = 0; k < 3; ++k) {
    TA axis[3];
    mat3_vec(RA, FA[k], axis);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { F[k][i] = T(axis[i]); cg.F[k][i] = F[k][i]; }
  }
  DPLL_UNROLL for (int r = 0; r < 3; ++r)
    DPLL_UNROLL for (int c = 0; c < 3; ++c) { cg.Ra[r][c] = T(RA[r][c]); cg.R[r][c] = T(RB[r][c]); }
  // J = F (J_Bc - J_Ac)  (multibody_terms.py:503-513)
  T pa[3], pb[3];
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { pa[i] = T(ptA[i]); pb[i] = T(ptB[i]); }
  CJac<T, NJ, false> JA, JB;
  contact_jacobian<T, NJ>(kin, ba, pa, JA);
  contact_jacobian<T, NJ>(kin, bb, pb, JB);
  cjac_dense<T, NJ>(JB, F, T(1), false, cg.J);
  cjac_dense<T, NJ>(JA, F, T(-1), true, cg.J);
}

template <typename T, typename TA, int NJ, int NG, class MD>
DPLL_HD void compute_contact(const MD& md, const Derived<T, NJ, NG>& dp, const Kin<T, NJ>& kin,
                             const Kin<TA, NJ>& kinA, int contact, ContactGeom<T, NJ, MD::kGeneral>& cg,
                             const T* witness = nullptr, const TA* pair_dir = nullptr, const T* witness_a = nullptr) {
  constexpr int NB = NJ + 1;
  const int g = contact / kQuery;
  const int slot = contact % kQuery;
  if constexpr (MD::kGeneral) {
    if (g >= kMaxGeoms) {  // the group behind the geometries: slot p is body-body pair p
      compute_pair_contact<T, TA, NJ>(md, dp, kin, kinA, slot, cg, pair_dir, witness, witness_a);
      return;
    }
  }
  // geometry data by g (g differs from lane to lane in the lane-per-contact builds)
  T habs[3], gorg[3];
  T mu = dp.mu[0];
  int b = 0, kind = kGeomBox;
  bool masked = false;
  DPLL_UNROLL for (int r = 0; r < 3; ++r) { habs[r] = dp.habs[0][r]; gorg[r] = T(md.geom_origin[0][r]); }
  if constexpr (MD::kGeneral) { b = md.geom_body[0]; kind = md.geom_kind[0]; masked = md.n_geoms < 1; }
  DPLL_UNROLL for (int gg = 1; gg < (MD::kGeneral ? kMaxGeoms : NG); ++gg) {
    const bool pick = (g == gg);
    DPLL_UNROLL for (int r = 0; r < 3; ++r) {
      habs[r] = pick ? dp.habs[gg][r] : habs[r];
      gorg[r] = pick ? T(md.geom_origin[gg][r]) : gorg[r];
    }
    mu = pick ? dp.mu[gg] : mu;
    if constexpr (MD::kGeneral) {
      b = pick ? md.geom_body[gg] : b;
      kind = pick ? md.geom_kind[gg] : kind;
      masked = pick ? (md.n_geoms < gg + 1) : masked;
    } else {
      b = pick ? gg : b;
    }
  }
  cg.body = b;
  cg.geom = g;
  cg.mu = mu;
  // body data by b
  T o[3];
  TA Rz[3], oz = kinA.o[0][2];  // third row of the body rotation and origin height, in TA, for phi
  DPLL_UNROLL for (int r = 0; r < 3; ++r) {
    DPLL_UNROLL for (int c = 0; c < 3; ++c) cg.R[r][c] = kin.R[0][r][c];
    o[r] = kin.o[0][r];
    Rz[r] = kinA.R[0][2][r];
  }
  DPLL_UNROLL for (int j = 1; j < NB; ++j) {
    const bool pick = (b == j);
    DPLL_UNROLL for (int r = 0; r < 3; ++r) {
      DPLL_UNROLL for (int c = 0; c < 3; ++c) cg.R[r][c] = pick ? kin.R[j][r][c] : cg.R[r][c];
      o[r] = pick ? kin.o[j][r] : o[r];
      Rz[r] = pick ? kinA.R[j][2][r] : Rz[r];
    }
    oz = pick ? kinA.o[j][2] : oz;
  }
  if constexpr (MD::kGeneral) {
    if (DPLL_ROTATED(md) & 2) turn_to_geometry<T, TA>(md.geom_rot, g, cg.R, Rz);
  }
  // support direction in the geometry frame: -(row 2 of R_AB) (geometry.py:560-564)
  const T d[3] = {-cg.R[2][0], -cg.R[2][1], -cg.R[2][2]};
  T wit[3];
  DPLL_UNROLL for (int i = 0; i < 3; ++i) cg.drad[i] = T(0);
  cg.vidx = -1;
  bool use_witness = witness != nullptr;
  if constexpr (MD::kGeneral) use_witness = use_witness && kind == kGeomMesh;  // (the model's other geometries: chosen here)
  if (use_witness) {
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { cg.sgn[i] = T(0); wit[i] = witness[i]; }
  } else {
    box_corner_signs(d, habs, slot, cg.sgn);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) wit[i] = cg.sgn[i] * habs[i];
    if constexpr (MD::kGeneral) {
      // Sphere.support_points (geometry.py:440-452): ONE witness, direction * radius; slots 1..3 do not exist
      const bool sphere = (kind == kGeomSphere);
      masked = masked || (sphere && slot != 0);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) {
        wit[i] = sphere ? d[i] * habs[0] : wit[i];
        cg.drad[i] = sphere ? d[i] : T(0);
        cg.sgn[i] = sphere ? T(0) : cg.sgn[i];
      }
      // Polygon (geometry.py:220-252 through SparseVertexConvexCollisionGeometry.support_points, :162-202): the
      // n_query = 4 vertices with the largest d . vertex, in no particular order (quirk Q3).  Slot s takes the vertex
      // of rank s (ties by index); the witness is that vertex itself, so its adjoint goes to the vertex unchanged.
      if (kind == kGeomPolygon) {
        T vert[kMaxPolyVerts][3], dots[kMaxPolyVerts];
        const int nv = md.geom_nverts[g < kMaxGeoms ? g : 0];
        DPLL_UNROLL for (int u = 0; u < kMaxPolyVerts; ++u) {
          DPLL_UNROLL for (int i = 0; i < 3; ++i) vert[u][i] = dp.geo[MD::kGeoStride * g + 3 * u + i];
          dots[u] = u < nv ? d[0] * vert[u][0] + d[1] * vert[u][1] + d[2] * vert[u][2] : T(-3.0e38);
        }
        DPLL_UNROLL for (int i = 0; i < 3; ++i) { wit[i] = T(0); cg.sgn[i] = T(0); }
        DPLL_UNROLL for (int u = 0; u < kMaxPolyVerts; ++u) {
          int rank = 0;
          DPLL_UNROLL for (int o2 = 0; o2 < kMaxPolyVerts; ++o2)
            rank += (o2 != u && (o2 < u ? dots[o2] >= dots[u] : dots[o2] > dots[u])) ? 1 : 0;  // ties: lower index first
          const bool mine = (rank == slot);
          cg.vidx = mine ? u : cg.vidx;
          DPLL_UNROLL for (int i = 0; i < 3; ++i) wit[i] = mine ? vert[u][i] : wit[i];
        }
      }
    }
  }
  T r_b[3], rho[3], pt[3];
  DPLL_UNROLL for (int i = 0; i < 3; ++i) r_b[i] = gorg[i] + wit[i];
  mat3_vec(cg.R, r_b, rho);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) pt[i] = o[i] + rho[i];
  TA phiA = oz;
  DPLL_UNROLL for (int i = 0; i < 3; ++i) phiA += Rz[i] * (TA(gorg[i]) + TA(wit[i]));
  cg.phi = masked ? T(kMaskedPhi) : T(phiA);
  if constexpr (MD::kGeneral) {
    cg.pair = false;
    cg.pidx = -1;
    cg.gpar = g;
    cg.body_a = 0; cg.gpar_a = 0; cg.vidx_a = -1;
    DPLL_UNROLL for (int r = 0; r < 3; ++r) {
      cg.sgn_a[r] = T(0); cg.drad_a[r] = T(0); cg.dir[r] = T(0);
      DPLL_UNROLL for (int c = 0; c < 3; ++c) { cg.F[r][c] = T(r == c ? 1 : 0); cg.Ra[r][c] = T(r == c ? 1 : 0); }
    }
    CJac<T, NJ, false> P;
    contact_jacobian<T, NJ>(kin, b, pt, P);
    cjac_dense<T, NJ>(P, cg.F, T(1), false, cg.J);
  } else {
    contact_jacobian<T, NJ>(kin, b, pt, cg.J);
  }
}

// d/d(witness points) of  a1 . (J w1) + a2 . (J w2) + phibar phi  for one contact (a1, a2: contact-frame vectors with
// the friction coefficient folded in): the world-frame point adjoint  rho_bar = a1 x omega(w1) + a2 x omega(w2) +
// phibar n  taken to the body frame; a body-body contact has the same with the opposite sign on the side of A.
template <typename T, int NJ, bool GEN>
DPLL_HD void witness_adjoint(const Kin<T, NJ>& kin, const ContactGeom<T, NJ, GEN>& cg, const T (&a1)[3], const T* w1,
                             const T (&a2)[3], const T* w2, T phibar, T (&rbar)[3], T (&rbar_a)[3]) {
  T aw1[3], aw2[3], nW[3] = {T(0), T(0), T(1)};
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { aw1[i] = a1[i]; aw2[i] = a2[i]; rbar_a[i] = T(0); }
  if constexpr (GEN) {
    DPLL_UNROLL for (int i = 0; i < 3; ++i) {
      aw1[i] = a1[0] * cg.F[0][i] + a1[1] * cg.F[1][i] + a1[2] * cg.F[2][i];
      aw2[i] = a2[0] * cg.F[0][i] + a2[1] * cg.F[1][i] + a2[2] * cg.F[2][i];
      nW[i] = cg.F[2][i];
    }
  }
  T o1[3], o2[3], c1[3], c2[3], rho[3];
  world_omega<T, NJ>(kin, cg.body, w1, o1);
  world_omega<T, NJ>(kin, cg.body, w2, o2);
  cross(aw1, o1, c1);
  cross(aw2, o2, c2);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) rho[i] = c1[i] + c2[i] + phibar * nW[i];
  mat3t_vec(cg.R, rho, rbar);
  if constexpr (GEN) {
    world_omega<T, NJ>(kin, cg.body_a, w1, o1);
    world_omega<T, NJ>(kin, cg.body_a, w2, o2);
    cross(aw1, o1, c1);
    cross(aw2, o2, c2);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) rho[i] = cg.pair ? -(c1[i] + c2[i] + phibar * nW[i]) : T(0);
    mat3t_vec(cg.Ra, rho, rbar_a);
  }
}

}  // namespace dpll
