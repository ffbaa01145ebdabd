// dpll_gjk.hpp -- the body-body direction of two LEARNED convex shapes (DeepSupportConvex x DeepSupportConvex).
//
// dair_pll routes exactly this pair to GeometryCollider.collide_mesh_mesh (geometry.py:543-546, :585-643): fcl is handed
// the two meshes that extract_mesh builds from the networks' support points over the 296 surface directions
// (geometry.py:343-358, deep_support_function.py:12-16, :93-123) and is asked for ONE direction per item -- the
// difference of the nearest points when the meshes are apart (fcl.distance, :621-625), a contact normal when they
// overlap (fcl.collide, :615-618).  fcl is not in this image and is unpinned; its role is restated (oracle/dpll_oracle.py
// pair_direction_exact) as the unit direction d that maximises the separation  min_b d.b - max_a d.a  of the two vertex
// sets: with C = B - A (Minkowski difference) that is the direction of the point of C closest to the origin when the
// origin is outside C, and minus the outward normal of the facet of C closest to the origin when it is inside.
//
// Here: GJK for the first case and EPA for the second, on support functions of the two vertex arrays (<= 296 vertices
// each, which rules out the candidate enumeration pair_direction uses for boxes and polygons).  Both terminate on
// DISCRETE events -- the support vertex returned is already part of the simplex / polytope -- so the answer is the exact
// feature of the two polytopes up to the rounding of one projection, not an iteration tolerance.
// One item is worked by the lanes of its group (16 on the device): the support scans are split over the lanes and
// combined with one DPP butterfly; the small simplex / polytope bookkeeping is replicated (every lane computes and stores
// the same values, so no lane reads what another lane alone has written).
#pragma once
#include "dpll_core.hpp"

namespace dpll {

constexpr int kHullDirs = 296;     // surface directions of the reference's mesh extraction: the vertex arrays' length
constexpr int kEpaMaxVerts = 96;   // points of C the expanding polytope may hold
constexpr int kEpaMaxFaces = 192;
constexpr int kEpaMaxHorizon = 64;
constexpr int kGjkMaxIter = 64;

// the two vertex arrays and the pose of B in the frame of A:  x_A = R x_B + p
template <typename S> struct HullPair {
  const S (*va)[3];
  int na;
  const S (*vb)[3];
  int nb;
  S R[3][3], p[3];
};

// polytope of the expansion; on the device this lives in LDS, one per lane group
template <typename S> struct EpaStore {
  S vx[kEpaMaxVerts][3];
  short via[kEpaMaxVerts], vib[kEpaMaxVerts];
  S fn[kEpaMaxFaces][3], fd[kEpaMaxFaces];   // outward unit normal and plane offset (>= 0: the origin is inside); fd < 0: dead
  unsigned char tri[kEpaMaxFaces][3];
  unsigned char hz[kEpaMaxHorizon][2];
};

// the vertex of `v` furthest along w (ties: the lowest index), scanned by the lanes of the group together
template <typename S, class Lanes>
DPLL_HD void hull_support(const S (*v)[3], int n, const S (&w)[3], int& idx, S (&pt)[3]) {
  S best = S(-3.0e38);
  int bi = 0x7fffffff;
  S bp[3] = {S(0), S(0), S(0)};
  for (int u = Lanes::lane_in_group(); u < n; u += Lanes::kGroup) {
    const S t = w[0] * v[u][0] + w[1] * v[u][1] + w[2] * v[u][2];
    if (t > best) { best = t; bi = u; bp[0] = v[u][0]; bp[1] = v[u][1]; bp[2] = v[u][2]; }
  }
  Lanes::group_best(best, bi, bp);
  idx = bi;
  DPLL_UNROLL for (int i = 0; i < 3; ++i) pt[i] = bp[i];
}

// support point of C = (R B + p) - A along w (frame of A) and the vertices of A and B that make it
template <typename S, class Lanes>
DPLL_HD void minkowski_support(const HullPair<S>& hp, const S (&w)[3], S (&c)[3], int& ia, int& ib) {
  const S wn[3] = {-w[0], -w[1], -w[2]};
  S wb[3], a[3], b[3], rb[3];
  mat3t_vec(hp.R, w, wb);
  hull_support<S, Lanes>(hp.va, hp.na, wn, ia, a);
  hull_support<S, Lanes>(hp.vb, hp.nb, wb, ib, b);
  mat3_vec(hp.R, b, rb);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) c[i] = rb[i] + hp.p[i] - a[i];
}

// ---- the point of a simplex closest to the origin (Ericson, Real-Time Collision Detection 5.1.2, 5.1.5, 5.1.6) ----
// barycentric weights in lam (zero weight = the vertex does not support the closest point)
template <typename S> DPLL_HD void closest_segment(const S (&a)[3], const S (&b)[3], S (&lam)[4]) {
  const S ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  const S den = dot3(ab, ab);
  S t = den > S(0) ? -dot3(a, ab) / den : S(0);
  t = t < S(0) ? S(0) : (t > S(1) ? S(1) : t);
  lam[0] = S(1) - t; lam[1] = t; lam[2] = S(0); lam[3] = S(0);
}
template <typename S> DPLL_HD void closest_triangle(const S (&a)[3], const S (&b)[3], const S (&c)[3], S (&lam)[4]) {
  const S ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
  const S d1 = -dot3(ab, a), d2 = -dot3(ac, a);
  lam[3] = S(0);
  if (d1 <= S(0) && d2 <= S(0)) { lam[0] = S(1); lam[1] = S(0); lam[2] = S(0); return; }
  const S d3 = -dot3(ab, b), d4 = -dot3(ac, b);
  if (d3 >= S(0) && d4 <= d3) { lam[0] = S(0); lam[1] = S(1); lam[2] = S(0); return; }
  const S vc = d1 * d4 - d3 * d2;
  if (vc <= S(0) && d1 >= S(0) && d3 <= S(0)) { const S t = d1 / (d1 - d3); lam[0] = S(1) - t; lam[1] = t; lam[2] = S(0); return; }
  const S d5 = -dot3(ab, c), d6 = -dot3(ac, c);
  if (d6 >= S(0) && d5 <= d6) { lam[0] = S(0); lam[1] = S(0); lam[2] = S(1); return; }
  const S vb = d5 * d2 - d1 * d6;
  if (vb <= S(0) && d2 >= S(0) && d6 <= S(0)) { const S t = d2 / (d2 - d6); lam[0] = S(1) - t; lam[1] = S(0); lam[2] = t; return; }
  const S va = d3 * d6 - d5 * d4;
  if (va <= S(0) && (d4 - d3) >= S(0) && (d5 - d6) >= S(0)) {
    const S t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    lam[0] = S(0); lam[1] = S(1) - t; lam[2] = t;
    return;
  }
  const S den = S(1) / (va + vb + vc);
  lam[1] = vb * den; lam[2] = vc * den; lam[0] = S(1) - lam[1] - lam[2];
}

// GJK simplex: up to 4 points of C with the vertex pairs they come from
template <typename S> struct GjkSimplex {
  S pt[4][3];
  int ia[4], ib[4];
  int n;
};
template <typename S> DPLL_HD void simplex_point(const GjkSimplex<S>& sx, const S (&lam)[4], S (&v)[3]) {
  DPLL_UNROLL for (int i = 0; i < 3; ++i) {
    S s = S(0);
    DPLL_UNROLL for (int k = 0; k < 4; ++k) s += (k < sx.n) ? lam[k] * sx.pt[k][i] : S(0);
    v[i] = s;
  }
}
// keeps the vertices with a positive weight (order preserved)
template <typename S> DPLL_HD void simplex_reduce(GjkSimplex<S>& sx, const S (&lam)[4]) {
  int m = 0;
  DPLL_UNROLL for (int k = 0; k < 4; ++k) {
    if (k < sx.n && lam[k] > S(0)) {
      DPLL_UNROLL for (int j = 0; j < 4; ++j)
        if (j == m) {
          DPLL_UNROLL for (int i = 0; i < 3; ++i) sx.pt[j][i] = sx.pt[k][i];
          sx.ia[j] = sx.ia[k]; sx.ib[j] = sx.ib[k];
        }
      ++m;
    }
  }
  sx.n = m;
}
// closest point of the simplex to the origin; returns true when the simplex is a tetrahedron that contains the origin
template <typename S> DPLL_HD bool simplex_closest(GjkSimplex<S>& sx, S (&v)[3]) {
  S lam[4] = {S(1), S(0), S(0), S(0)};
  if (sx.n == 2) closest_segment(sx.pt[0], sx.pt[1], lam);
  else if (sx.n == 3) closest_triangle(sx.pt[0], sx.pt[1], sx.pt[2], lam);
  else if (sx.n == 4) {
    // the faces the origin is outside of (on the other side than the fourth vertex); the closest of their closest points
    S best = S(3.0e38);
    bool outside_any = false;
    S blam[4] = {S(0), S(0), S(0), S(0)};
    DPLL_UNROLL for (int f = 0; f < 4; ++f) {
      // face f leaves out vertex f
      const int i0 = f == 0 ? 1 : 0, i1 = f <= 1 ? 2 : 1, i2 = f <= 2 ? 3 : 2;
      S e1[3], e2[3], nrm[3];
      DPLL_UNROLL for (int i = 0; i < 3; ++i) { e1[i] = sx.pt[i1][i] - sx.pt[i0][i]; e2[i] = sx.pt[i2][i] - sx.pt[i0][i]; }
      cross(e1, e2, nrm);
      S to4[3];
      DPLL_UNROLL for (int i = 0; i < 3; ++i) to4[i] = sx.pt[f][i] - sx.pt[i0][i];
      const S side4 = dot3(nrm, to4), side0 = -dot3(nrm, sx.pt[i0]);
      // origin and the fourth vertex on opposite sides (or a flat tetrahedron: treat the face as a candidate)
      const bool outside = side4 * side0 < S(0) || side4 == S(0);
      if (outside) {
        S fl[4];
        closest_triangle(sx.pt[i0], sx.pt[i1], sx.pt[i2], fl);
        S q[3];
        DPLL_UNROLL for (int i = 0; i < 3; ++i) q[i] = fl[0] * sx.pt[i0][i] + fl[1] * sx.pt[i1][i] + fl[2] * sx.pt[i2][i];
        const S d2 = dot3(q, q);
        if (d2 < best) {
          best = d2;
          DPLL_UNROLL for (int k = 0; k < 4; ++k) blam[k] = S(0);
          DPLL_UNROLL for (int k = 0; k < 4; ++k) blam[k] = (k == i0) ? fl[0] : ((k == i1) ? fl[1] : ((k == i2) ? fl[2] : S(0)));
        }
        outside_any = true;
      }
    }
    if (!outside_any) { v[0] = S(0); v[1] = S(0); v[2] = S(0); return true; }
    DPLL_UNROLL for (int k = 0; k < 4; ++k) lam[k] = blam[k];
  }
  simplex_point(sx, lam, v);
  simplex_reduce(sx, lam);
  return false;
}

// ---- EPA ------------------------------------------------------------------------------------------------------------
template <typename S> DPLL_HD void epa_set_face(EpaStore<S>& st, int f, int i, int j, int k) {
  S e1[3], e2[3], n[3];
  DPLL_UNROLL for (int c = 0; c < 3; ++c) { e1[c] = st.vx[j][c] - st.vx[i][c]; e2[c] = st.vx[k][c] - st.vx[i][c]; }
  cross(e1, e2, n);
  const S n2 = dot3(n, n);
  const S inv = n2 > S(0) ? S(1) / tsqrt(n2) : S(0);
  DPLL_UNROLL for (int c = 0; c < 3; ++c) n[c] *= inv;
  // plane offset: the mean over the three vertices (they agree to rounding)
  const S vi[3] = {st.vx[i][0], st.vx[i][1], st.vx[i][2]}, vj[3] = {st.vx[j][0], st.vx[j][1], st.vx[j][2]}, vk[3] = {st.vx[k][0], st.vx[k][1], st.vx[k][2]};
  const S d = (dot3(n, vi) + dot3(n, vj) + dot3(n, vk)) * S(1.0 / 3.0);
  st.tri[f][0] = (unsigned char)i; st.tri[f][1] = (unsigned char)j; st.tri[f][2] = (unsigned char)k;
  DPLL_UNROLL for (int c = 0; c < 3; ++c) st.fn[f][c] = n[c];
  st.fd[f] = n2 > S(0) ? tmax(d, S(0)) : S(-1);  // a degenerate triangle is born dead
}

// Result of the search: the unit direction from A towards B and what it was derived from
template <typename S> struct PairDirResult {
  S d[3];
  S sep;         // separation along d (> 0 apart, < 0 overlapping: minus the penetration depth)
  int gjk_iters, epa_iters;
  int status;    // 0 ok, 1 GJK ran out of iterations, 2 EPA ran out of room / iterations (best face so far), 3 degenerate start
};

template <typename S, class Lanes>
DPLL_HD_CALL void hull_pair_direction(const HullPair<S>& hp, EpaStore<S>& st, PairDirResult<S>& out) {
  out.status = 0; out.gjk_iters = 0; out.epa_iters = 0;
  out.d[0] = S(0); out.d[1] = S(0); out.d[2] = S(1); out.sep = S(0);
  // ---- GJK: the point of C closest to the origin -----------------------------------------------------------------
  GjkSimplex<S> sx;
  S v[3];
  {
    S w0[3] = {hp.p[0], hp.p[1], hp.p[2]};
    if (!(dot3(w0, w0) > S(0))) { w0[0] = S(1); w0[1] = S(0); w0[2] = S(0); }
    const S wneg[3] = {-w0[0], -w0[1], -w0[2]};
    minkowski_support<S, Lanes>(hp, wneg, sx.pt[0], sx.ia[0], sx.ib[0]);
    sx.n = 1;
    DPLL_UNROLL for (int k = 1; k < 4; ++k) { sx.ia[k] = -1; sx.ib[k] = -1; DPLL_UNROLL for (int i = 0; i < 3; ++i) sx.pt[k][i] = S(0); }
    DPLL_UNROLL for (int i = 0; i < 3; ++i) v[i] = sx.pt[0][i];
  }
  bool inside = false, separated = false;
  S scale2 = dot3(v, v);  // the size of C, for relative tolerances
  int it = 0;
  for (; it < kGjkMaxIter; ++it) {
    const S vv = dot3(v, v);
    if (!(vv > S(0))) { inside = true; break; }  // the origin is ON the simplex: touching, handled as overlapping
    const S wdir[3] = {-v[0], -v[1], -v[2]};
    S c[3];
    int ia, ib;
    minkowski_support<S, Lanes>(hp, wdir, c, ia, ib);
    scale2 = tmax(scale2, dot3(c, c));
    // discrete termination: the support vertex is already in the simplex -- v IS the closest point of C -- or no vertex
    // of C lies beyond the supporting plane at v at all
    bool have = false;
    DPLL_UNROLL for (int k = 0; k < 4; ++k) have = have || (k < sx.n && sx.ia[k] == ia && sx.ib[k] == ib);
    if (have || !(vv - dot3(v, c) > S(0))) { separated = true; break; }
    GjkSimplex<S> trial = sx;
    DPLL_UNROLL for (int k = 0; k < 4; ++k)
      if (k == trial.n) {
        DPLL_UNROLL for (int i = 0; i < 3; ++i) trial.pt[k][i] = c[i];
        trial.ia[k] = ia; trial.ib[k] = ib;
      }
    trial.n += 1;
    S vt[3];
    if (simplex_closest(trial, vt)) { sx = trial; inside = true; break; }
    // no progress (the new vertex ties with the supporting ones up to rounding: several vertices on one facet of C): v stands
    if (!(dot3(vt, vt) < vv)) { separated = true; break; }
    sx = trial;
    DPLL_UNROLL for (int i = 0; i < 3; ++i) v[i] = vt[i];
  }
  out.gjk_iters = it;
  if (!inside && !separated) { out.status = 1; separated = true; }  // out of iterations: the best point so far
  if (separated) {
    const S vv = dot3(v, v);
    if (vv > S(0)) {
      const S inv = S(1) / tsqrt(vv);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) out.d[i] = v[i] * inv;
      out.sep = vv * inv;
      return;
    }
    inside = true;
  }
  // ---- EPA: the facet of C closest to the origin --------------------------------------------------------------------
  if (sx.n < 4) {
    // touching configurations end GJK with the origin on a vertex / edge / triangle of C; blow the simplex up to a
    // tetrahedron with supports along directions that leave its affine hull
    for (int grow = 0; grow < 6 && sx.n < 4; ++grow) {
      S dir[3] = {S(0), S(0), S(0)};
      if (sx.n == 1) {
        dir[grow % 3] = (grow < 3) ? S(1) : S(-1);
      } else if (sx.n == 2) {
        const S e[3] = {sx.pt[1][0] - sx.pt[0][0], sx.pt[1][1] - sx.pt[0][1], sx.pt[1][2] - sx.pt[0][2]};
        S ax[3] = {S(0), S(0), S(0)};
        ax[grow % 3] = S(1);
        cross(e, ax, dir);
        if (grow >= 3) { dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
      } else {
        const S e1[3] = {sx.pt[1][0] - sx.pt[0][0], sx.pt[1][1] - sx.pt[0][1], sx.pt[1][2] - sx.pt[0][2]};
        const S e2[3] = {sx.pt[2][0] - sx.pt[0][0], sx.pt[2][1] - sx.pt[0][1], sx.pt[2][2] - sx.pt[0][2]};
        cross(e1, e2, dir);
        if (grow & 1) { dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
      }
      if (!(dot3(dir, dir) > S(0))) continue;
      S c[3];
      int ia, ib;
      minkowski_support<S, Lanes>(hp, dir, c, ia, ib);
      bool have = false;
      DPLL_UNROLL for (int k = 0; k < 4; ++k) have = have || (k < sx.n && sx.ia[k] == ia && sx.ib[k] == ib);
      if (have) continue;
      // must leave the affine hull of what is there
      bool independent = true;
      if (sx.n == 2) {
        const S e[3] = {sx.pt[1][0] - sx.pt[0][0], sx.pt[1][1] - sx.pt[0][1], sx.pt[1][2] - sx.pt[0][2]};
        const S f[3] = {c[0] - sx.pt[0][0], c[1] - sx.pt[0][1], c[2] - sx.pt[0][2]};
        S x[3];
        cross(e, f, x);
        independent = dot3(x, x) > S(1e-24) * scale2 * scale2;
      } else if (sx.n == 3) {
        const S e1[3] = {sx.pt[1][0] - sx.pt[0][0], sx.pt[1][1] - sx.pt[0][1], sx.pt[1][2] - sx.pt[0][2]};
        const S e2[3] = {sx.pt[2][0] - sx.pt[0][0], sx.pt[2][1] - sx.pt[0][1], sx.pt[2][2] - sx.pt[0][2]};
        const S f[3] = {c[0] - sx.pt[0][0], c[1] - sx.pt[0][1], c[2] - sx.pt[0][2]};
        S x[3];
        cross(e1, e2, x);
        const S vol = dot3(x, f);
        independent = vol * vol > S(1e-36) * scale2 * scale2 * scale2;
      }
      if (!independent) continue;
      DPLL_UNROLL for (int k = 0; k < 4; ++k)
        if (k == sx.n) {
          DPLL_UNROLL for (int i = 0; i < 3; ++i) sx.pt[k][i] = c[i];
          sx.ia[k] = ia; sx.ib[k] = ib;
        }
      sx.n += 1;
    }
    if (sx.n < 4) { out.status = 3; return; }
  }
  int nv = 4, nf = 4;
  DPLL_UNROLL for (int k = 0; k < 4; ++k) {
    DPLL_UNROLL for (int i = 0; i < 3; ++i) st.vx[k][i] = sx.pt[k][i];
    st.via[k] = (short)sx.ia[k]; st.vib[k] = (short)sx.ib[k];
  }
  {
    // orientation: faces counter-clockwise seen from outside
    const S e1[3] = {st.vx[1][0] - st.vx[0][0], st.vx[1][1] - st.vx[0][1], st.vx[1][2] - st.vx[0][2]};
    const S e2[3] = {st.vx[2][0] - st.vx[0][0], st.vx[2][1] - st.vx[0][1], st.vx[2][2] - st.vx[0][2]};
    const S e3[3] = {st.vx[3][0] - st.vx[0][0], st.vx[3][1] - st.vx[0][1], st.vx[3][2] - st.vx[0][2]};
    S x[3];
    cross(e1, e2, x);
    const bool pos = dot3(x, e3) > S(0);  // vertex 3 on the positive side of (0, 1, 2): that face must be flipped
    if (pos) { epa_set_face(st, 0, 0, 2, 1); epa_set_face(st, 1, 0, 1, 3); epa_set_face(st, 2, 1, 2, 3); epa_set_face(st, 3, 2, 0, 3); }
    else { epa_set_face(st, 0, 0, 1, 2); epa_set_face(st, 1, 0, 3, 1); epa_set_face(st, 2, 1, 3, 2); epa_set_face(st, 3, 2, 3, 0); }
  }
  const S tol = S(1e-13) * tsqrt(scale2);
  int best_f = 0;
  int eit = 0;
  bool done = false;
  for (; eit < 4 * kEpaMaxVerts && !done; ++eit) {
    // the live face closest to the origin
    S bd = S(3.0e38);
    best_f = -1;
    for (int f = 0; f < nf; ++f) {
      const S d = st.fd[f];
      if (d >= S(0) && d < bd) { bd = d; best_f = f; }
    }
    if (best_f < 0) { out.status = 2; break; }
    const S n[3] = {st.fn[best_f][0], st.fn[best_f][1], st.fn[best_f][2]};
    S c[3];
    int ia, ib;
    minkowski_support<S, Lanes>(hp, n, c, ia, ib);
    // is the support vertex part of the polytope already?
    int known = -1;
    for (int u = 0; u < nv; ++u)
      if (st.via[u] == (short)ia && st.vib[u] == (short)ib) { known = u; break; }
    const bool on_face = known >= 0 && (known == st.tri[best_f][0] || known == st.tri[best_f][1] || known == st.tri[best_f][2]);
    const S gap = dot3(n, c) - bd;
    if (on_face || !(gap > tol)) { done = true; break; }  // the face's plane supports C: a facet of C
    if (known >= 0) { st.fd[best_f] = S(-1); continue; }  // an inconsistent face (its normal is rounding noise): drop it
    if (nv >= kEpaMaxVerts) { out.status = 2; break; }
    // new vertex; faces that see it go, the horizon they leave gets new faces
    DPLL_UNROLL for (int i = 0; i < 3; ++i) st.vx[nv][i] = c[i];
    st.via[nv] = (short)ia; st.vib[nv] = (short)ib;
    int nh = 0;
    bool overflow = false;
    for (int f = 0; f < nf; ++f) {
      if (!(st.fd[f] >= S(0))) continue;
      const S fnn[3] = {st.fn[f][0], st.fn[f][1], st.fn[f][2]};
      const bool visible = (f == best_f) || (dot3(fnn, c) - st.fd[f] > tol);
      if (!visible) continue;
      st.fd[f] = S(-1);
      DPLL_UNROLL for (int e = 0; e < 3; ++e) {
        const unsigned char a = st.tri[f][e], b = st.tri[f][(e + 1) % 3];
        int found = -1;
        for (int h = 0; h < nh; ++h)
          if (st.hz[h][0] == b && st.hz[h][1] == a) { found = h; break; }
        if (found >= 0) {  // shared with another visible face: not on the horizon
          st.hz[found][0] = st.hz[nh - 1][0]; st.hz[found][1] = st.hz[nh - 1][1];
          --nh;
        } else if (nh < kEpaMaxHorizon) {
          st.hz[nh][0] = a; st.hz[nh][1] = b;
          ++nh;
        } else {
          overflow = true;
        }
      }
    }
    if (overflow) { out.status = 2; break; }
    // new faces into dead slots first, then at the end
    int slot = 0;
    for (int h = 0; h < nh; ++h) {
      while (slot < nf && st.fd[slot] >= S(0)) ++slot;
      if (slot >= kEpaMaxFaces) { overflow = true; break; }
      epa_set_face(st, slot, st.hz[h][0], st.hz[h][1], nv);
      if (slot >= nf) nf = slot + 1;
      ++slot;
    }
    ++nv;
    if (overflow) { out.status = 2; break; }
  }
  out.epa_iters = eit;
  if (!done && out.status == 0) out.status = 2;
  if (best_f >= 0) {
    DPLL_UNROLL for (int i = 0; i < 3; ++i) out.d[i] = -st.fn[best_f][i];
    out.sep = -st.fd[best_f];
  }
}

}  // namespace dpll
