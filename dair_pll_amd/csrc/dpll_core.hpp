// dpll_core.hpp -- per-trajectory-item math of the contact-dynamics hot path.
//
// Everything here is straight-line, fully unrolled, register-resident arithmetic for ONE batch item,
// written so that a group of G lanes (one lane per contact on the GPU, G = 1 in the host-side
// sanitizer build under tests/hostsim) cooperates on the item: per-contact quantities live in the
// lane that owns the contact, the small dense blocks (M, Hessian, Cholesky factors; n_v = 6..8) are
// replicated across the group, and sums over contacts go through `Lanes::group_sum`.
//
// What is computed follows dair_pll (file:line under /root/reference/dair_pll):
//   * MultibodyTerms.forward                multibody_terms.py:584-609
//   * LagrangianTerms.forward               multibody_terms.py:214-237 (M, M^-1 F; definitions :123-157)
//   * ContactTerms.forward                  multibody_terms.py:428-521 (phi, J; plane-vs-box geometry.py:554-582)
//   * InertialParameterConverter            inertia.py:206-234, 305-331, 377-382
//   * contactnets_loss                      multibody_learnable_system.py:104-197
//   * forward_dynamics                      multibody_learnable_system.py:199-304
//   * VelocityIntegrator.step / exponential integrator.py:153-162, state_space.py:466-486, quaternion.py:89-147,276-309
// The cone QP that dair_pll delegates to sappy.SAPSolver is solved here by a semi-smooth Newton
// method on its unconstrained primal, in generalized-velocity coordinates (see sap_newton).
//
// The design does NOT mirror the reference's tensor graph: there is no D = J M^-1 J^T, no M^-1, no
// J matrix in memory; the backward pass is a hand-derived adjoint that emits gradients with respect
// to the per-body inertial 10-vectors, the pair friction coefficients and the box half lengths, which
// are reduced over the batch on chip and chained to the learnable parameters once per launch.
#pragma once

#include <math.h>
#if defined(DPLL_TRACE)
#include <cstdio>
#endif
#include <stdint.h>

#if defined(__HIPCC__)
#define DPLL_HD __host__ __device__ __forceinline__
#else
#define DPLL_HD inline __attribute__((always_inline))
#endif

// a real call instead of an inlined copy in every kernel: the body-body direction search (big, rare, off the hot path)
#if defined(__HIPCC__)
#define DPLL_HD_CALL __host__ __device__ __attribute__((noinline))
#else
#define DPLL_HD_CALL inline __attribute__((noinline))
#endif

#define DPLL_UNROLL _Pragma("unroll")
#if defined(__clang__)
#define DPLL_NOUNROLL _Pragma("clang loop unroll(disable)")
#else
#define DPLL_NOUNROLL _Pragma("GCC unroll 1")
#endif
#ifndef DPLL_INCREMENTAL
#define DPLL_INCREMENTAL 1
#endif
#ifndef DPLL_PHASE_BEGIN
#define DPLL_PHASE_BEGIN() do {} while (0)
#define DPLL_PHASE(slot) do {} while (0)
#define DPLL_PHASE_END() do {} while (0)
#define DPLL_PHASE_COUNT(slot) do {} while (0)
#define DPLL_PHASE_EVENT(slot, happened) do {} while (0)
#endif
#ifndef DPLL_CORE_STAMP
#define DPLL_CORE_STAMP(slot) do {} while (0)
#endif
#ifndef DPLL_ITER_HOOK  // host-side statistics of the solver (tests/hostsim): (iteration, item still active, step length taken)
#define DPLL_ITER_HOOK(it, active, alpha) do {} while (0)
#endif

namespace dpll {

constexpr int kMaxJoints = 3;   // revolute joints of the tree hanging off the floating base
constexpr int kMaxBodies = kMaxJoints + 1;
constexpr int kMaxGeoms = 3;    // convex collision geometries of a model (each against the ground half-space)
constexpr int kGeomBox = 0, kGeomSphere = 1, kGeomPolygon = 2;
// a learned convex shape (DeepSupportConvex, geometry.py:255-364): its support points come from the ICNN kernels as
// `witness` inputs; the general build with mesh geometry (csrc/dpll_genmesh.hip)
constexpr int kGeomMesh = 3;
constexpr int kMaxPolyVerts = 8;  // vertices of a Polygon (geometry.py:220-252); the general build only
constexpr int kMaxPairs = 4;      // body-body collision candidates (geometry.py:585-643); the general build only
// geometry slots of the general build: behind the geometries one more group of kQuery contact slots, one per pair (a pair
// makes ONE contact, geometry.py:639-643)
constexpr int kGenSlots = kMaxGeoms + 1;
static_assert(kMaxPairs <= 4, "the pairs share the kQuery contact slots of one group");
constexpr int kJointRevolute = 0, kJointPrismatic = 1;
constexpr int kQuery = 4;       // witness points per convex geometry (geometry.py:47-48)
constexpr int kIota = 10;       // per-body inertial vector [m, h = m c (3), I_o (xx,yy,zz,xy,xz,yz)]
// continuation stages a build with racing copies can start from: its starting regularisation eps * factor^(stages - 1) is
// formed by a loop unrolled this many times (no divergent loop per lane); dpll_model_set_solver refuses longer schedules --
// n_stages, loss_n_stages, race_stages -- unless the copies are switched off (portfolio = 1)
constexpr int kRaceMaxStages = 8;

#ifndef DPLL_PRISMATIC  // (diagnostic builds define it to 0: every joint a hinge at compile time)
#define DPLL_PRISMATIC 1
#endif
#ifndef DPLL_ROTATED  // (diagnostic builds define it to 0: the code for turned frames compiled out)
#define DPLL_ROTATED(md) ((md).rotated)
#endif
// Plain-old-data model description, passed to kernels by value.
struct ModelDesc {
  int32_t n_joints;
  int32_t inertia_mode;  // 0: reference_literal (rotational inertia taken as I_cm / m), 1: physical
  double dt;
  double gravity_z;
  double joint_origin[kMaxJoints][3];  // joint j+1 frame origin in the parent body frame
  double joint_axis[kMaxJoints][3];    // unit axis, same in parent and child frames
  double geom_origin[kMaxGeoms][3];    // geometry g: its origin in the frame of its body
  // General models only (n_geoms > 0; MultibodyTerms handles any tree and any number of geometries,
  // multibody_terms.py:328-382, drake_utils.py:309-335).  The two fast builds (cube, elbow) leave these zero and mean: a
  // serial chain, one box per body, geometry g on body g.
  int32_t parent[kMaxJoints];          // parent body of body j + 1 (< j + 1)
  int32_t n_geoms;
  int32_t geom_body[kMaxGeoms];
  int32_t geom_kind[kMaxGeoms];        // kGeomBox | kGeomSphere | kGeomPolygon
  int32_t geom_nverts[kMaxGeoms];      // Polygon: number of vertices, 4 .. kMaxPolyVerts
  // body-body collision candidates (ContactTerms.collision_candidates beyond the ground pairs, multibody_terms.py:286-297):
  // geometry pair_a[p] against geometry pair_b[p], ordered as the reference orders a pair (geometry.py:46, 66-74)
  int32_t n_pairs;
  int32_t pair_a[kMaxPairs];
  int32_t pair_b[kMaxPairs];
  // frames turned against each other (URDF rpy; general build only, see include/dpll.h): bit 0 = body_rot, bit 1 = geom_rot
  int32_t rotated;
  double body_rot[kMaxBodies][3][3];  // inertial parameters' frame -> the kernels' frame of the body
  double geom_rot[kMaxGeoms][3][3];   // geometry frame in the kernels' frame of its body; geom_origin is in the geometry frame
  int32_t joint_kind[kMaxJoints];     // kJointRevolute | kJointPrismatic (general build; the fast builds: revolute)
  int32_t reserved;
  static constexpr bool kGeneral = false;
  static constexpr int kGeoStride = 3;  // numbers per geometry in the `lengths` parameter block: a box's length_params
};
// same layout; selects the tree / geometry-table code paths at compile time
struct GeneralDesc : ModelDesc {
  static constexpr bool kGeneral = true;
  // box: length_params (3) | sphere: length_param (1) | polygon: vertices (n_verts, 3) row-major; the rest padding
  static constexpr int kGeoStride = 3 * kMaxPolyVerts;
};

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
template <typename T> DPLL_HD T tsqrt(T x) { return sqrt(x); }
template <> DPLL_HD float tsqrt<float>(float x) { return sqrtf(x); }
template <typename T> DPLL_HD T tabs(T x) { return x < T(0) ? -x : x; }
template <typename T> DPLL_HD T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> DPLL_HD T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> DPLL_HD T texp(T x) { return exp(x); }
template <> DPLL_HD float texp<float>(float x) { return expf(x); }
template <typename T> DPLL_HD void tsincos(T x, T& s, T& c) { s = sin(x); c = cos(x); }
template <> DPLL_HD void tsincos<float>(float x, float& s, float& c) { s = sinf(x); c = cosf(x); }

DPLL_HD bool bad_number(float x) {
  uint32_t u; __builtin_memcpy(&u, &x, 4);
  return (u & 0x7f800000u) == 0x7f800000u;  // inf or nan
}
DPLL_HD bool bad_number(double x) {
  uint64_t u; __builtin_memcpy(&u, &x, 8);
  return (u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
}

// Reciprocal / reciprocal square root.  float on the GPU: the 1-ulp hardware approximations
// (v_rcp_f32 / v_rsq_f32), enough for a self-correcting Newton iteration whose answer is judged to 1e-4;
// double and the host build: exact division / sqrt.
template <typename T> DPLL_HD T fast_rcp(T x) { return T(1) / x; }
template <typename T> DPLL_HD T fast_rsqrt(T x) { return T(1) / tsqrt(x); }
template <typename T> DPLL_HD T fast_sqrt(T x) { return tsqrt(x); }
#if defined(__HIP_DEVICE_COMPILE__)
template <> DPLL_HD float fast_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <> DPLL_HD float fast_rsqrt<float>(float x) { return __builtin_amdgcn_rsqf(x); }
// (v_sqrt_f32, 1 ulp, instead of sqrtf's correctly rounded sequence of a dozen instructions: for quantities that only scale a tolerance)
template <> DPLL_HD float fast_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
#endif

template <typename T> DPLL_HD void cross(const T (&a)[3], const T (&b)[3], T (&c)[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T> DPLL_HD T dot3(const T (&a)[3], const T (&b)[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// y = A x, y = A^T x for 3x3
template <typename T> DPLL_HD void mat3_vec(const T (&A)[3][3], const T (&x)[3], T (&y)[3]) {
  DPLL_UNROLL for (int i = 0; i < 3; ++i) y[i] = A[i][0] * x[0] + A[i][1] * x[1] + A[i][2] * x[2];
}
template <typename T> DPLL_HD void mat3t_vec(const T (&A)[3][3], const T (&x)[3], T (&y)[3]) {
  DPLL_UNROLL for (int i = 0; i < 3; ++i) y[i] = A[0][i] * x[0] + A[1][i] * x[1] + A[2][i] * x[2];
}
template <typename T> DPLL_HD void mat3_mul(const T (&A)[3][3], const T (&B)[3][3], T (&C)[3][3]) {
  DPLL_UNROLL for (int i = 0; i < 3; ++i)
    DPLL_UNROLL for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}
// symmetric 3x3 stored as (xx,yy,zz,xy,xz,yz) times vector
template <typename T> DPLL_HD void sym3_vec(const T* s, const T (&x)[3], T (&y)[3]) {
  y[0] = s[0] * x[0] + s[3] * x[1] + s[4] * x[2];
  y[1] = s[3] * x[0] + s[1] * x[1] + s[5] * x[2];
  y[2] = s[4] * x[0] + s[5] * x[1] + s[2] * x[2];
}

// quaternion.rotate as a matrix: homogeneous quadratic in the quaternion, NOT normalised
// (quaternion.py:150-164; quirk Q2 in DESIGN.md).
template <typename T> DPLL_HD void quat_to_rot(const T* q, T (&R)[3][3]) {
  const T w = q[0], x = q[1], y = q[2], z = q[3];
  const T ww = w * w, xx = x * x, yy = y * y, zz = z * z;
  R[0][0] = ww + xx - yy - zz; R[0][1] = T(2) * (x * y - w * z); R[0][2] = T(2) * (x * z + w * y);
  R[1][0] = T(2) * (x * y + w * z); R[1][1] = ww - xx + yy - zz; R[1][2] = T(2) * (y * z - w * x);
  R[2][0] = T(2) * (x * z - w * y); R[2][1] = T(2) * (y * z + w * x); R[2][2] = ww - xx - yy + zz;
}

// Rodrigues rotation about a unit axis.
template <typename T> DPLL_HD void axis_rot(const T (&k)[3], T angle, T (&R)[3][3]) {
  T s, c; tsincos(angle, s, c);
  const T v = T(1) - c;
  R[0][0] = c + k[0] * k[0] * v;        R[0][1] = k[0] * k[1] * v - k[2] * s; R[0][2] = k[0] * k[2] * v + k[1] * s;
  R[1][0] = k[1] * k[0] * v + k[2] * s; R[1][1] = c + k[1] * k[1] * v;        R[1][2] = k[1] * k[2] * v - k[0] * s;
  R[2][0] = k[2] * k[0] * v - k[1] * s; R[2][1] = k[2] * k[1] * v + k[0] * s; R[2][2] = c + k[2] * k[2] * v;
}

// ---------------------------------------------------------------------------------------------
// inertial parameterisation: theta (log-Cholesky, 10) -> iota = [m, h = m c, I_o_eff] about the
// body origin.  Generic in the scalar type so that the finalize kernel can push dual numbers
// through it.  inertia.py:206-234 (theta_to_pi_o), :305-331 (pi_o_to_pi_cm), :377-382 (I_cm / m).
// ---------------------------------------------------------------------------------------------
// forward-mode dual number: value + one directional derivative
template <typename T> struct DualT {
  T v, d;
  DPLL_HD DualT() : v(0), d(0) {}
  DPLL_HD DualT(T a) : v(a), d(0) {}
  DPLL_HD DualT(T a, T b) : v(a), d(b) {}
};
template <typename T> DPLL_HD DualT<T> operator+(const DualT<T>& a, const DualT<T>& b) { return DualT<T>(a.v + b.v, a.d + b.d); }
template <typename T> DPLL_HD DualT<T> operator-(const DualT<T>& a, const DualT<T>& b) { return DualT<T>(a.v - b.v, a.d - b.d); }
template <typename T> DPLL_HD DualT<T> operator*(const DualT<T>& a, const DualT<T>& b) { return DualT<T>(a.v * b.v, a.d * b.v + a.v * b.d); }
template <typename T> DPLL_HD DualT<T> operator/(const DualT<T>& a, const DualT<T>& b) {
  const T q = a.v / b.v;
  return DualT<T>(q, (a.d - q * b.d) / b.v);
}
template <typename T> DPLL_HD DualT<T> operator-(const DualT<T>& a) { return DualT<T>(-a.v, -a.d); }
template <typename T> DPLL_HD DualT<T>& operator+=(DualT<T>& a, const DualT<T>& b) { a.v += b.v; a.d += b.d; return a; }
template <typename T> DPLL_HD DualT<T>& operator-=(DualT<T>& a, const DualT<T>& b) { a.v -= b.v; a.d -= b.d; return a; }
template <typename T> DPLL_HD DualT<T>& operator*=(DualT<T>& a, const DualT<T>& b) { a = a * b; return a; }
// comparisons look at the value only: branches of the primal computation are piecewise constant in the seed
template <typename T> DPLL_HD bool operator<(const DualT<T>& a, const DualT<T>& b) { return a.v < b.v; }
template <typename T> DPLL_HD bool operator>(const DualT<T>& a, const DualT<T>& b) { return a.v > b.v; }
template <typename T> DPLL_HD bool operator<=(const DualT<T>& a, const DualT<T>& b) { return a.v <= b.v; }
template <typename T> DPLL_HD bool operator>=(const DualT<T>& a, const DualT<T>& b) { return a.v >= b.v; }
template <typename T> DPLL_HD DualT<T> tsqrt(DualT<T> x) {
  const T r = tsqrt(x.v);
  return DualT<T>(r, x.d == T(0) ? T(0) : x.d / (r + r));  // a constant keeps derivative 0 even at sqrt(0)
}
template <typename T> DPLL_HD void tsincos(DualT<T> x, DualT<T>& s, DualT<T>& c) {
  T sv, cv;
  tsincos(x.v, sv, cv);
  s = DualT<T>(sv, cv * x.d);
  c = DualT<T>(cv, -sv * x.d);
}
DPLL_HD float s_exp(const float& x) { return expf(x); }
DPLL_HD double s_exp(const double& x) { return exp(x); }
template <typename T> DPLL_HD DualT<T> s_exp(const DualT<T>& x) { const T e = s_exp(x.v); return DualT<T>(e, e * x.d); }

template <typename S> DPLL_HD void theta_to_iota(const S (&th)[10], int inertia_mode, S (&iota)[kIota]) {
  const S &alpha = th[0], &d1 = th[1], &d2 = th[2], &d3 = th[3], &s12 = th[4], &s23 = th[5], &s13 = th[6],
          &t1 = th[7], &t2 = th[8], &t3 = th[9];
  const S e1 = s_exp(d1), e2 = s_exp(d2), e3 = s_exp(d3);
  const S sc = s_exp(alpha + alpha);
  // pi_o = [m, m c, I_o(xx,yy,zz,xy,xz,yz)]
  const S m = sc * (t1 * t1 + t2 * t2 + t3 * t3 + S(1));
  const S h0 = sc * (t1 * e1), h1 = sc * (t1 * s12 + t2 * e2), h2 = sc * (t1 * s13 + t2 * s23 + t3 * e3);
  const S oxx = sc * (s12 * s12 + s23 * s23 + s13 * s13 + e2 * e2 + e3 * e3);
  const S oyy = sc * (s13 * s13 + s23 * s23 + e1 * e1 + e3 * e3);
  const S ozz = sc * (s12 * s12 + e1 * e1 + e2 * e2);
  const S oxy = sc * (S(0) - s12 * e1), oxz = sc * (S(0) - s13 * e1), oyz = sc * (S(0) - s12 * s13 - s23 * e2);
  iota[0] = m; iota[1] = h0; iota[2] = h1; iota[3] = h2;
  if (inertia_mode == 1) {  // physical: rotational inertia about the origin is pi_o's own
    iota[4] = oxx; iota[5] = oyy; iota[6] = ozz; iota[7] = oxy; iota[8] = oxz; iota[9] = oyz;
    return;
  }
  // reference_literal: central inertia I_cm = I_o + m S(c)^2 is divided by m, then shifted back to
  // the origin with the true mass: I_eff = I_cm / m - m S(c)^2 = I_o / m + (1 - m) S(c)^2 ... written
  // out with S(c)^2 = c c^T - |c|^2 1.
  const S c0 = h0 / m, c1 = h1 / m, c2 = h2 / m;
  const S cc = c0 * c0 + c1 * c1 + c2 * c2;
  const S k = S(1) - m;  // coefficient of S(c)^2
  const S im = S(1) / m;
  iota[4] = oxx * im + k * (c0 * c0 - cc);
  iota[5] = oyy * im + k * (c1 * c1 - cc);
  iota[6] = ozz * im + k * (c2 * c2 - cc);
  iota[7] = oxy * im + k * (c0 * c1);
  iota[8] = oxz * im + k * (c0 * c2);
  iota[9] = oyz * im + k * (c1 * c2);
}

// the inertial vector of a body taken to a frame turned by A (coordinates v -> A v):  h -> A h,  I_o -> A I_o A^T
template <typename S> DPLL_HD void rotate_iota(const double (&A)[3][3], S (&iota)[kIota]) {
  const S h[3] = {iota[1], iota[2], iota[3]};
  const S I[3][3] = {{iota[4], iota[7], iota[8]}, {iota[7], iota[5], iota[9]}, {iota[8], iota[9], iota[6]}};
  S AI[3][3];
  DPLL_UNROLL for (int r = 0; r < 3; ++r) {
    iota[1 + r] = S(A[r][0]) * h[0] + S(A[r][1]) * h[1] + S(A[r][2]) * h[2];
    DPLL_UNROLL for (int c = 0; c < 3; ++c) AI[r][c] = S(A[r][0]) * I[0][c] + S(A[r][1]) * I[1][c] + S(A[r][2]) * I[2][c];
  }
  auto out = [&](int r, int c) { return AI[r][0] * S(A[c][0]) + AI[r][1] * S(A[c][1]) + AI[r][2] * S(A[c][2]); };
  iota[4] = out(0, 0); iota[5] = out(1, 1); iota[6] = out(2, 2);
  iota[7] = out(0, 1); iota[8] = out(0, 2); iota[9] = out(1, 2);
}
// R <- R G: a body's world rotation taken to the frame of a geometry that sits turned in the body
template <typename S> DPLL_HD void mat3_mul_const(S (&R)[3][3], const double (&G)[3][3]) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) {
    const S a = R[r][0], b = R[r][1], c = R[r][2];
    DPLL_UNROLL for (int k = 0; k < 3; ++k) R[r][k] = a * S(G[0][k]) + b * S(G[1][k]) + c * S(G[2][k]);
  }
}

// A geometry turned in its body: R (the body's world rotation) becomes the geometry's, R_WB R_BG, and so does its third
// row Rz.  (Inlined: as a real call it was no cheaper.  Models without turned frames skip it on a uniform branch; its
// presence alone costs the double-precision general kernels of 2-joint models ~10 %, tools/diag/time_general.py with
// -DDPLL_ROTATED(md)=0 as the other build.)
template <typename T, typename TA>
DPLL_HD void turn_to_geometry(const double (&geom_rot)[kMaxGeoms][3][3], int g, T (&R)[3][3], TA (&Rz)[3]) {
  double G[3][3];
  DPLL_UNROLL for (int r = 0; r < 3; ++r)
    DPLL_UNROLL for (int c = 0; c < 3; ++c) {
      G[r][c] = geom_rot[0][r][c];
      DPLL_UNROLL for (int gg = 1; gg < kMaxGeoms; ++gg) G[r][c] = (g == gg) ? geom_rot[gg][r][c] : G[r][c];
    }
  mat3_mul_const<T>(R, G);
  const TA z0 = Rz[0], z1 = Rz[1], z2 = Rz[2];
  DPLL_UNROLL for (int k = 0; k < 3; ++k) Rz[k] = z0 * TA(G[0][k]) + z1 * TA(G[1][k]) + z2 * TA(G[2][k]);
}

// spatial inertia applied to a motion vector (w, u):  n = I_o w + h x u,  f = m u - h x w
template <typename T>
DPLL_HD void inertia_apply(const T (&io)[kIota], const T (&w)[3], const T (&u)[3], T (&n)[3], T (&f)[3]) {
  const T h[3] = {io[1], io[2], io[3]};
  T hw[3], hu[3];
  cross(h, w, hw);
  cross(h, u, hu);
  sym3_vec(&io[4], w, n);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { n[i] += hu[i]; f[i] = io[0] * u[i] - hw[i]; }
}

// gradient of coef * Y^T I6 Z with respect to iota, accumulated into g
template <typename T>
DPLL_HD void inertia_bilinear_grad(T coef, const T (&yw)[3], const T (&yu)[3], const T (&zw)[3], const T (&zu)[3],
                                   T (&g)[kIota]) {
  T a[3], b[3];
  cross(zu, yw, a);
  cross(yu, zw, b);
  g[0] += coef * dot3(yu, zu);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) g[1 + i] += coef * (a[i] + b[i]);
  g[4] += coef * (yw[0] * zw[0]);
  g[5] += coef * (yw[1] * zw[1]);
  g[6] += coef * (yw[2] * zw[2]);
  g[7] += coef * (yw[0] * zw[1] + yw[1] * zw[0]);
  g[8] += coef * (yw[0] * zw[2] + yw[2] * zw[0]);
  g[9] += coef * (yw[1] * zw[2] + yw[2] * zw[1]);
}

// ---------------------------------------------------------------------------------------------
// chain kinematics
// ---------------------------------------------------------------------------------------------
template <typename T, int NJ> struct Kin {
  static constexpr int NB = NJ + 1;
  T R[NB][3][3];    // world <- body
  T o[NB][3];       // body origin in the world
  T Rpc[NB][3][3];  // parent <- child (index >= 1)
  T pj[NB][3];      // joint origin in the parent frame (index >= 1)
  T ax[NB][3];      // joint axis in body coordinates (index >= 1)
  T axw[NB][3];     // joint axis in the world (index >= 1)
  bool pris[NB];    // prismatic: the body slides along the axis (pj then includes the travel); else it turns about it
  int par[NB];      // parent body (serial chain: j - 1, known at compile time; general models: from the description)
  bool anc[NB][NB]; // anc[b][j]: joint j (the joint of body j >= 1) lies between the base and body b
};

// arr[idx] for idx < upto, without run-time indexing (register arrays): the selects fold when idx is a constant
template <typename T, int N> DPLL_HD void pick3(const T (&arr)[N][3], int idx, int upto, T (&out)[3]) {
  DPLL_UNROLL for (int i = 0; i < 3; ++i) out[i] = arr[0][i];
  DPLL_UNROLL for (int c = 1; c < N; ++c)
    if (c < upto) {
      const bool pick = (idx == c);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) out[i] = pick ? arr[c][i] : out[i];
    }
}
template <typename T, int N> DPLL_HD void pick33(const T (&arr)[N][3][3], int idx, int upto, T (&out)[3][3]) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r)
    DPLL_UNROLL for (int i = 0; i < 3; ++i) out[r][i] = arr[0][r][i];
  DPLL_UNROLL for (int c = 1; c < N; ++c)
    if (c < upto) {
      const bool pick = (idx == c);
      DPLL_UNROLL for (int r = 0; r < 3; ++r)
        DPLL_UNROLL for (int i = 0; i < 3; ++i) out[r][i] = pick ? arr[c][r][i] : out[r][i];
    }
}
// does joint jj move body b (b may differ from lane to lane)?
template <typename T, int NJ> DPLL_HD bool joint_moves(const Kin<T, NJ>& k, int jj, int b) {
  bool yes = false;
  DPLL_UNROLL for (int bb = 1; bb <= NJ; ++bb)
    if (bb >= jj) yes = yes || (b == bb && k.anc[bb][jj]);
  return yes;
}

template <typename T, int NJ, class MD> DPLL_HD void kinematics(const MD& md, const T* q, Kin<T, NJ>& k) {
  quat_to_rot(q, k.R[0]);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) k.o[0][i] = q[4 + i];
  k.par[0] = 0;
  k.pris[0] = false;
  DPLL_UNROLL for (int b = 0; b <= NJ; ++b)
    DPLL_UNROLL for (int j = 0; j <= NJ; ++j) k.anc[b][j] = false;
  DPLL_UNROLL for (int j = 1; j <= NJ; ++j) {
    int p = j - 1;
    if constexpr (MD::kGeneral) p = md.parent[j - 1];
    k.par[j] = p;
    k.anc[j][j] = true;
    DPLL_UNROLL for (int a = 1; a < j; ++a) {  // the parent's ancestors are mine
      bool up = false;
      DPLL_UNROLL for (int c = 1; c < j; ++c) up = up || (p == c && k.anc[c][a]);
      k.anc[j][a] = up;
    }
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { k.pj[j][i] = T(md.joint_origin[j - 1][i]); k.ax[j][i] = T(md.joint_axis[j - 1][i]); }
    k.pris[j] = false;
    if constexpr (MD::kGeneral) {
      // a prismatic joint: no turn (angle 0 gives the identity exactly), the child origin travels along the axis
      k.pris[j] = DPLL_PRISMATIC && md.joint_kind[j - 1] == kJointPrismatic;
      DPLL_UNROLL for (int i = 0; i < 3; ++i) k.pj[j][i] += k.pris[j] ? k.ax[j][i] * q[7 + j - 1] : T(0);
      axis_rot(k.ax[j], k.pris[j] ? T(0) : q[7 + j - 1], k.Rpc[j]);
    } else {
      axis_rot(k.ax[j], q[7 + j - 1], k.Rpc[j]);
    }
    T Rp[3][3], op[3];
    pick33(k.R, p, j, Rp);
    pick3(k.o, p, j, op);
    mat3_mul(Rp, k.Rpc[j], k.R[j]);
    T t[3];
    mat3_vec(Rp, k.pj[j], t);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) k.o[j][i] = op[i] + t[i];
    mat3_vec(k.R[j], k.ax[j], k.axw[j]);
  }
}

// Body-frame spatial velocities Y_b = S_b y (angular; linear at the body origin) for a generalized
// velocity y = [omega_body(3), v_world(3), joint rates]  (state_space.py:412-424).
template <typename T, int NJ>
DPLL_HD void body_twists(const Kin<T, NJ>& k, const T* y, T (&Yw)[NJ + 1][3], T (&Yu)[NJ + 1][3]) {
  const T vl[3] = {y[3], y[4], y[5]};
  DPLL_UNROLL for (int i = 0; i < 3; ++i) Yw[0][i] = y[i];
  mat3t_vec(k.R[0], vl, Yu[0]);
  DPLL_UNROLL for (int j = 1; j <= NJ; ++j) {
    T pw[3], pu[3], wxp[3], t[3];
    pick3(Yw, k.par[j], j, pw);
    pick3(Yu, k.par[j], j, pu);
    cross(pw, k.pj[j], wxp);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) t[i] = pu[i] + wxp[i];
    mat3t_vec(k.Rpc[j], t, Yu[j]);
    mat3t_vec(k.Rpc[j], pw, t);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) {  // joint column s_j = (axis, 0) revolute, (0, axis) prismatic
      const T sr = k.ax[j][i] * y[6 + j - 1];
      Yw[j][i] = t[i] + (k.pris[j] ? T(0) : sr);
      Yu[j][i] += k.pris[j] ? sr : T(0);
    }
  }
}

// Mass matrix by the composite-rigid-body recursion on the 10-vector representation
// (equals gamma^T M_drake gamma of multibody_terms.py:131).  Full symmetric NV x NV.
template <typename T, int NJ>
DPLL_HD void mass_matrix(const Kin<T, NJ>& k, const T (&iota)[NJ + 1][kIota], T (&M)[6 + NJ][6 + NJ]) {
  constexpr int NB = NJ + 1;
  T comp[NB][kIota];
  DPLL_UNROLL for (int b = 0; b < NB; ++b)
    DPLL_UNROLL for (int i = 0; i < kIota; ++i) comp[b][i] = iota[b][i];
  DPLL_UNROLL for (int j = NJ; j >= 1; --j) {
    // joint column: F = I^c_j s_j with s_j = (axis, 0), a prismatic joint's (0, axis)
    T n[3], f[3];
    const T zero[3] = {T(0), T(0), T(0)};
    if (k.pris[j]) inertia_apply(comp[j], zero, k.ax[j], n, f);
    else inertia_apply(comp[j], k.ax[j], zero, n, f);
    M[6 + j - 1][6 + j - 1] = k.pris[j] ? dot3(k.ax[j], f) : dot3(k.ax[j], n);
    DPLL_UNROLL for (int a = 1; a < j; ++a) { M[6 + a - 1][6 + j - 1] = T(0); M[6 + j - 1][6 + a - 1] = T(0); }
    int cur = j;  // the wrench is expressed in the frame of body `cur`; walk up through the ancestors only
    DPLL_UNROLL for (int a = j; a >= 1; --a) {
      const bool on = (a == cur);
      // transform the wrench from body a to its parent
      T rn[3], rf[3], pxf[3];
      mat3_vec(k.Rpc[a], n, rn);
      mat3_vec(k.Rpc[a], f, rf);
      cross(k.pj[a], rf, pxf);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) { n[i] = on ? rn[i] + pxf[i] : n[i]; f[i] = on ? rf[i] : f[i]; }
      cur = on ? k.par[a] : cur;
      DPLL_UNROLL for (int c = 1; c < a; ++c) {
        const bool hit = on && (cur == c);
        const T val = k.pris[c] ? dot3(k.ax[c], f) : dot3(k.ax[c], n);
        M[6 + c - 1][6 + j - 1] = hit ? val : M[6 + c - 1][6 + j - 1];
        M[6 + j - 1][6 + c - 1] = hit ? val : M[6 + j - 1][6 + c - 1];
      }
    }
    T wf[3];
    mat3_vec(k.R[0], f, wf);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) {
      M[i][6 + j - 1] = n[i]; M[6 + j - 1][i] = n[i];
      M[3 + i][6 + j - 1] = wf[i]; M[6 + j - 1][3 + i] = wf[i];
    }
    // fold body j's composite inertia into its parent: rotate, then shift the origin by pj
    const T hc_child[3] = {comp[j][1], comp[j][2], comp[j][3]};
    T hc[3];
    mat3_vec(k.Rpc[j], hc_child, hc);
    const T I[3][3] = {{comp[j][4], comp[j][7], comp[j][8]}, {comp[j][7], comp[j][5], comp[j][9]}, {comp[j][8], comp[j][9], comp[j][6]}};
    T RI[3][3], RIRt[3][3];
    mat3_mul(k.Rpc[j], I, RI);
    DPLL_UNROLL for (int r = 0; r < 3; ++r)
      DPLL_UNROLL for (int c = 0; c < 3; ++c)
        RIRt[r][c] = RI[r][0] * k.Rpc[j][c][0] + RI[r][1] * k.Rpc[j][c][1] + RI[r][2] * k.Rpc[j][c][2];
    const T m = comp[j][0];
    const T(&d)[3] = k.pj[j];
    // I' = I - m S(d)^2 - S(d) S(hc) - S(hc) S(d);  S(a) S(b) = b a^T - (a.b) 1
    const T dd = dot3(d, d), dh = dot3(d, hc);
    T Ip[3][3];
    DPLL_UNROLL for (int r = 0; r < 3; ++r)
      DPLL_UNROLL for (int c = 0; c < 3; ++c) {
        const T delta = (r == c) ? T(1) : T(0);
        Ip[r][c] = RIRt[r][c] - m * (d[r] * d[c] - dd * delta) - (hc[r] * d[c] + d[r] * hc[c] - T(2) * dh * delta);
      }
    const T add[kIota] = {m, hc[0] + m * d[0], hc[1] + m * d[1], hc[2] + m * d[2], Ip[0][0], Ip[1][1], Ip[2][2], Ip[0][1], Ip[0][2], Ip[1][2]};
    DPLL_UNROLL for (int c = 0; c < j; ++c) {
      const bool mine = (k.par[j] == c);
      DPLL_UNROLL for (int i = 0; i < kIota; ++i) comp[c][i] += mine ? add[i] : T(0);
    }
  }
  // base block [[I_o, S(h) R^T],[R S(h)^T, m 1]]
  const T(&c0)[kIota] = comp[0];
  M[0][0] = c0[4]; M[1][1] = c0[5]; M[2][2] = c0[6];
  M[0][1] = M[1][0] = c0[7]; M[0][2] = M[2][0] = c0[8]; M[1][2] = M[2][1] = c0[9];
  const T Sh[3][3] = {{T(0), -c0[3], c0[2]}, {c0[3], T(0), -c0[1]}, {-c0[2], c0[1], T(0)}};
  DPLL_UNROLL for (int r = 0; r < 3; ++r)
    DPLL_UNROLL for (int c = 0; c < 3; ++c) {
      // (S(h) R^T)[r][c] = sum_k Sh[r][k] R[c][k]
      const T val = Sh[r][0] * k.R[0][c][0] + Sh[r][1] * k.R[0][c][1] + Sh[r][2] * k.R[0][c][2];
      M[r][3 + c] = val; M[3 + c][r] = val;
      M[3 + r][3 + c] = (r == c) ? c0[0] : T(0);
    }
}

// Non-contact generalized force F(q, v) = gamma^T(-C + tau_g) (multibody_terms.py:142-146, n_u = 0)
// by recursive Newton-Euler at zero generalized acceleration.  Also returns the body twists V and
// the bias-minus-gravity spatial accelerations AG = A_b - G_b needed by the backward pass.
template <typename T, int NJ, class MD>
DPLL_HD void bias_forces(const MD& md, const Kin<T, NJ>& k, const T (&iota)[NJ + 1][kIota], const T* v,
                         T (&F)[6 + NJ], T (&Vw)[NJ + 1][3], T (&Vu)[NJ + 1][3], T (&AGw)[NJ + 1][3],
                         T (&AGu)[NJ + 1][3]) {
  constexpr int NB = NJ + 1;
  body_twists<T, NJ>(k, v, Vw, Vu);
  T Aw[NB][3], Au[NB][3];
  {
    T wxu[3];
    cross(Vw[0], Vu[0], wxu);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { Aw[0][i] = T(0); Au[0][i] = -wxu[i]; }
  }
  DPLL_UNROLL for (int j = 1; j <= NJ; ++j) {
    T pw[3], pu[3], wxp[3], t[3], r1[3], r2[3];
    pick3(Aw, k.par[j], j, pw);
    pick3(Au, k.par[j], j, pu);
    cross(pw, k.pj[j], wxp);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) t[i] = pu[i] + wxp[i];
    mat3t_vec(k.Rpc[j], t, r2);
    mat3t_vec(k.Rpc[j], pw, r1);
    const T rate = v[6 + j - 1];
    T sr[3], c1[3], c2[3];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) sr[i] = k.ax[j][i] * rate;
    // V x (s rate): (w x s_w, w x s_u + u x s_w) -- revolute s = (axis, 0), prismatic s = (0, axis)
    cross(Vw[j], sr, c1);
    cross(Vu[j], sr, c2);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) {
      Aw[j][i] = r1[i] + (k.pris[j] ? T(0) : c1[i]);
      Au[j][i] = r2[i] + (k.pris[j] ? c1[i] : c2[i]);
    }
  }
  const T gw[3] = {T(0), T(0), T(md.gravity_z)};
  T Wn[NB][3], Wf[NB][3];
  DPLL_UNROLL for (int b = 0; b < NB; ++b) {
    T gb[3];
    mat3t_vec(k.R[b], gw, gb);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { AGw[b][i] = Aw[b][i]; AGu[b][i] = Au[b][i] - gb[i]; }
    T n1[3], f1[3], hn[3], hf[3];
    inertia_apply(iota[b], AGw[b], AGu[b], n1, f1);
    inertia_apply(iota[b], Vw[b], Vu[b], hn, hf);
    // V x* (n, f) = (w x n + u x f, w x f)
    T a1[3], a2[3], a3[3];
    cross(Vw[b], hn, a1);
    cross(Vu[b], hf, a2);
    cross(Vw[b], hf, a3);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) { Wn[b][i] = n1[i] + a1[i] + a2[i]; Wf[b][i] = f1[i] + a3[i]; }
  }
  DPLL_UNROLL for (int j = NJ; j >= 1; --j) {
    F[6 + j - 1] = k.pris[j] ? -dot3(k.ax[j], Wf[j]) : -dot3(k.ax[j], Wn[j]);
    T rn[3], rf[3], pxf[3];
    mat3_vec(k.Rpc[j], Wn[j], rn);
    mat3_vec(k.Rpc[j], Wf[j], rf);
    cross(k.pj[j], rf, pxf);
    DPLL_UNROLL for (int c = 0; c < j; ++c) {
      const bool mine = (k.par[j] == c);
      DPLL_UNROLL for (int i = 0; i < 3; ++i) { Wn[c][i] += mine ? rn[i] + pxf[i] : T(0); Wf[c][i] += mine ? rf[i] : T(0); }
    }
  }
  T wf[3];
  mat3_vec(k.R[0], Wf[0], wf);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { F[i] = -Wn[0][i]; F[3 + i] = -wf[i]; }
}

// ---------------------------------------------------------------------------------------------
// dense symmetric positive definite N x N: Cholesky (lower), solves
// ---------------------------------------------------------------------------------------------
template <typename T, int N> DPLL_HD void cholesky(const T (&A)[N][N], T (&L)[N][N], T (&invd)[N]) {
  DPLL_UNROLL for (int j = 0; j < N; ++j) {
    T s = A[j][j];
    DPLL_UNROLL for (int p = 0; p < j; ++p) s -= L[j][p] * L[j][p];
    const T d = tsqrt(s);
    const T id = T(1) / d;
    L[j][j] = d;
    invd[j] = id;
    DPLL_UNROLL for (int i = j + 1; i < N; ++i) {
      T t = A[i][j];
      DPLL_UNROLL for (int p = 0; p < j; ++p) t -= L[i][p] * L[j][p];
      L[i][j] = t * id;
    }
  }
}
template <typename T, int N>
DPLL_HD void chol_solve(const T (&L)[N][N], const T (&invd)[N], const T (&b)[N], T (&x)[N]) {
  T y[N];
  DPLL_UNROLL for (int i = 0; i < N; ++i) {
    T s = b[i];
    DPLL_UNROLL for (int p = 0; p < i; ++p) s -= L[i][p] * y[p];
    y[i] = s * invd[i];
  }
  DPLL_UNROLL for (int i = N - 1; i >= 0; --i) {
    T s = y[i];
    DPLL_UNROLL for (int p = i + 1; p < N; ++p) s -= L[p][i] * x[p];
    x[i] = s * invd[i];
  }
}
template <typename T, int N> DPLL_HD void symv(const T (&A)[N][N], const T (&x)[N], T (&y)[N]) {
  DPLL_UNROLL for (int i = 0; i < N; ++i) {
    T s = T(0);
    DPLL_UNROLL for (int j = 0; j < N; ++j) s += A[i][j] * x[j];
    y[i] = s;
  }
}
template <typename T, int N> DPLL_HD T dotn(const T (&a)[N], const T (&b)[N]) {
  T s = T(0);
  DPLL_UNROLL for (int i = 0; i < N; ++i) s += a[i] * b[i];
  return s;
}

// ---------------------------------------------------------------------------------------------
// contacts: box vs ground half-space (geometry.py:554-582, :162-202, :393-403)
// ---------------------------------------------------------------------------------------------
// The `slot`-th (0..3) of the four box corners with the largest support value in body direction d
// (the reference's torch.topk(sorted=False) leaves the order of the four unspecified, quirk Q3).
// With a_i = |d_i| h_i sorted a1 >= a2 >= a3 the four best corners are: all signs aligned with d;
// smallest flipped; middle flipped; and either largest flipped (a1 < a2 + a3) or both smaller ones.
template <typename T> DPLL_HD void box_corner_signs(const T (&d)[3], const T (&habs)[3], int slot, T (&sgn)[3]) {
  const T a[3] = {tabs(d[0]) * habs[0], tabs(d[1]) * habs[1], tabs(d[2]) * habs[2]};
  // rank[i] = number of entries strictly larger (ties broken by index) -> 0 largest .. 2 smallest
  int rank[3];
  rank[0] = (a[1] > a[0]) + (a[2] > a[0]);
  rank[1] = (a[0] >= a[1]) + (a[2] > a[1]);
  rank[2] = (a[0] >= a[2]) + (a[1] >= a[2]);
  T amax = T(0), arest = T(0);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { amax = (rank[i] == 0) ? a[i] : amax; arest += (rank[i] == 0) ? T(0) : a[i]; }
  const bool flip_largest = amax < arest;
  DPLL_UNROLL for (int i = 0; i < 3; ++i) {
    bool flip = false;
    flip = flip || (slot == 1 && rank[i] == 2);
    flip = flip || (slot == 2 && rank[i] == 1);
    flip = flip || (slot == 3 && (flip_largest ? rank[i] == 0 : rank[i] != 0));
    const T s = (d[i] < T(0)) ? T(-1) : T(1);
    sgn[i] = flip ? -s : s;
  }
}

// Contact-point Jacobian of a point `pt` rigidly attached to body `b` (world = contact frame, the ground's
// kinematics are identically zero):  Jp = [ A | 1 | j_1 .. j_NJ ]  with  A = -S(pt - o_0) R_0  and
// j_i = a_i x (pt - o_i) for the joints between the base and body b (0 otherwise)
// (multibody_terms.py:385-399 with tensor_utils.py:257-302, restated in closed form).  Only A and the joint
// columns are stored: the identity block costs nothing.
// DENSE (the general build): all 3 x n_v entries -- a body-body contact is the difference of two point Jacobians rotated
// into its contact frame (multibody_terms.py:497-513), which has no identity block.
template <typename T, int NJ, bool DENSE = false> struct CJac {
  T A[3][3];
  T j[NJ > 0 ? NJ : 1][3];
};
template <typename T, int NJ> struct CJac<T, NJ, true> {
  T m[3][6 + NJ];
};

template <typename T, int NJ>
DPLL_HD void contact_jacobian(const Kin<T, NJ>& k, int b, const T (&pt)[3], CJac<T, NJ>& J) {
  T d0[3];
  DPLL_UNROLL for (int i = 0; i < 3; ++i) d0[i] = pt[i] - k.o[0][i];
  DPLL_UNROLL for (int c = 0; c < 3; ++c) {
    const T col[3] = {k.R[0][0][c], k.R[0][1][c], k.R[0][2][c]};
    T x[3];
    cross(col, d0, x);
    DPLL_UNROLL for (int r = 0; r < 3; ++r) J.A[r][c] = x[r];
  }
  DPLL_UNROLL for (int jj = 1; jj <= NJ; ++jj) {
    T dj[3], x[3];
    DPLL_UNROLL for (int i = 0; i < 3; ++i) dj[i] = pt[i] - k.o[jj][i];
    cross(k.axw[jj], dj, x);
    const bool moves = joint_moves(k, jj, b);
    DPLL_UNROLL for (int r = 0; r < 3; ++r) J.j[jj - 1][r] = moves ? (k.pris[jj] ? k.axw[jj][r] : x[r]) : T(0);
  }
}
// Jp y and Jp^T a
template <typename T, typename TY, int NJ> DPLL_HD void cjac_apply(const CJac<T, NJ>& J, const TY* y, TY (&out)[3]) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) {
    TY s = y[3 + r];
    DPLL_UNROLL for (int c = 0; c < 3; ++c) s += TY(J.A[r][c]) * y[c];
    DPLL_UNROLL for (int jj = 0; jj < NJ; ++jj) s += TY(J.j[jj][r]) * y[6 + jj];
    out[r] = s;
  }
}
template <typename T, int NJ> DPLL_HD void cjac_apply_t_add(const CJac<T, NJ>& J, const T (&a)[3], T* out) {
  DPLL_UNROLL for (int c = 0; c < 3; ++c) {
    out[c] += J.A[0][c] * a[0] + J.A[1][c] * a[1] + J.A[2][c] * a[2];
    out[3 + c] += a[c];
  }
  DPLL_UNROLL for (int jj = 0; jj < NJ; ++jj) out[6 + jj] += J.j[jj][0] * a[0] + J.j[jj][1] * a[1] + J.j[jj][2] * a[2];
}

template <typename T, typename TY, int NJ> DPLL_HD void cjac_apply(const CJac<T, NJ, true>& J, const TY* y, TY (&out)[3]) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) {
    TY s = TY(0);
    DPLL_UNROLL for (int c = 0; c < 6 + NJ; ++c) s += TY(J.m[r][c]) * y[c];
    out[r] = s;
  }
}
template <typename T, int NJ> DPLL_HD void cjac_apply_t_add(const CJac<T, NJ, true>& J, const T (&a)[3], T* out) {
  DPLL_UNROLL for (int c = 0; c < 6 + NJ; ++c) out[c] += J.m[0][c] * a[0] + J.m[1][c] * a[1] + J.m[2][c] * a[2];
}
// dense = sign * F [A | 1 | j] (+ what is there already when `add`); F rows = contact-frame axes in the world
template <typename T, int NJ>
DPLL_HD void cjac_dense(const CJac<T, NJ, false>& P, const T (&F)[3][3], T sign, bool add, CJac<T, NJ, true>& J) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) {
    DPLL_UNROLL for (int c = 0; c < 3; ++c) {
      const T a = sign * (F[r][0] * P.A[0][c] + F[r][1] * P.A[1][c] + F[r][2] * P.A[2][c]);
      J.m[r][c] = add ? J.m[r][c] + a : a;
      J.m[r][3 + c] = add ? J.m[r][3 + c] + sign * F[r][c] : sign * F[r][c];
    }
    DPLL_UNROLL for (int jj = 0; jj < NJ; ++jj) {
      const T a = sign * (F[r][0] * P.j[jj][0] + F[r][1] * P.j[jj][1] + F[r][2] * P.j[jj][2]);
      J.m[r][6 + jj] = add ? J.m[r][6 + jj] + a : a;
    }
  }
}
// column i of the Jacobian (the terms kernels write J out)
template <typename T, int NJ> DPLL_HD void cjac_column(const CJac<T, NJ, false>& J, int i, T (&col)[3]) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r)
    col[r] = i < 3 ? J.A[r][i] : (i < 6 ? (r == i - 3 ? T(1) : T(0)) : J.j[(i >= 6 && i - 6 < NJ) ? i - 6 : 0][r]);
}
template <typename T, int NJ> DPLL_HD void cjac_column(const CJac<T, NJ, true>& J, int i, T (&col)[3]) {
  DPLL_UNROLL for (int r = 0; r < 3; ++r) col[r] = J.m[r][i];
}
// H (lower triangle) += J^T C J for one contact, C symmetric 3 x 3
template <typename T, int NJ>
DPLL_HD void hessian_add(const CJac<T, NJ, false>& J, const T (&C)[3][3], T (&H)[6 + NJ][6 + NJ]) {
  T CA[3][3];
  mat3_mul(C, J.A, CA);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) {
    DPLL_UNROLL for (int j = 0; j <= i; ++j) {
      H[i][j] += J.A[0][i] * CA[0][j] + J.A[1][i] * CA[1][j] + J.A[2][i] * CA[2][j];
      H[3 + i][3 + j] += C[i][j];
    }
    DPLL_UNROLL for (int j = 0; j < 3; ++j) H[3 + i][j] += CA[i][j];
  }
  DPLL_UNROLL for (int jj = 0; jj < NJ; ++jj) {
    T u[3];
    mat3_vec(C, J.j[jj], u);
    DPLL_UNROLL for (int i = 0; i < 3; ++i) {
      H[6 + jj][i] += J.A[0][i] * u[0] + J.A[1][i] * u[1] + J.A[2][i] * u[2];
      H[6 + jj][3 + i] += u[i];
    }
    DPLL_UNROLL for (int kk = 0; kk <= jj; ++kk) H[6 + jj][6 + kk] += dot3(J.j[kk], u);
  }
}
template <typename T, int NJ>
DPLL_HD void hessian_add(const CJac<T, NJ, true>& J, const T (&C)[3][3], T (&H)[6 + NJ][6 + NJ]) {
  constexpr int NV = 6 + NJ;
  T CJ[3][NV];
  DPLL_UNROLL for (int r = 0; r < 3; ++r)
    DPLL_UNROLL for (int c = 0; c < NV; ++c) CJ[r][c] = C[r][0] * J.m[0][c] + C[r][1] * J.m[1][c] + C[r][2] * J.m[2][c];
  DPLL_UNROLL for (int i = 0; i < NV; ++i)
    DPLL_UNROLL for (int j = 0; j <= i; ++j) H[i][j] += J.m[0][i] * CJ[0][j] + J.m[1][i] * CJ[1][j] + J.m[2][i] * CJ[2][j];
}

// world angular velocity of body b under generalized velocity y
template <typename T, int NJ> DPLL_HD void world_omega(const Kin<T, NJ>& k, int b, const T* y, T (&w)[3]) {
  const T yb[3] = {y[0], y[1], y[2]};
  mat3_vec(k.R[0], yb, w);
  DPLL_UNROLL for (int j = 1; j <= NJ; ++j) {
    const bool moves = joint_moves(k, j, b) && !k.pris[j];  // (a prismatic joint turns nothing)
    DPLL_UNROLL for (int i = 0; i < 3; ++i) w[i] += moves ? k.axw[j][i] * y[6 + j - 1] : T(0);
  }
}

// ---------------------------------------------------------------------------------------------
// Lorentz-cone projection, z = [t_x, t_y, n] (tensor_utils.py:460-497 ordering), unit cone
// ---------------------------------------------------------------------------------------------
// The three regions share one parametrisation of the generalised Jacobian,
//   dP = cp p p^T + a (t t^T + e3 e3^T) + b (t e3^T + e3 t^T),   t = z_t / |z_t|,  p = (-t_y, t_x, 0):
// identity region (cp, a, b) = (1, 1, 0), cone surface (s / r, 1/2, 1/2), polar region (0, 0, 0); the projection
// itself is g = (cp z_t, inside ? n : s) with s = max((n + r) / 2, 0).  Everything downstream is select free.
template <typename T> struct Proj {
  T g[3];       // projection
  T that[2];    // unit tangential direction ((1, 0) where z_t = 0)
  T cp, a, b;   // coefficients above
  bool inside;  // identity region
  bool polar;   // zero region (neither: the "mid" region, projection onto the cone surface)
};
template <typename T> DPLL_HD void lorentz_project(const T (&z)[3], Proj<T>& p) {
  const T r2 = z[0] * z[0] + z[1] * z[1];
  // (a tangential part whose square is not a normal number counts as none: v_rsq_f32 returns inf for a denormal input and
  // r = r2 * inf poisons the solve -- a cube that has come to rest in a long rollout gets there, |z_t| ~ 1e-20)
  const bool pos = r2 > T(sizeof(T) == 4 ? 1e-34 : 1e-300);
  const T ir = pos ? fast_rsqrt(r2) : T(0);
  const T r = r2 * ir;
  const T n = z[2];
  const T sraw = T(0.5) * (n + r);
  p.inside = r <= n;
  p.polar = !(sraw > T(0)) && !p.inside;
  const T s = tmax(sraw, T(0));
  p.that[0] = pos ? z[0] * ir : T(1);
  p.that[1] = z[1] * ir;
  const T half = (sraw > T(0)) ? T(0.5) : T(0);
  p.cp = p.inside ? T(1) : s * ir;
  p.a = p.inside ? T(1) : half;
  p.b = p.inside ? T(0) : half;
  p.g[0] = p.cp * z[0];
  p.g[1] = p.cp * z[1];
  p.g[2] = p.inside ? n : s;
}
// w^T dP w
template <typename T> DPLL_HD T proj_quadratic(const Proj<T>& p, const T (&w)[3]) {
  const T a1 = p.that[0] * w[1] - p.that[1] * w[0];
  const T u = p.that[0] * w[0] + p.that[1] * w[1];
  return p.cp * a1 * a1 + p.a * (u * u + w[2] * w[2]) + (p.b + p.b) * u * w[2];
}
// dP as a symmetric 3x3 (xx, yy, zz, xy, xz, yz)
template <typename T> DPLL_HD void proj_jacobian(const Proj<T>& p, T (&d)[6]) {
  const T tx = p.that[0], ty = p.that[1];
  const T txx = tx * tx, tyy = ty * ty, txy = tx * ty;
  d[0] = p.cp * tyy + p.a * txx;
  d[1] = p.cp * txx + p.a * tyy;
  d[2] = p.a;
  d[3] = (p.a - p.cp) * txy;
  d[4] = p.b * tx;
  d[5] = p.b * ty;
}

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
};

template <typename T, int NJ, int NG, class MD>
DPLL_HD void derive_params(const MD& md, const T* theta, const T* friction, const T* lengths, Derived<T, NJ, NG>& dp) {
  const T mu0 = tabs(friction[0]);
  DPLL_UNROLL for (int b = 0; b <= NJ; ++b) {
    T th[10];
    DPLL_UNROLL for (int i = 0; i < 10; ++i) th[i] = theta[10 * b + i];
    theta_to_iota<T>(th, md.inertia_mode, dp.iota[b]);
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
