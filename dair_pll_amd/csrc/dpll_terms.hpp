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
  int32_t inertia_mode;  // 0: reference_literal (rotational inertia taken as I_cm / m), 1: physical, 2: composed (rows are iota)
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
  int32_t n_u;                        // actuators (general build): input k is a generalized force on joint act_joint[k] + 1
  int32_t act_joint[kMaxJoints];
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
  if (inertia_mode == 2) {  // the rows ARE the bodies' inertial vectors (composites of welded links: csrc/dpll_weld.hip)
    DPLL_UNROLL for (int i = 0; i < kIota; ++i) iota[i] = th[i];
    return;
  }
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

}  // namespace dpll
