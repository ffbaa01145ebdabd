// dpll_icnn.hpp -- support points of a learned convex shape: DeepSupportConvex / HomogeneousICNN of dair_pll
// (geometry.py:255-325, deep_support_function.py:125-266), depth 2, width 256, LeakyReLU(0.5).
//
//   query q = normalize(d + perturbation_s), d = -(row 2 of R_world<-body)        geometry.py:309-325, 560-564
//   h0 = act(q Wd0)            m0 = act'(h0)                                        deep_support_function.py:213-236
//   h1 = act(h0 |Wh| + q Wd1)  m1 = act'(h1)
//   support point p = d f / d q = Wd1 (|wout| . m1) + Wd0 ((|Wh| (|wout| . m1)) . m0)      :238-266
// The masks are constants of the reference's autograd graph, so p is LINEAR in each weight tensor given the
// masks; with u1 = |wout| . m1, v = |Wh| u1, u0 = v . m0 and an upstream p_bar:
//   dWd1 = p_bar u1^T,  dWd0 = p_bar u0^T,  v_bar = (Wd0^T p_bar) . m0,  d|Wh| = v_bar u1^T,
//   u1_bar = Wd1^T p_bar + |Wh|^T v_bar,  d|wout| = u1_bar . m1, then the sign of the raw weights.
#pragma once
#include "dpll_terms.hpp"  // (DPLL_HD, tsqrt: nothing of the solver or the item kernels)

namespace dpll {

constexpr int kIcnnWidth = 256;
constexpr double kIcnnSlope = 0.5;

// device (or host) pointers to the raw, signed parameters and the fixed perturbation buffer
template <typename T> struct IcnnWeights {
  const T* Wh;    // (256, 256) hidden_weights.0: h1_pre[j] += h0[k] |Wh[k][j]|
  const T* Wd0;   // (3, 256)   input_weights.0
  const T* Wd1;   // (3, 256)   input_weights.1
  const T* wout;  // (256,)     output_weight
  const T* pert;  // (4, 3)     perturbations, row 0 is zero (geometry.py:306-307)
  // Queries: n = qpi * item + j.  By default (dirs == nullptr, qpi = 4) query j of an item is the ground direction seen from
  // the body whose quaternion the kernels are given, plus perturbation row j.  With `dirs` (N, 3) the unit directions are
  // read from memory instead: the general build with learned shapes (csrc/dpll_genmesh.hip) writes the four ground queries
  // of the geometry there and, behind them, one query per body-body candidate the geometry is part of; the mesh extraction
  // (the 296 surface directions, deep_support_function.py:12-16) is the same with qpi = 1.
  const T* dirs = nullptr;
  int qpi = 4;
  // support points / their adjoints of query (item, j) sit at element item * point_stride + qoff[j] of the arrays the
  // kernels are given: one geometry alone 12 and {0, 3, 6, 9}; one of several: 3 * (contacts per item) with the arrays
  // offset by 12 * geometry index; the general build: its (slot, side) layout, see dpll_general_kernels.hpp
  int point_stride = 12;
  int qoff[8] = {0, 3, 6, 9, 0, 0, 0, 0};
};
template <typename T> DPLL_HD long long icnn_point_index(long long n, const IcnnWeights<T>& w) {
  const long long item = n / w.qpi;
  const int j = (int)(n - item * w.qpi);
  int off = w.qoff[0];
  DPLL_UNROLL for (int c = 1; c < 8; ++c) off = (j == c) ? w.qoff[c] : off;
  return item * w.point_stride + off;
}

// un-normalised rotation row 2 from the quaternion (same polynomial as quat_to_rot) -> query direction
template <typename T> DPLL_HD void icnn_query(const T* quat, const T* pert3, T (&q)[3]) {
  const T w = quat[0], x = quat[1], y = quat[2], z = quat[3];
  const T d[3] = {-T(2) * (x * z - w * y), -T(2) * (y * z + w * x), -(w * w - x * x - y * y + z * z)};
  T n2 = T(0);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) { q[i] = d[i] + pert3[i]; n2 += q[i] * q[i]; }
  const T inv = T(1) / tsqrt(n2);
  DPLL_UNROLL for (int i = 0; i < 3; ++i) q[i] *= inv;
}

template <typename T> DPLL_HD T icnn_act(T x) { return x > T(0) ? x : T(kIcnnSlope) * x; }
template <typename T> DPLL_HD T icnn_mask(T pre) { return pre > T(0) ? T(1) : T(kIcnnSlope); }

}  // namespace dpll
