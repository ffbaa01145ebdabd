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
#include "dpll_core.hpp"

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
  // support points / their adjoints of query n = 4 item + s sit at element (n / 4) * point_stride + 3 s of the arrays the
  // kernels are given: 12 when this geometry is the item's only one, 3 * (contacts per item) when it is one of several
  // (the arrays are then offset by 12 * geometry index)
  int point_stride = 12;
};
DPLL_HD long long icnn_point_index(long long n, int point_stride) { return (n >> 2) * point_stride + (n & 3) * 3; }

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
