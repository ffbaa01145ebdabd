// dpll_weld.hpp -- the arithmetic of dpll_weld_compose / dpll_weld_compose_backward (include/dpll.h), one entry per call so
// that the device kernels (dpll_weld.hip: a thread per entry) and the host build of the tests (tests/hostsim) run the same
// code.  Rows: theta format, one per link; host[r] = the kernel body that carries link r; X: (n_rows, 10, 10) row-major, the
// rigid transform of an inertial vector [m, m c, I_o] from the link's frame to the body's.  Double arithmetic.
#pragma once
#include "dpll_terms.hpp"

namespace dpll {

// component i of body b's inertial vector: sum over its links of (X_r theta_to_iota(theta_r))[i]
template <typename T>
DPLL_HD double weld_compose_entry(int inertia_mode, int n_rows, const int32_t* host, const double* X, const T* theta, int b, int i) {
  double total = 0.0;
  for (int r = 0; r < n_rows; ++r) {
    if (host[r] != b) continue;
    double th[10], io[kIota];
    for (int k = 0; k < 10; ++k) th[k] = double(theta[10 * r + k]);
    theta_to_iota<double>(th, inertia_mode, io);
    const double* row = X + (size_t)r * 100 + i * 10;
    for (int j = 0; j < kIota; ++j) total += row[j] * io[j];
  }
  return total;
}

// d / d theta_r[c] = sum_i g[host r][i] sum_j X_r[i][j] d iota_r[j] / d theta_r[c]   (g: the bodies' d loss / d iota)
template <typename T>
DPLL_HD double weld_backward_entry(int inertia_mode, const int32_t* host, const double* X, const T* theta, const T* grad_iota, int r, int c) {
  DualT<double> th[10], io[kIota];
  for (int k = 0; k < 10; ++k) th[k] = DualT<double>(double(theta[10 * r + k]), k == c ? 1.0 : 0.0);
  theta_to_iota<DualT<double>>(th, inertia_mode, io);
  const T* g = grad_iota + 10 * host[r];
  double total = 0.0;
  for (int i = 0; i < kIota; ++i) {
    const double* row = X + (size_t)r * 100 + i * 10;
    double xd = 0.0;
    for (int j = 0; j < kIota; ++j) xd += row[j] * io[j].d;
    total += double(g[i]) * xd;
  }
  return total;
}

}  // namespace dpll
