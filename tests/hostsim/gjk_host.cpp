// TEST INFRASTRUCTURE ONLY: csrc/dpll_gjk.hpp compiled for the host (one lane), for tests/test_gjk.py.
#include <cstdint>
#include "../../dair_pll_amd/csrc/dpll_gjk.hpp"
using namespace dpll;
extern "C" int gjk_host_direction(const double* va, int na, const double* vb, int nb, const double* R, const double* p,
                                  double* d, double* sep, int* info) {
  if (na < 1 || nb < 1) return -1;
  HullPair<double> hp;
  hp.va = (const double (*)[3])va; hp.na = na; hp.vb = (const double (*)[3])vb; hp.nb = nb;
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) hp.R[r][c] = R[3 * r + c]; hp.p[r] = p[r]; }
  static thread_local EpaStore<double> st;
  PairDirResult<double> out;
  hull_pair_direction<double, OneLane>(hp, st, out);
  for (int i = 0; i < 3; ++i) d[i] = out.d[i];
  *sep = out.sep;
  info[0] = out.status; info[1] = out.gjk_iters; info[2] = out.epa_iters;
  return 0;
}
