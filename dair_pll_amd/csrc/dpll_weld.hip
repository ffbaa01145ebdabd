// dpll_weld.hip -- inertial rows of links welded together by `fixed` joints (include/dpll.h: dpll_weld_compose*).
//
// The reference learns one theta row per DRAKE body (multibody_terms.py:161-207) and Drake keeps a welded link as a body of its
// own; the kernels' bodies are the links that move against each other.  A kernel body's inertial vector is the sum of its
// links' vectors, each taken to the body's frame by a constant linear map X_r ([m, m c, I_o] transforms linearly under a rigid
// change of frame), so the map theta_rows -> iota_bodies and its transpose-Jacobian product are two one-workgroup kernels
// around the launches of a DPLL_INERTIA_COMPOSED model.  Arithmetic in double whatever the storage type: 10 numbers per row.
#include <hip/hip_runtime.h>

#include "../../include/dpll.h"
#include "dpll_weld.hpp"

int dpll_fail(int code, const char* fmt, const char* detail = "");  // (dpll_kernels.hip)
int dpll_check_launch(const char* what);

namespace {

using dpll::kIota;

// thread (b, i): component i of body b's inertial vector
template <typename T>
__global__ void weld_compose_kernel(int inertia_mode, int n_rows, int n_bodies, const int32_t* __restrict__ host,
                                    const double* __restrict__ X, const T* __restrict__ theta, T* __restrict__ iota) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_bodies * kIota) return;
  iota[t] = T(dpll::weld_compose_entry<T>(inertia_mode, n_rows, host, X, theta, t / kIota, t % kIota));
}

// thread (r, c): entry c of row r's gradient
template <typename T>
__global__ void weld_backward_kernel(int inertia_mode, int n_rows, const int32_t* __restrict__ host, const double* __restrict__ X,
                                     const T* __restrict__ theta, const T* __restrict__ grad_iota, T* __restrict__ grad_theta,
                                     int accumulate) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_rows * 10) return;
  const double total = dpll::weld_backward_entry<T>(inertia_mode, host, X, theta, grad_iota, t / 10, t % 10);
  grad_theta[t] = T(accumulate ? double(grad_theta[t]) + total : total);
}

int check(const char* who, int dtype, int inertia_mode, int n_rows, int n_bodies, const void* host, const void* transforms) {
  if (dtype != DPLL_F32 && dtype != DPLL_F64) return dpll_fail(-1, "%s: unknown dtype", who);
  if (inertia_mode != DPLL_INERTIA_REFERENCE_LITERAL && inertia_mode != DPLL_INERTIA_PHYSICAL)
    return dpll_fail(-1, "%s: inertia_mode must be REFERENCE_LITERAL or PHYSICAL (it applies to the links' own rows)", who);
  if (n_bodies < 1 || n_rows < n_bodies || n_rows > DPLL_MAX_WELD_ROWS)
    return dpll_fail(-1, "%s: need 1 <= n_bodies <= n_rows <= DPLL_MAX_WELD_ROWS", who);
  if (!host || !transforms) return dpll_fail(-1, "%s: host / transforms missing", who);
  return 0;
}

}  // namespace

extern "C" int dpll_weld_compose(int dtype, int inertia_mode, int n_rows, int n_bodies, const int32_t* host, const double* transforms,
                                 const void* theta_rows, void* iota, void* stream) {
  if (int rc = check("dpll_weld_compose", dtype, inertia_mode, n_rows, n_bodies, host, transforms)) return rc;
  if (!theta_rows || !iota) return dpll_fail(-1, "dpll_weld_compose: theta_rows / iota missing%s");
  const int threads = n_bodies * kIota, block = 64, grid = (threads + block - 1) / block;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DPLL_F32)
    weld_compose_kernel<float><<<grid, block, 0, s>>>(inertia_mode, n_rows, n_bodies, host, transforms, (const float*)theta_rows, (float*)iota);
  else
    weld_compose_kernel<double><<<grid, block, 0, s>>>(inertia_mode, n_rows, n_bodies, host, transforms, (const double*)theta_rows, (double*)iota);
  return dpll_check_launch("dpll_weld_compose");
}

extern "C" int dpll_weld_compose_backward(int dtype, int inertia_mode, int n_rows, int n_bodies, const int32_t* host,
                                          const double* transforms, const void* theta_rows, const void* grad_iota,
                                          void* grad_theta_rows, int accumulate, void* stream) {
  if (int rc = check("dpll_weld_compose_backward", dtype, inertia_mode, n_rows, n_bodies, host, transforms)) return rc;
  if (!theta_rows || !grad_iota || !grad_theta_rows) return dpll_fail(-1, "dpll_weld_compose_backward: a buffer is missing%s");
  const int threads = n_rows * 10, block = 64, grid = (threads + block - 1) / block;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DPLL_F32)
    weld_backward_kernel<float><<<grid, block, 0, s>>>(inertia_mode, n_rows, host, transforms, (const float*)theta_rows,
                                                       (const float*)grad_iota, (float*)grad_theta_rows, accumulate);
  else
    weld_backward_kernel<double><<<grid, block, 0, s>>>(inertia_mode, n_rows, host, transforms, (const double*)theta_rows,
                                                        (const double*)grad_iota, (double*)grad_theta_rows, accumulate);
  return dpll_check_launch("dpll_weld_compose_backward");
}
