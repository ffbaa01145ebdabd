// dpll_forest_api.hpp -- entry points of the forest build (dpll_forest.hip) for the C ABI in dpll_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dpll.h"

struct dpll_model;
struct AdamArgs;

namespace dpll_forest_api {
// validates the description; 0 or a negative status with dpll_last_error set
int check_desc(const dpll_forest_desc_t* desc);
int n_x(const dpll_model* m);
int n_contacts(const dpll_model* m);
int n_u(const dpll_model* m);  // actuators: the width of dpll_params_t.u
int param_count(const dpll_model* m);
long long workspace_bytes(const dpll_model* m, long long batch);
void release(dpll_model* m);  // the device copy of the description
int loss(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp, long long ld_xp,
         long long batch, const void* weights, double scale, void* loss_out, void* grad, void* loss_total, void* force,
         int32_t* iters, void* workspace, long long ws_bytes, hipStream_t stream, const AdamArgs* adam = nullptr);
int simulate(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x0, long long ld_x, long long batch,
             long long steps, void* out, long long ld_item, long long ld_step, int write_x0, int32_t* iters, hipStream_t stream);
int step_backward(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* gx,
                  long long ld_g, long long batch, void* grad, void* grad_x, long long ld_gx, void* workspace, long long ws_bytes,
                  hipStream_t stream);
int terms(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm, void* M,
          void* J, void* phi, void* a, hipStream_t stream);
}  // namespace dpll_forest_api
