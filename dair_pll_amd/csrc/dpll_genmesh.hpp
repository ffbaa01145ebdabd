// dpll_genmesh.hpp -- entry points of the general build WITH learned shapes (dpll_genmesh.hip) for the mesh pipeline in
// dpll_kernels.hip: the item kernels that take support points as inputs, and the kernel that prepares the networks' queries
// (ground directions of every state and -- for a body-body candidate between two learned shapes, the reference's own case,
// geometry.py:543-546, :585-643 -- the direction between the two shapes by GJK / EPA on their extracted vertex sets).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dpll.h"

struct dpll_model;
struct AdamArgs;

namespace dpll_genmesh {

constexpr int kMaxQueries = 8;  // per geometry and item: 4 ground queries + one per body-body candidate it is part of

// what the query kernel reads and writes; arrays over the geometries g < DPLL_MAX_GEOMS (entries of geometries that are
// not learned shapes are null / zero)
struct QueryPlan {
  const void* pert[DPLL_MAX_GEOMS];   // (4, 3) perturbations of geometry g
  void* dirs[DPLL_MAX_GEOMS];         // (batch, qpi[g], 3) <- unit query directions of network g
  const void* hull[DPLL_MAX_GEOMS];   // (296, 3) support points of network g over the surface directions (its vertex set)
  int32_t qpi[DPLL_MAX_GEOMS];        // queries per item of network g
  int32_t query_a[DPLL_MAX_PAIRS];    // candidate p: index of its query among those of geometry pair_a[p] (direction d) ...
  int32_t query_b[DPLL_MAX_PAIRS];    // ... and of pair_b[p] (direction -d seen from B)
  double* pdirs;                      // (batch, DPLL_MAX_PAIRS, 3) <- direction of candidate p in the frame of A
  int32_t* status;                    // optional (batch, DPLL_MAX_PAIRS) <- 0 ok | search diagnostics (dpll_gjk.hpp)
};

// floats per item of the witness / adjoint arrays: (slot 0..15, side 0..1, 3)
int wit_per_item();
// element offset of (slot, side) inside an item's block
inline int wit_offset(int slot, int side) { return (slot * 2 + side) * 3; }

int surface_directions(int dtype, void* out, hipStream_t stream);  // (296, 3) unit directions, deep_support_function.py:12-16
int queries(const dpll_model* m, int dtype, const QueryPlan& plan, const void* x, long long ld_x, long long batch, hipStream_t stream);

long long workspace_bytes(const dpll_model* m, long long batch);  // rows + chain matrix + folded rows of the item kernels
int loss_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp, long long ld_xp,
               long long batch, const void* weights, double scale, void* loss_out, void* force, int32_t* iters, void* rows,
               int want_grad, const void* wit, void* rbar, const double* pdirs, hipStream_t stream);
// sums the rows loss_items left and chains them to [theta | friction | lengths] (the head of the gradient)
// adam: the fused training step -- the kernel that writes the head of the gradient applies Adam to the head of the parameters
int finalize(const dpll_model* m, int dtype, long long batch, void* rows, void* grad, void* loss_total, hipStream_t stream,
             const AdamArgs* adam = nullptr);
int step_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* x_next,
               long long ld_next, const void* wit, const double* pdirs, hipStream_t stream);
// backward of one step (parameter rows + state adjoint + witness adjoints); finalize() chains the rows afterwards
int step_backward_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* gx,
                        long long ld_g, long long batch, void* rows, void* grad_x, long long ld_gx, const void* wit, void* rbar,
                        const double* pdirs, hipStream_t stream);
int terms_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm,
                void* M, void* J, void* phi, void* a, const void* wit, const double* pdirs, hipStream_t stream);
}  // namespace dpll_genmesh
