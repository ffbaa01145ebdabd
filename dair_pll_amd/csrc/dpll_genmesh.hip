// dpll_genmesh.hip -- the general build WITH learned shapes: DeepSupportConvex geometries (geometry.py:255-364) on any
// tree the general build takes, next to boxes / spheres / polygons, and body-body candidates between two learned shapes --
// the one pair the reference itself dispatches (GeometryCollider.collide -> collide_mesh_mesh, geometry.py:543-546,
// :585-643; contactnets_elbow_mesh.urdf with the collision filter removed, or assets/clasp_mesh.urdf here).
//
// Pipeline of a call (orchestrated in dpll_kernels.hip, where the ICNN kernels live):
//   per network   icnn_prep; the 296 surface directions through the forward kernels -> the network's vertex set
//                 (extract_mesh, deep_support_function.py:93-123 -- the hull fcl is given; duplicates are harmless to a
//                 support-function search)
//   query kernel  (this file) per item: the ground direction seen from every learned geometry + its perturbations -> 4
//                 queries per geometry; per candidate between two learned shapes the direction d by GJK / EPA on the two
//                 vertex sets in LDS (csrc/dpll_gjk.hpp) -> query d for A, -R_AB^T d for B (geometry.py:627-629)
//   per network   forward kernels on its (4 + candidates) queries per item -> support points into the (slot, side) layout
//   item kernel   (this file: dpll_general_kernels.hpp with MESH = true) loss / step / terms with those witnesses; the
//                 loss writes the witness adjoints in the same layout
//   per network   backward kernels -> weight gradients; row fold + chain -> the head of the gradient
#include <hip/hip_runtime.h>

#include <cstdio>

#include "dpll_general_kernels.hpp"
#include "dpll_genmesh.hpp"
#include "dpll_gjk.hpp"

namespace {

// ---- the reference's surface directions: boundary nodes of an 8 x 8 x 8 grid on [-1, 1]^3, normalised, in the order of
// torch.cartesian_prod (deep_support_function.py:12-16); one block of 512 threads = the grid nodes
template <typename T> __global__ __launch_bounds__(512) void surface_dirs_kernel(T* __restrict__ out) {
  __shared__ int wave_count[8];
  const int t = threadIdx.x, i = t >> 6, j = (t >> 3) & 7, k = t & 7;
  const bool boundary = i == 0 || i == 7 || j == 0 || j == 7 || k == 0 || k == 7;
  const unsigned long long mask = __ballot(boundary);
  const int lane = t & 63, wv = t >> 6;
  if (lane == 0) wave_count[wv] = __popcll(mask);
  __syncthreads();
  int rank = __popcll(mask & ((1ull << lane) - 1ull));
  for (int w = 0; w < wv; ++w) rank += wave_count[w];
  if (!boundary) return;
  // torch.linspace(-1, 1, 8): start + i * step in the lower half, end - (7 - i) * step in the upper
  auto node = [](int a) { const double step = 2.0 / 7.0; return a < 4 ? -1.0 + a * step : 1.0 - (7 - a) * step; };
  const double c[3] = {node(i), node(j), node(k)};
  const double inv = 1.0 / sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  for (int a = 0; a < 3; ++a) out[3 * rank + a] = T(c[a] * inv);
}

// ---- queries of the networks ------------------------------------------------------------------------------------------
struct QueryArgs {
  const void* pert[kMaxGeoms];
  void* dirs[kMaxGeoms];
  const void* hull[kMaxGeoms];
  int qpi[kMaxGeoms];
  int query_a[kMaxPairs], query_b[kMaxPairs];
  double* pdirs;
  int* status;
};

// One item per 16-lane group (the general build's mapping), one wave per workgroup.  LDS: the two vertex sets of the
// candidate being searched (double, 2 x 7 KB) and one expansion polytope per group (4 x 9.5 KB).
template <typename T, int NJ>
__global__ __launch_bounds__(kWave) void mesh_query_kernel(GeneralDesc md, QueryArgs qa, const T* __restrict__ x, long long ld_x,
                                                           long long batch) {
  using TA = double;
  using Lanes = GenLanes;
  constexpr int G = kQuery * kNG, NB = NJ + 1;
  __shared__ TA hull_a[kHullDirs][3], hull_b[kHullDirs][3];
  __shared__ EpaStore<TA> store[kWave / G];
  const int lane = threadIdx.x, cidx = lane % G, slot = lane / G;
  const long long stride = (long long)gridDim.x * kIPW;
  for (long long base = (long long)blockIdx.x * kIPW; base < batch; base += stride) {
    const long long item = base + slot;
    const bool valid = item < batch;
    const long long it = valid ? item : batch - 1;
    TA q[7 + NJ];
#pragma unroll
    for (int i = 0; i < 7 + NJ; ++i) q[i] = TA(x[it * ld_x + i]);
    Kin<TA, NJ> kin;
    kinematics<TA, NJ>(md, q, kin);
    // ---- ground queries: lane 4 g + s -> query s of geometry g (collide_plane_convex, geometry.py:560-564, with
    // DeepSupportConvex.get_vertices, :309-325)
    {
      const int g = cidx / kQuery, s = cidx % kQuery;
      bool mesh = false;
      int body = 0;
      const T* pert = nullptr;
      T* out = nullptr;
      int qpi = 0;
      double G3[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
      for (int gg = 0; gg < kMaxGeoms; ++gg) {
        const bool pick = (g == gg);
        mesh = pick ? (gg < md.n_geoms && md.geom_kind[gg] == kGeomMesh) : mesh;
        body = pick ? md.geom_body[gg] : body;
        pert = pick ? (const T*)qa.pert[gg] : pert;
        out = pick ? (T*)qa.dirs[gg] : out;
        qpi = pick ? qa.qpi[gg] : qpi;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) G3[r][c] = pick ? md.geom_rot[gg][r][c] : G3[r][c];
      }
      TA R[3][3];
      pick33(kin.R, body, NB, R);
      if (DPLL_ROTATED(md) & 2) mat3_mul_const<TA>(R, G3);
      if (mesh && valid && g < kMaxGeoms) {
        TA d[3], n2 = TA(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) { d[i] = -R[2][i] + TA(pert[3 * s + i]); n2 += d[i] * d[i]; }
        const TA inv = TA(1) / sqrt(n2);
#pragma unroll
        for (int i = 0; i < 3; ++i) out[(it * qpi + s) * 3 + i] = T(d[i] * inv);
      }
    }
    // ---- body-body candidates between two learned shapes: the direction by GJK / EPA on their vertex sets
    for (int p = 0; p < kMaxPairs; ++p) {
      if (p >= md.n_pairs) break;
      const int ga = md.pair_a[p], gb = md.pair_b[p];
      if (md.geom_kind[ga] != kGeomMesh || md.geom_kind[gb] != kGeomMesh) continue;
      __syncthreads();  // (one wave per workgroup: orders the previous candidate's reads before these writes)
      for (int u = lane; u < kHullDirs; u += kWave)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          hull_a[u][i] = TA(((const T*)qa.hull[ga])[3 * u + i]);
          hull_b[u][i] = TA(((const T*)qa.hull[gb])[3 * u + i]);
        }
      __syncthreads();
      // the two geometry frames in the world (as pair_setup of dpll_core.hpp)
      HullPair<TA> hp;
      hp.va = hull_a; hp.na = kHullDirs; hp.vb = hull_b; hp.nb = kHullDirs;
      TA RA[3][3], RB[3][3], oA[3], oB[3];
      pick33(kin.R, md.geom_body[ga], NB, RA);
      pick33(kin.R, md.geom_body[gb], NB, RB);
      pick3(kin.o, md.geom_body[ga], NB, oA);
      pick3(kin.o, md.geom_body[gb], NB, oB);
      if (DPLL_ROTATED(md) & 2) {
        mat3_mul_const<TA>(RA, md.geom_rot[ga]);
        mat3_mul_const<TA>(RB, md.geom_rot[gb]);
      }
      TA gorgA[3], gorgB[3], cA[3], cB[3], rel[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { gorgA[i] = TA(md.geom_origin[ga][i]); gorgB[i] = TA(md.geom_origin[gb][i]); }
      mat3_vec(RA, gorgA, cA);
      mat3_vec(RB, gorgB, cB);
#pragma unroll
      for (int i = 0; i < 3; ++i) rel[i] = (cB[i] + oB[i]) - (cA[i] + oA[i]);
      mat3t_vec(RA, rel, hp.p);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) hp.R[r][c] = RA[0][r] * RB[0][c] + RA[1][r] * RB[1][c] + RA[2][r] * RB[2][c];
      PairDirResult<TA> res;
      hull_pair_direction<TA, Lanes>(hp, store[slot], res);
      if (valid && cidx == 0) {
        TA nd[3], dB[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) nd[i] = -res.d[i];
        mat3t_vec(hp.R, nd, dB);  // -d in the frame of B (geometry.py:628-629)
        T* da = (T*)qa.dirs[ga];
        T* db = (T*)qa.dirs[gb];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          qa.pdirs[(it * kMaxPairs + p) * 3 + i] = res.d[i];
          da[(it * qa.qpi[ga] + qa.query_a[p]) * 3 + i] = T(res.d[i]);
          db[(it * qa.qpi[gb] + qa.query_b[p]) * 3 + i] = T(dB[i]);
        }
        if (qa.status) qa.status[it * kMaxPairs + p] = res.status | (res.gjk_iters << 8) | (res.epa_iters << 16);
      }
    }
  }
}

template <typename T, int NJ>
int launch_queries(const dpll_model* m, const dpll_genmesh::QueryPlan& plan, const void* x, long long ld_x, long long batch,
                   hipStream_t stream) {
  QueryArgs qa;
  for (int g = 0; g < kMaxGeoms; ++g) { qa.pert[g] = plan.pert[g]; qa.dirs[g] = plan.dirs[g]; qa.hull[g] = plan.hull[g]; qa.qpi[g] = plan.qpi[g]; }
  for (int p = 0; p < kMaxPairs; ++p) { qa.query_a[p] = plan.query_a[p]; qa.query_b[p] = plan.query_b[p]; }
  qa.pdirs = plan.pdirs;
  qa.status = plan.status;
  long long blocks = (batch + kIPW - 1) / kIPW;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((mesh_query_kernel<T, NJ>), dim3((int)blocks), dim3(kWave), 0, stream, general_desc(m), qa, (const T*)x, ld_x, batch);
  return dpll_check_launch("mesh_query_kernel");
}

// ---- the item kernels with witnesses ------------------------------------------------------------------------------------
template <typename T, int NJ>
int launch_loss_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp,
                      long long ld_xp, long long batch, const void* weights, double scale, void* loss, void* force, int32_t* iters,
                      void* rows_ws, int want_grad, const void* wit, void* rbar, const double* pdirs, hipStream_t stream) {
  const int rows = row_blocks(batch);
  hipLaunchKernelGGL((gen_loss_kernel<T, NJ, true>), dim3(rows + 1), dim3(kWave), 0, stream, general_desc(m), m->opts[dtype],
                     (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)xp, ld_xp,
                     batch, (const T*)weights, scale, (T*)loss, (T*)force, (int*)iters, (double*)rows_ws, want_grad,
                     (const T*)wit, (T*)rbar, pdirs);
  return dpll_check_launch("gen_loss_kernel (mesh)");
}
template <typename T, int NJ>
int launch_finalize(const dpll_model* m, long long batch, void* rows_ws, void* grad, void* loss_total, hipStream_t stream,
                    const AdamArgs* adam) {
  return finalize_rows<T, NJ>((double*)rows_ws, row_blocks(batch), (T*)grad, (T*)loss_total, stream, m, adam);
}
template <typename T, int NJ>
int launch_step_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, long long batch,
                      void* x_next, long long ld_next, const void* wit, const double* pdirs, hipStream_t stream) {
  long long blocks = (batch + kIPW - 1) / kIPW;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((gen_simulate_kernel<T, NJ, true>), dim3((int)blocks), dim3(kWave), 0, stream, general_desc(m), m->opts[dtype],
                     (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, batch, 1LL, (T*)x_next,
                     ld_next, 0LL, 0, (int*)nullptr, (const T*)wit, pdirs);
  return dpll_check_launch("gen_simulate_kernel (mesh)");
}
template <typename T, int NJ>
int launch_step_backward_items(const dpll_model* m, const dpll_params_t* p, const void* x, long long ld_x, const void* gx, long long ld_g,
                               long long batch, void* rows_ws, void* grad_x, long long ld_gx, const void* wit, void* rbar,
                               const double* pdirs, hipStream_t stream) {
  const int rows = row_blocks(batch);
  hipLaunchKernelGGL((gen_step_backward_kernel<T, NJ, true>), dim3(rows + 1), dim3(kWave), 0, stream, general_desc(m), m->opts[DPLL_F64],
                     (const T*)p->theta, (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, (const T*)gx, ld_g, batch,
                     (double*)rows_ws, (T*)grad_x, ld_gx, (const T*)wit, (T*)rbar, pdirs);
  return dpll_check_launch("gen_step_backward_kernel (mesh)");
}
template <typename T, int NJ>
int launch_terms_items(const dpll_model* m, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm, void* M,
                       void* J, void* phi, void* a, const void* wit, const double* pdirs, hipStream_t stream) {
  if (!J) return dpll_fail(-1, "dpll_terms_mesh (general build): the J output is required%s");
  const long long blocks = (batch + kWave - 1) / kWave;
  hipLaunchKernelGGL((gen_terms_kernel<T, NJ, true>), dim3((int)blocks), dim3(kWave), 0, stream, general_desc(m), (const T*)p->theta,
                     (const T*)p->friction, (const T*)p->lengths, (const T*)x, ld_x, batch, (T*)Dm, (T*)M, (T*)J, (T*)phi, (T*)a,
                     (const T*)wit, pdirs);
  return dpll_check_launch("gen_terms_kernel (mesh)");
}

#define DPLL_GENMESH_DISPATCH(FN, ...)                                                        \
  do {                                                                                        \
    const int nj = m->desc.n_joints;                                                          \
    if (dtype == DPLL_F32 && nj == 0) return FN<float, 0>(__VA_ARGS__);                       \
    if (dtype == DPLL_F32 && nj == 1) return FN<float, 1>(__VA_ARGS__);                       \
    if (dtype == DPLL_F32 && nj == 2) return FN<float, 2>(__VA_ARGS__);                       \
    if (dtype == DPLL_F32 && nj == 3) return FN<float, 3>(__VA_ARGS__);                       \
    if (dtype == DPLL_F64 && nj == 0) return FN<double, 0>(__VA_ARGS__);                      \
    if (dtype == DPLL_F64 && nj == 1) return FN<double, 1>(__VA_ARGS__);                      \
    if (dtype == DPLL_F64 && nj == 2) return FN<double, 2>(__VA_ARGS__);                      \
    if (dtype == DPLL_F64 && nj == 3) return FN<double, 3>(__VA_ARGS__);                      \
    return dpll_fail(-2, "%s: the general build covers 0 to 3 joints", #FN);                   \
  } while (0)

}  // namespace

namespace dpll_genmesh {

int wit_per_item() { return kWitPerItem; }

int surface_directions(int dtype, void* out, hipStream_t stream) {
  if (dtype == DPLL_F64) hipLaunchKernelGGL(surface_dirs_kernel<double>, dim3(1), dim3(512), 0, stream, (double*)out);
  else hipLaunchKernelGGL(surface_dirs_kernel<float>, dim3(1), dim3(512), 0, stream, (float*)out);
  return dpll_check_launch("surface_dirs_kernel");
}

int queries(const dpll_model* m, int dtype, const QueryPlan& plan, const void* x, long long ld_x, long long batch, hipStream_t stream) {
  DPLL_GENMESH_DISPATCH(launch_queries, m, plan, x, ld_x, batch, stream);
}

long long workspace_bytes(const dpll_model* m, long long batch) {
  const long long rows = row_blocks(batch);
  long long doubles = 0;
  switch (m->desc.n_joints) {
    case 0: doubles = (rows + folded_rows(rows)) * GD<double, 0>::PI + GD<double, 0>::CHAIN; break;
    case 1: doubles = (rows + folded_rows(rows)) * GD<double, 1>::PI + GD<double, 1>::CHAIN; break;
    case 2: doubles = (rows + folded_rows(rows)) * GD<double, 2>::PI + GD<double, 2>::CHAIN; break;
    default: doubles = (rows + folded_rows(rows)) * GD<double, 3>::PI + GD<double, 3>::CHAIN; break;
  }
  return doubles * (long long)sizeof(double);
}

int loss_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* xp, long long ld_xp,
               long long batch, const void* weights, double scale, void* loss_out, void* force, int32_t* iters, void* rows,
               int want_grad, const void* wit, void* rbar, const double* pdirs, hipStream_t stream) {
  DPLL_GENMESH_DISPATCH(launch_loss_items, m, dtype, p, x, ld_x, xp, ld_xp, batch, weights, scale, loss_out, force, iters, rows,
                        want_grad, wit, rbar, pdirs, stream);
}

int finalize(const dpll_model* m, int dtype, long long batch, void* rows, void* grad, void* loss_total, hipStream_t stream,
             const AdamArgs* adam) {
  DPLL_GENMESH_DISPATCH(launch_finalize, m, batch, rows, grad, loss_total, stream, adam);
}

int step_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* x_next,
               long long ld_next, const void* wit, const double* pdirs, hipStream_t stream) {
  DPLL_GENMESH_DISPATCH(launch_step_items, m, dtype, p, x, ld_x, batch, x_next, ld_next, wit, pdirs, stream);
}

int step_backward_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, const void* gx,
                        long long ld_g, long long batch, void* rows, void* grad_x, long long ld_gx, const void* wit, void* rbar,
                        const double* pdirs, hipStream_t stream) {
  DPLL_GENMESH_DISPATCH(launch_step_backward_items, m, p, x, ld_x, gx, ld_g, batch, rows, grad_x, ld_gx, wit, rbar, pdirs, stream);
}

int terms_items(const dpll_model* m, int dtype, const dpll_params_t* p, const void* x, long long ld_x, long long batch, void* Dm,
                void* M, void* J, void* phi, void* a, const void* wit, const double* pdirs, hipStream_t stream) {
  DPLL_GENMESH_DISPATCH(launch_terms_items, m, p, x, ld_x, batch, Dm, M, J, phi, a, wit, pdirs, stream);
}

}  // namespace dpll_genmesh
