/* dpll.h -- C ABI of libdpll_hip.so: the MI355X (gfx950) contact-dynamics hot path of dair_pll.
 *
 * Drop-in boundary.  dair_pll is pure Python and has no FFI of its own; the functions below are what a
 * binding for the hot path would call, one per method of the reference's System / Integrator surface
 * (file:line under /root/reference/dair_pll):
 *
 *   dpll_contactnets_loss   <- MultibodyLearnableSystem.contactnets_loss   multibody_learnable_system.py:104-197
 *                              + the backward pass torch autograd would run (experiment.py:359)
 *   dpll_step               <- VelocityIntegrator.step(sim_step/forward_dynamics)  integrator.py:153-162,
 *                              multibody_learnable_system.py:199-313
 *   dpll_simulate           <- Integrator.simulate                          integrator.py:75-99
 *   dpll_terms              <- MultibodyTerms.forward                       multibody_terms.py:584-609
 *   dpll_model_create       <- MultibodyLearnableSystem.__init__ (the facts Drake extracts from the URDF,
 *                              multibody_learnable_system.py:51-80, drake_utils.py:248-335)
 *
 * Conventions: plain pointers and sizes only.  Every data pointer is a DEVICE pointer owned by the caller;
 * nothing is allocated, freed or synchronised inside a call, so calls may be captured into a hipGraph.
 * `stream` is a hipStream_t passed as void*.  Batched arrays are row-major (B, n) with an explicit row
 * stride in elements.  dtype selects float32 or float64 for every array of the call.  Return value: 0 on
 * success, negative on error with a message available from dpll_last_error() (no exceptions cross the ABI).
 *
 * State layout (dair_pll/state_space.py:412-424):  x = [quat wxyz, p_world(3), joint angles |
 * omega_body(3), v_world(3), joint rates], n_x = 13 + 2 n_joints.
 * Parameter layout: theta (n_bodies, 10) log-Cholesky inertial parameters (inertia.py:206-234);
 * friction (1 + n_geoms,) with the ground first (multibody_terms.py:314-317, drake_utils.py:280-288);
 * lengths (n_geoms, 3) box half lengths (geometry.py:367-403), a sphere's radius in [g][0] (geometry.py:415-456);
 * n_geoms = n_bodies for the fast builds.
 */
#ifndef DPLL_H_
#define DPLL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPLL_MAX_JOINTS 3
#define DPLL_MAX_BODIES 4
#define DPLL_MAX_GEOMS 3
#define DPLL_MAX_PAIRS 4                         /* body-body collision candidates */
/* geometry slots of the general build: behind the geometries one group of 4 contact slots, slot p = candidate p */
#define DPLL_GEN_SLOTS (DPLL_MAX_GEOMS + 1)

enum dpll_dtype { DPLL_F32 = 0, DPLL_F64 = 1 };
/* DPLL_INERTIA_COMPOSED (general and forest builds): the rows of dpll_params_t.theta are not theta-format parameters but the
 * bodies' inertial vectors [m, m c (3), I_o (xx, yy, zz, xy, xz, yz)] themselves -- what dpll_weld_compose below writes for a
 * model whose URDF welds links together; `grad` then holds d loss / d (those vectors), which dpll_weld_compose_backward chains
 * to the links' own parameters. */
enum dpll_inertia_mode { DPLL_INERTIA_REFERENCE_LITERAL = 0, DPLL_INERTIA_PHYSICAL = 1, DPLL_INERTIA_COMPOSED = 2 };
#define DPLL_MAX_POLYGON_VERTICES 8
#define DPLL_GEOM_BLOCK (3 * DPLL_MAX_POLYGON_VERTICES) /* numbers per geometry in the general build's `lengths` block */

/* DPLL_GEOM_MESH: a learned convex shape (DeepSupportConvex, geometry.py:255-364) of the GENERAL build: its network's
 * parameters travel in the dpll_mesh_params_t array of the *_mesh entry points; its block of `lengths` is padding */
enum dpll_geom_kind { DPLL_GEOM_BOX = 0, DPLL_GEOM_SPHERE = 1, DPLL_GEOM_POLYGON = 2, DPLL_GEOM_MESH = 3 };
/* a prismatic joint moves its body along joint_axis by the joint coordinate (metres); its velocity column is the axis */
enum dpll_joint_kind { DPLL_JOINT_REVOLUTE = 0, DPLL_JOINT_PRISMATIC = 1 };

/* One floating-base tree of revolute joints with convex collision geometries against the ground half-space at z = 0:
 * what Drake extracts from the URDF for MultibodyTerms (multibody_terms.py:328-382, drake_utils.py:248-335).
 * n_geoms = 0 selects the two fast builds (a serial chain with one box per body, geometry g on body g: the cube and
 * elbow systems); n_geoms > 0 the general build: any parent tree, 1..DPLL_MAX_GEOMS boxes / spheres / polygons (vertex
 * sets, geometry.py:220-252) on any bodies. */
typedef struct dpll_model_desc {
  int32_t n_joints;
  int32_t inertia_mode;
  double dt;
  double gravity_z;
  double joint_origin[DPLL_MAX_JOINTS][3]; /* joint j+1: origin in the parent body frame */
  double joint_axis[DPLL_MAX_JOINTS][3];   /* unit axis (see joint_kind) */
  double geom_origin[DPLL_MAX_GEOMS][3];   /* collision geometry origin in the frame of its body (see geom_rot) */
  int32_t parent[DPLL_MAX_JOINTS];         /* general build: parent body of body j + 1 */
  int32_t n_geoms;                         /* 0: fast builds */
  int32_t geom_body[DPLL_MAX_GEOMS];
  int32_t geom_kind[DPLL_MAX_GEOMS];       /* dpll_geom_kind */
  int32_t geom_nverts[DPLL_MAX_GEOMS];     /* DPLL_GEOM_POLYGON: vertices, 4 .. DPLL_MAX_POLYGON_VERTICES */
  /* body-body collision candidates beyond the ground pairs (ContactTerms.collision_candidates, multibody_terms.py:286-297;
   * GeometryCollider.collide_mesh_mesh, geometry.py:585-643: ONE contact per pair along a direction found by an exact
   * search in place of fcl): geometry pair_a[p] against geometry pair_b[p], on different bodies */
  int32_t n_pairs;
  int32_t pair_a[DPLL_MAX_PAIRS];
  int32_t pair_b[DPLL_MAX_PAIRS];
  /* General build, URDFs whose <origin>s carry a rotation (rpy).  The kernels' body frames all coincide at zero joint
   * angles; joint_origin / joint_axis are given in those frames.  body_rot[b] takes a vector from the frame the inertial
   * parameters of body b are expressed in (the URDF's link frame, as the reference's parameters are) to the kernels'
   * frame of body b; geom_rot[g] is the orientation of geometry g's own frame in the kernels' frame of its body, and
   * geom_origin[g] is then given in that geometry frame.  rotated: bit 0 = some body_rot, bit 1 = some geom_rot is
   * not the identity (both arrays must hold identities otherwise). */
  int32_t rotated;
  double body_rot[DPLL_MAX_BODIES][3][3];
  double geom_rot[DPLL_MAX_GEOMS][3][3];
  int32_t joint_kind[DPLL_MAX_JOINTS];     /* general build: dpll_joint_kind of joint j + 1 (the fast builds: revolute) */
  /* Actuation (general build): the reference's lagrangian_forces(q, v, u, inertia) carries B u (multibody_terms.py:142-146,
   * 235-236) -- input k of `u` (dpll_params_t.u, n_u numbers per item) is a generalized force on the coordinate of joint
   * act_joint[k] + 1 (one JointActuator per <transmission> of the URDF, in file order; gear ratio 1).  n_u = 0: no actuators
   * (every system of the reference).  The fast builds take n_u = 0 only. */
  int32_t n_u;
  int32_t act_joint[DPLL_MAX_JOINTS];
  int32_t reserved;
} dpll_model_desc_t;

/* ---- the forest build: several models in one system, any tree, limits an order of magnitude above the general build's ----
 * What the reference's MultibodyLearnableSystem(init_urdfs: Dict[str, str], ...) describes (multibody_learnable_system.py:51-54):
 * every URDF one model with a floating (or, welded to the world, fixed) base, state = the models' states one after the other
 * (ProductSpace of FloatingBaseSpace / FixedBaseSpace, drake_utils.py:309-335, state_space.py:650-730), collision candidates
 * between any two geometries Drake does not filter -- inside a model and across models.  One wave works on one item with
 * everything in LDS (csrc/dpll_forest.hip): boxes, spheres, polygons; learned shapes stay on the general build. */
#define DPLL_FOREST_MAX_BODIES 16
#define DPLL_FOREST_MAX_GEOMS 12
#define DPLL_FOREST_MAX_PAIRS 16
#define DPLL_FOREST_MAX_CONTACTS 64   /* 4 per box / polygon, 1 per sphere, 1 per candidate */
#define DPLL_FOREST_MAX_V 32          /* generalized velocities */
/* joint kinds of a forest body beyond dpll_joint_kind: the root of a model, free in the world (q: quaternion wxyz + position,
 * v: omega_body + v_world) or welded to it (no coordinates) */
#define DPLL_JOINT_FLOATING 2
#define DPLL_JOINT_FIXED 3

typedef struct dpll_forest_desc {
  int32_t n_bodies, n_geoms, n_pairs, n_contacts, n_q, n_v, inertia_mode, rotated, max_depth;
  int32_t n_u;                                 /* actuators (see act_body below); 0: dpll_params_t.u must be NULL */
  double dt, gravity_z;
  int32_t parent[DPLL_FOREST_MAX_BODIES];      /* -1 for a root, else a body listed before this one */
  int32_t joint_kind[DPLL_FOREST_MAX_BODIES];  /* dpll_joint_kind | DPLL_JOINT_FLOATING | DPLL_JOINT_FIXED */
  int32_t q_index[DPLL_FOREST_MAX_BODIES];     /* first coordinate of the body's joint in q */
  int32_t v_index[DPLL_FOREST_MAX_BODIES];     /* first velocity of the body's joint in v (after its parent's) */
  int32_t depth[DPLL_FOREST_MAX_BODIES];       /* joints between the body and its root */
  double joint_origin[DPLL_FOREST_MAX_BODIES][3]; /* in the parent's frame (a fixed root: in the world); frames as in dpll_model_desc_t */
  double joint_axis[DPLL_FOREST_MAX_BODIES][3];
  double body_rot[DPLL_FOREST_MAX_BODIES][3][3];
  int32_t dof_body[DPLL_FOREST_MAX_V];         /* velocity i belongs to the joint of this body */
  int32_t geom_body[DPLL_FOREST_MAX_GEOMS], geom_kind[DPLL_FOREST_MAX_GEOMS], geom_nverts[DPLL_FOREST_MAX_GEOMS];
  double geom_origin[DPLL_FOREST_MAX_GEOMS][3];
  double geom_rot[DPLL_FOREST_MAX_GEOMS][3][3];
  int32_t pair_a[DPLL_FOREST_MAX_PAIRS], pair_b[DPLL_FOREST_MAX_PAIRS];
  /* contact c, in the reference's order: witness contact_slot[c] of geometry contact_geom[c] against the ground -- the 4 (a
   * sphere's 1) of every geometry whose body can move, in geometry order; a geometry on a body welded to the world (joint kind
   * FIXED: the root of a fixed-base model) is anchored like the ground itself and lists none (Drake filters anchored-anchored
   * candidates, drake_utils.py:178-184) --, then (contact_geom[c] = -1) the contact of candidate contact_slot[c]; no candidate
   * joins two anchored geometries */
  int32_t contact_geom[DPLL_FOREST_MAX_CONTACTS], contact_slot[DPLL_FOREST_MAX_CONTACTS];
  /* actuator k (a URDF <transmission>, in the plant's order: the models one after the other, each in file order) drives the
   * revolute / prismatic joint of body act_body[k]: column k of dpll_params_t.u is added to that joint's generalized force --
   * the B u of lagrangian_forces (multibody_terms.py:142-146) */
  int32_t act_body[DPLL_FOREST_MAX_V];
} dpll_forest_desc_t;

typedef struct dpll_solver_opts {
  int32_t max_iter;  /* Newton iterations */
  int32_t max_ls;    /* line-search evaluations per iteration */
  double tol;        /* Newton decrement / (1 + |y|_M) */
  double stall_tol;  /* a decrement that stopped halving ends the solve only below this */
  double ls_tol;     /* |l'(alpha)| / |l'(0)| */
  /* continuation in the regularisation eps: start at eps * stage_factor^(n_stages-1), divide by stage_factor when
   * a stage reached stage_tol or stage_max_iter iterations; n_stages = 1 disables it */
  int32_t n_stages;
  int32_t stage_max_iter;
  double stage_factor;
  double stage_tol;
  double stage_ls_tol;    /* line-search tolerance and probe cap of the non-final stages */
  int32_t stage_max_ls;
  int32_t fast_ls;        /* probes per iteration while the decrement keeps falling; 0 = always the full search */
  int32_t warm_start;     /* loss solve starts from the observed velocity jump dv instead of 0 */
  int32_t wide;           /* loss-kernel build: -1 = by batch size, 0 = one lane per contact, 1 = one lane per item */
  double loss_stage_factor; /* continuation schedule of the loss solve (eps 1e-3) when it differs from the dynamics solve's */
  int32_t loss_n_stages;    /* 0 = n_stages / stage_factor for both solves; like n_stages at most 8 unless portfolio = 1 */
  int32_t f64_refine;       /* DPLL_F64 solves: 1 = float iterations refined in double to `tol` (default), 0 = all double */
  int32_t mesh_gemm;        /* DPLL_F32 mesh pipeline, form of the ICNN GEMM kernels.  4 (default) = two fp16 planes, the low one scaled by
                             * 2^11: x = h + l to 2^-24, so the three products per k-step are f32-grade, at the cost of the 2-plane forms;
                             * adjoint operands are scaled into fp16's range by powers of two taken from the data; weights must stay
                             * below 2^14 (|input_weights.0| below 2^12) -- beyond, every solve of the launch is invalidated (NaN support
                             * points), never a finite wrong number.  0 = f32 MFMA (v_mfma_f32_32x32x2_f32; one wave per SIMD, pipelined),
                             * 1 = f32 MFMA, the 8-wave kernels of rounds 1-4, 2 / 3 = bf16 matrix cores on 2 / 3 bf16 planes (2: products
                             * to 2^-16) */
  /* Racing continuation schedules (loss solve, lane-per-contact builds): a launch that leaves SIMDs idle gives every item
   * `portfolio` copies of its lane group; copy 0 runs the schedule above, copy v >= 1 runs (race_stages[v-1],
   * race_factor[v-1]) with race_flags[v-1] (1 = warm start, 2 = full Newton steps only: no line search, so none of its
   * steps sends the wave through the fall-back code); the item is finished when ANY copy has met `tol` (copies >= 1: with
   * finite numbers) and the finished copy with the lowest index supplies every output -- deterministic, and in iterations
   * never behind the schedule above alone.
   * 0 = chosen from the model and the batch size (four copies where they fit a 16-lane row -- cube; float elbow, on a build
   * with two contacts per lane -- and keep the launch within one wave per SIMD, else none), 1 = off, 2, 4. */
  int32_t portfolio;
  int32_t race_stages[3];
  int32_t race_flags[3];
  double race_factor[3];
} dpll_solver_opts_t;

typedef struct dpll_model dpll_model_t;

/* Learnable parameters of one call: device pointers, caller owned, never written. */
typedef struct dpll_params {
  const void* theta;    /* (n_bodies, 10) */
  const void* friction; /* fast builds (1 + n_bodies,); general build (1 + DPLL_GEN_SLOTS,): ground, the geometries, padding (any non-zero number).
                           A body-body candidate has no entry: its coefficient combines its two geometries' entries */
  const void* lengths;  /* fast builds (n_bodies, 3): length_params of the boxes.  General build (DPLL_GEN_SLOTS, DPLL_GEOM_BLOCK),
                           geometry g's block: box length_params (3) | sphere length_param (1) | polygon vertices
                           (geom_nverts, 3) row-major; the rest of a block, and the whole block behind the geometries', is padding
                           (its gradient comes back zero) */
  const void* u;        /* actuation inputs of the call, (batch, n_u) with row stride ld_u elements, dtype of the call: the u of
                           contactnets_loss(x, u, x_plus) / forward_dynamics(q, v, u) (multibody_learnable_system.py:104, 199).
                           NULL = no actuation (required when the model has n_u = 0; rollouts -- dpll_simulate with steps > 1 --
                           run unactuated as the reference's sim_step does, :311) */
  int64_t ld_u;
} dpll_params_t;

const char* dpll_last_error(void);
int dpll_abi_version(void);

int dpll_model_create(const dpll_model_desc_t* desc, dpll_model_t** out);
void dpll_model_destroy(dpll_model_t* model);
/* default solver settings are chosen per dtype at creation; this overrides them */
int dpll_model_set_solver(dpll_model_t* model, int dtype, const dpll_solver_opts_t* opts);
int dpll_model_get_solver(const dpll_model_t* model, int dtype, dpll_solver_opts_t* opts);

int dpll_n_x(const dpll_model_t* model);          /* 13 + 2 n_joints */
/* A model on the forest build.  The handle works with every entry point below that takes a model handle (loss, step,
 * simulate, step backward, terms, the fused training step with ar = NULL; not the *_mesh and allreduce-fused ones).
 * Layouts: state rows (n_q + n_v); parameters [theta (n_bodies, 10) | friction (1 + n_geoms) | lengths (n_geoms,
 * DPLL_GEOM_BLOCK)]; forces / phi / J / D over the model's n_contacts contacts in the reference's order (no padding slots).
 * racing copies: none.  Creation touches no device (it works in a process without a GPU); the FIRST compute call on a handle
 * places the description in device memory (one hipMalloc + hipMemcpy, and hipFuncSetAttribute for arenas over 48 KB): make
 * one call before capturing the handle's launches in a hipGraph; every later call allocates nothing and is capturable. */
int dpll_forest_model_create(const dpll_forest_desc_t* desc, dpll_model_t** out);

int dpll_n_contacts(const dpll_model_t* model);   /* fast builds 4 n_bodies; general build 4 DPLL_GEN_SLOTS contact SLOTS: slot
                                                     4 g + s = witness s of geometry g (a sphere: s = 0 only), slot
                                                     4 DPLL_MAX_GEOMS + p = body-body candidate p; the others are masked */
int dpll_param_count(const dpll_model_t* model);  /* layout [theta | friction | lengths] as in dpll_params_t: fast builds
                                                     10 n_b + (1 + n_b) + 3 n_b, general build 10 n_b + (1 + DPLL_GEN_SLOTS) + DPLL_GEN_SLOTS DPLL_GEOM_BLOCK */

/* bytes of scratch dpll_contactnets_loss needs for a batch of `batch` items (gradient partial sums) */
int64_t dpll_workspace_bytes(const dpll_model_t* model, int64_t batch);

/* Racing copies per item (dpll_solver_opts_t.portfolio) that a launch of `batch` items would run with the model's current
 * solver settings: what = 0 the loss launch (dpll_contactnets_loss and its variants), 1 the rollout / step launch
 * (dpll_simulate, dpll_step), 4 the loss launch of dpll_contactnets_loss_mesh (a single body with a learned shape races like
 * the box cube).  1 = no copies (always: general and forest build).  -1 on a bad argument.
 * The SHAPE of the loss launch (specialised builds only, else -1): what = 2 its item workgroups (= partial rows; the grid has
 * one more, which writes the chain matrix), what = 3 the lanes of one copy of an item (1: the wide build). */
int dpll_racing_copies(const dpll_model_t* model, int dtype, int64_t batch, int what);

/* ContactNets loss of `batch` transitions (x -> x_plus), forward and backward in one pass.
 *   x, x_plus   (batch, n_x), row strides ld_x / ld_xp elements
 *   weights     optional (batch,) upstream gradient d(total)/d(loss_i); the effective weight of item i is
 *               scale * weights[i] (scale alone when weights is NULL), e.g. scale = 1/batch for .mean()
 *   loss        optional (batch,) per-item loss, as the reference returns it
 *   grad        optional (dpll_param_count,) <- sum_i weight_i * d loss_i / d params; NULL = forward only
 *   loss_total  optional (1,) <- sum_i weight_i * loss_i   (requires grad != NULL)
 *   force       optional (batch, 3 n_contacts) contact impulses in the reference's order
 *               [normals | (t_x, t_y) per contact] (multibody_terms.py:415-426); contacts of one geometry
 *               appear in this library's corner order (the reference's topk order is unspecified)
 *   iters       optional (batch,) int32 Newton iterations used
 *   workspace   >= dpll_workspace_bytes(model, batch) bytes, 16-byte aligned (may be NULL when grad is NULL)
 */
int dpll_contactnets_loss(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x,
                          int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, const void* weights,
                          double scale, void* loss, void* grad, void* loss_total, void* force, int32_t* iters,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* Measuring utility (synchronises; not for graph capture), HIP events on `stream`: `reps` loss kernels back to back
 * between two events give the average loss-kernel launch duration; `reps` (loss, finalize) pairs between two more
 * give the finalize kernel as the difference (an event between every two kernels would add microseconds of its own).
 * Milliseconds.  Same arguments as dpll_contactnets_loss with uniform weights. */
int dpll_profile_contactnets_loss(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x,
                                  int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, double scale,
                                  void* grad, void* workspace, int64_t workspace_bytes, void* stream, int32_t reps,
                                  float* ms_loss_kernel, float* ms_finalize_kernel);

/* One VelocityIntegrator step: x (batch, n_x) -> x_next (batch, n_x). */
int dpll_step(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x, int64_t ld_x,
              int64_t batch, void* x_next, int64_t ld_next, int32_t* iters, void* stream);

/* Backward of dpll_step: given grad_x_next (batch, n_x) = d(total)/d x_next, writes grad (dpll_param_count,) =
 * d(total)/d params and, when grad_x is not NULL, grad_x (batch, n_x) = d(total)/d x (row stride ld_gx; the four
 * quaternion components are independent variables, as they are for torch autograd in the reference).  This is what
 * back-propagating a prediction loss through forward_dynamics / Integrator.simulate needs (experiment.py:292-320):
 * the parameter gradient of every step plus the state adjoint that carries the loss from step t + 1 to step t.  The
 * cone solve is differentiated implicitly at its optimum (dair_pll delegates that to sappy).  Recomputes the forward
 * pass; workspace as for dpll_contactnets_loss.  Box geometry only. */
int dpll_step_backward(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x, int64_t ld_x,
                       const void* grad_x_next, int64_t ld_g, int64_t batch, void* grad, void* grad_x, int64_t ld_gx,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* Integrator.simulate: x0 (batch, n_x) -> traj (batch, steps + 1, n_x) contiguous, traj[:, 0] = x0. */
int dpll_simulate(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x0, int64_t ld_x,
                  int64_t batch, int64_t steps, void* traj, void* stream);

/* ---- mesh geometry: DeepSupportConvex / HomogeneousICNN (geometry.py:255-325, deep_support_function.py:125-266),
 * depth 2, width 256, ONE NETWORK PER BODY of a cube / elbow model (contactnets_cube_mesh.urdf,
 * contactnets_elbow_mesh.urdf): every `mesh` argument below points to an array of n_bodies dpll_mesh_params_t, in body
 * order.  Raw (signed) parameters, caller owned:
 *   hidden_weight (256, 256)  network.hidden_weights.0     input_weight0 / input_weight1 (3, 256)  network.input_weights.{0,1}
 *   output_weight (256,)      network.output_weight        perturbations (4, 3) fixed buffer, row 0 zero
 * Gradient layout of the mesh entry points: [theta(10 n_bodies) | friction(1 + n_bodies) | then per body: hidden_weight |
 * input_weight0 | input_weight1 | output_weight] = dpll_mesh_param_count() numbers.  The box `lengths` pointer of
 * dpll_params_t is ignored. */
typedef struct dpll_mesh_params {
  const void* hidden_weight;
  const void* input_weight0;
  const void* input_weight1;
  const void* output_weight;
  const void* perturbations;
} dpll_mesh_params_t;

int dpll_mesh_param_count(const dpll_model_t* model);
int64_t dpll_mesh_workspace_bytes(const dpll_model_t* model, int64_t batch, int dtype);

/* dpll_contactnets_loss with the body's collision shape given by the network: same arguments and outputs. */
int dpll_contactnets_loss_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params,
                               const dpll_mesh_params_t* mesh, const void* x, int64_t ld_x, const void* x_plus,
                               int64_t ld_xp, int64_t batch, const void* weights, double scale, void* loss, void* grad,
                               void* loss_total, void* force, int32_t* iters, void* workspace, int64_t workspace_bytes,
                               void* stream);

/* dpll_step_backward with the network shape: grad has the layout of dpll_contactnets_loss_mesh's
 * ([theta | friction | per body: hidden | input0 | input1 | output weights], dpll_mesh_param_count entries); the support point is
 * piecewise constant in the state (LeakyReLU network), so grad_x needs no network Jacobian. */
int dpll_step_backward_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                            const void* x, int64_t ld_x, const void* grad_x_next, int64_t ld_g, int64_t batch, void* grad,
                            void* grad_x, int64_t ld_gx, void* workspace, int64_t workspace_bytes, void* stream);

/* dpll_terms with the network shape (support points of the CURRENT state as witnesses). */
int dpll_terms_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                    const void* x, int64_t ld_x, int64_t batch, void* delassus, void* M, void* J, void* phi, void* a,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* Measuring utility, mesh pipeline: `reps` calls of dpll_contactnets_loss_mesh (with grad) with HIP events on the
 * launch stream after each of its kernels; ms_kernels[7] = average duration of
 * {prep, fwd1, fwd2, loss, bwd1, bwd2, reduce}.  Synchronises. */
int dpll_profile_contactnets_loss_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params,
                                       const dpll_mesh_params_t* mesh, const void* x, int64_t ld_x, const void* x_plus,
                                       int64_t ld_xp, int64_t batch, double scale, void* grad, void* workspace,
                                       int64_t workspace_bytes, void* stream, int32_t reps, float* ms_kernels);

/* dpll_step with the network shape (one step per call: the support points depend on the current state). */
int dpll_step_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                   const void* x, int64_t ld_x, int64_t batch, void* x_next, int64_t ld_next, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* dpll_simulate with the network shapes: x0 (batch, n_x) -> traj (batch, steps + 1, n_x) contiguous, traj[:, 0] = x0;
 * steps >= 1.  The weights are prepared once, then every step enqueues the networks' forward kernels on the current state
 * and the one-step kernel (no host work between the steps). */
int dpll_simulate_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                       const void* x0, int64_t ld_x, int64_t batch, int64_t steps, void* traj, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* DeepSupportConvex.get_vertices for the ground-contact direction of every state and body: points (batch, 4 n_bodies, 3),
 * each body's four in its own frame. */
int dpll_mesh_support_points(const dpll_model_t* model, int dtype, const dpll_mesh_params_t* mesh, const void* x,
                             int64_t ld_x, int64_t batch, void* points, void* workspace, int64_t workspace_bytes,
                             void* stream);

/* MultibodyTerms.forward at (q, v) taken from x: delassus (batch, 3k, 3k), M (batch, n_v, n_v),
 * J (batch, 3k, n_v), phi (batch, k), a (batch, n_v); any output may be NULL. */
int dpll_terms(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x, int64_t ld_x,
               int64_t batch, void* delassus, void* M, void* J, void* phi, void* a, void* stream);

/* ---- one-shot all-reduce of the tiny [loss | gradients] vector over xGMI peer memory (one process per GPU).
 * The reference has no distributed code; this is the path's single exchange step.  Every rank
 * pushes its vector into every peer's uncached, IPC-shared receive buffer and sums what arrived in rank order
 * (bitwise identical on all ranks).  Setup: dpll_ar_create on every rank -> exchange the handles by any means
 * (e.g. torch.distributed.all_gather) -> dpll_ar_connect.  dpll_ar_allreduce launches one kernel on `stream`
 * and may be captured into a hipGraph; spins are bounded and a timeout is reported by dpll_ar_status. */
typedef struct dpll_ar dpll_ar_t;
int64_t dpll_ar_handle_bytes(void);
int dpll_ar_create(int rank, int world, void* handle_out, dpll_ar_t** out);
int dpll_ar_connect(dpll_ar_t* ar, const void* handles);
int dpll_ar_allreduce(dpll_ar_t* ar, int dtype, void* data, int n, void* stream);
int dpll_ar_status(dpll_ar_t* ar);
void dpll_ar_destroy(dpll_ar_t* ar);

/* dpll_contactnets_loss (forward + backward, no per-item outputs) whose last kernel also sums the row
 * [loss_total | grad] over the ranks of `ar` before writing it: data-parallel training's only collective rides in the
 * launch that produces the gradients (no kernel of its own).  Counts as one call of dpll_ar_allreduce on every
 * rank.  `scale` should already carry 1 / global batch. */
int dpll_contactnets_loss_allreduce(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x,
                                    int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, const void* weights,
                                    double scale, void* grad, void* loss_total, void* workspace, int64_t workspace_bytes,
                                    dpll_ar_t* ar, void* stream);

/* ---- one training step in two launches: the loss launch, and a finalize kernel that -- after summing the rows, chaining
 * them to the parameters and (ar != NULL) exchanging [loss | gradients] with the other ranks -- applies Adam to the
 * parameters in place (experiment.py:332-363 with the optimizer of :213-228: torch.optim.Adam without amsgrad; weight_decay
 * is added to the gradient as torch does).  The specialised builds (cube / elbow, box geometry).  All pointers are device
 * pointers of the call's dtype except `state`: three doubles [steps taken, beta1^steps, beta2^steps], {0, 1, 1} before the
 * first step, advanced by the call (the bias corrections need no pow on the device and the step count survives graph replay).  `params` MUST be the flat buffer dpll_params_t.theta points to, laid out [theta | friction | lengths]
 * (dpll_param_count entries): the next call then reads the updated parameters.  grad receives the (reduced) gradient the
 * update used, loss_total the (reduced) weighted loss. */
typedef struct dpll_adam {
  void* params;
  void* exp_avg;
  void* exp_avg_sq;
  double* state;
  double lr, beta1, beta2, eps, weight_decay;
} dpll_adam_t;

/* (Which builds: the specialised box builds with the exchange of `ar` inside the same kernel; the general and the forest build
 * with ar = NULL -- Adam in the kernel that chains the folded rows, entries of the flat buffer that are padding left alone; the
 * models with learned shapes through dpll_contactnets_train_step_mesh below.) */
int dpll_contactnets_train_step(const dpll_model_t* model, int dtype, const dpll_params_t* params, const void* x,
                                int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, const void* weights,
                                double scale, void* grad, void* loss_total, void* workspace, int64_t workspace_bytes,
                                dpll_ar_t* ar, const dpll_adam_t* adam, void* stream);

/* The same for models with learned shapes -- a cube / elbow with one per body (the specialised mesh builds) or a general tree
 * with learned shapes among its geometries: the kernel that reduces a network's weight gradients applies Adam to them, the head
 * ([theta | friction], general build: [theta | friction | lengths] with its padding left alone) is updated where it is chained;
 * adam.params is the flat buffer [head | network 0 (Wh, Wd0, Wd1, wout) | network 1 ...] params.theta and the mesh pointers
 * point into. */
int dpll_contactnets_train_step_mesh(const dpll_model_t* model, int dtype, const dpll_params_t* params, const dpll_mesh_params_t* mesh,
                                     const void* x, int64_t ld_x, const void* x_plus, int64_t ld_xp, int64_t batch, const void* weights,
                                     double scale, void* grad, void* loss_total, void* workspace, int64_t workspace_bytes,
                                     const dpll_adam_t* adam, void* stream);

/* ---- links welded together by `fixed` joints --------------------------------------------------------------------------------
 * Drake keeps a link that a `fixed` joint welds to another as a body of its own, and the reference learns one theta row per
 * Drake body (multibody_terms.py:161-207: inertial_parameters is (n_bodies_drake, 10)); the kernels' bodies are the links that
 * move against each other.  These two calls are the map between the two: row r (theta format, the link alone in its own
 * frame) belongs to kernel body host[r], whose inertial vector is
 *     iota_b = sum over rows r with host[r] == b of  X_r  theta_to_iota(theta_r, inertia_mode)
 * with X_r (10 x 10, row-major, double) the rigid transform of an inertial vector from the link's frame to the body's (the
 * identity for the body's own link; linear because [m, m c, I_o] transforms linearly).  inertia_mode: REFERENCE_LITERAL or
 * PHYSICAL, applied per row as the reference applies it per Drake body.  All pointers are device memory; n_rows <= 64.
 *   dpll_weld_compose:           theta_rows (n_rows, 10) -> iota (n_bodies, 10), the `theta` of a DPLL_INERTIA_COMPOSED model
 *   dpll_weld_compose_backward:  grad_iota (n_bodies, 10) = the theta block of a gradient row of such a model
 *                                -> grad_theta_rows (n_rows, 10); overwritten, or added to when accumulate != 0 */
#define DPLL_MAX_WELD_ROWS 64
int dpll_weld_compose(int dtype, int inertia_mode, int n_rows, int n_bodies, const int32_t* host, const double* transforms,
                      const void* theta_rows, void* iota, void* stream);
int dpll_weld_compose_backward(int dtype, int inertia_mode, int n_rows, int n_bodies, const int32_t* host, const double* transforms,
                               const void* theta_rows, const void* grad_iota, void* grad_theta_rows, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPLL_H_ */
