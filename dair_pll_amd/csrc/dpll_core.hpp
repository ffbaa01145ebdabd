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
//
// Since round 5 this header is an umbrella over five parts, so that a change to one of them rebuilds only the translation units
// that use it (csrc/Makefile tracks the includes with -MMD):
//   dpll_terms.hpp    helpers, inertial parameterisation, chain kinematics, dense SPD solves, box / ground contact, cone projection
//   dpll_solver.hpp   the cone QP (sap_newton)
//   dpll_contact.hpp  Derived parameters, compute_terms, contact geometry incl. body-body candidates
//   dpll_loss.hpp     ContactNets loss + adjoint
//   dpll_step.hpp     simulation step, its adjoints, the chain to the learnable parameters
#pragma once

#include "dpll_terms.hpp"
#include "dpll_solver.hpp"
#include "dpll_contact.hpp"
#include "dpll_loss.hpp"
#include "dpll_step.hpp"
