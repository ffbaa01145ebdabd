"""Independent (non-circular) checks of the oracle's rigid-body terms and QP solver.

The reference gets M(q), F(q, v) and the geometry kinematics from Drake symbolics and the QP
solution from sappy; neither is installed ("parity unpinned").  These tests pin the
oracle's restatement to physics instead:
  * M(q) == Hessian in v of a kinetic energy computed from positions only (autograd jvp);
  * free flight under a = M^-1 F conserves energy, horizontal momentum and angular momentum
    about the system centre of mass (RK4, perturbed inertial parameters);
  * d(phi)/dt == J_n v and contact-point velocities == J_t v / mu (finite differences);
  * the solver's answer satisfies the KKT conditions of the strictly convex cone QP.
"""
import os

import pytest
import torch

from conftest import ASSET_DIR, REFERENCE_DIR
from oracle import dpll_oracle as O

torch.set_default_dtype(torch.float64)
DT = 0.0068


# every tree topology the kernels take: serial chains of 0 / 1 / 2 joints, a branching tree, bodies without geometry,
# frames turned by an rpy, a prismatic joint
TREES = ['cube.urdf', 'elbow.urdf', 'chain3.urdf', 'vee.urdf', 'mace.urdf', 'clasp.urdf', 'gripper.urdf', 'crank.urdf',
         'pincer.urdf', 'slider.urdf']
# ... and every kind of contact: boxes, spheres, polygons against the ground, body-body pairs (box-box, polygon-sphere,
# the two arms of a branching tree)
CONTACTS = TREES + ['ballcube.urdf', 'wedge.urdf:polygon', 'clasp_ball.urdf:polygon', 'vee_pair.urdf']


def _system(name, mode='physical'):
    name, _, representation = name.partition(':')
    system = O.OracleSystem(os.path.join(ASSET_DIR, name), DT, inertia_mode=mode,
                            mesh_representation=representation or 'deep_support')
    gen = torch.Generator().manual_seed(3)
    # perturb so that com != origin and the inertia tensor is full
    system.theta = system.theta + 0.2 * (torch.rand(system.theta.shape, generator=gen) - 0.5)
    return system


def _random_state(system, gen, n):
    quat = torch.randn((n, 4), generator=gen)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    pos = torch.rand((n, 3), generator=gen) + torch.tensor([0., 0., 1.])
    joints = 3 * (torch.rand((n, system.n_joints), generator=gen) - 0.5)
    v = 4 * (torch.rand((n, system.n_v), generator=gen) - 0.5)
    return torch.cat((quat, pos, joints), -1), v


def _q_dot(system, q, v):
    quat_dot = 0.5 * O.quat_multiply(q[..., :4], torch.cat((torch.zeros_like(v[..., :1]), v[..., :3]), -1))
    return torch.cat((quat_dot, v[..., 3:]), -1)


def _body_inertias(system):
    inertia = O.theta_to_spatial_inertia(system.theta)
    mass = inertia[:, 0]
    com = inertia[:, 1:4]
    i_cm = O._inertia_matrix(inertia[:, 4:])
    if system.inertia_mode == 'physical':
        i_cm = i_cm * mass[:, None, None]
    return mass, com, i_cm


def _mechanics(system, q, v):
    """energy / momenta from POSITION kinematics only (R_b(q), o_b(q)) + autograd time derivatives."""
    mass, com, i_cm = _body_inertias(system)
    qd = _q_dot(system, q, v)

    def positions(qq):
        rot, org, _, _, _ = O.chain_kinematics(system.spec, qq)
        coms = torch.stack([org[b] + (rot[b] @ com[b].unsqueeze(-1)).squeeze(-1) for b in range(len(rot))], -2)
        return coms, torch.stack(rot, -3)

    (coms, rots), (com_vel, rot_dot) = torch.autograd.functional.jvp(positions, (q,), (qd,), create_graph=True)
    omega_mat = rot_dot @ rots.transpose(-1, -2)
    omega = torch.stack((omega_mat[..., 2, 1], omega_mat[..., 0, 2], omega_mat[..., 1, 0]), -1)
    i_world = rots @ i_cm @ rots.transpose(-1, -2)
    kinetic = 0.5 * (mass * (com_vel**2).sum(-1)).sum(-1) + \
        0.5 * (omega.unsqueeze(-2) @ i_world @ omega.unsqueeze(-1)).reshape(omega.shape[:-1]).sum(-1)
    potential = -(mass * O.GRAVITY_Z * coms[..., 2]).sum(-1)
    momentum = (mass.unsqueeze(-1) * com_vel).sum(-2)
    total_mass = mass.sum()
    centre = (mass.unsqueeze(-1) * coms).sum(-2) / total_mass
    ang = ((i_world @ omega.unsqueeze(-1)).squeeze(-1) + mass.unsqueeze(-1) * torch.cross(coms, com_vel, dim=-1)
           ).sum(-2) - torch.cross(centre, momentum, dim=-1)
    return kinetic, potential, momentum, ang


@pytest.mark.parametrize('urdf', TREES)
@pytest.mark.parametrize('mode', ['physical', 'reference_literal'])
def test_mass_matrix_is_kinetic_energy_hessian(urdf, mode):
    system = _system(urdf, mode)
    q, v = _random_state(system, torch.Generator().manual_seed(0), 3)
    M, _ = system.lagrangian_terms(q, v)
    for row in range(q.shape[0]):
        def energy(vv):
            return _mechanics(system, q[row:row + 1], vv.unsqueeze(0))[0].squeeze(0)
        hessian = torch.autograd.functional.hessian(energy, v[row])
        assert torch.allclose(hessian, M[row], atol=1e-12, rtol=1e-10)


@pytest.mark.parametrize('urdf', TREES)
def test_free_flight_conserves_energy_and_momenta(urdf):
    system = _system(urdf)
    q, v = _random_state(system, torch.Generator().manual_seed(1), 2)

    def deriv(qq, vv):
        with torch.no_grad():
            _, acc = system.lagrangian_terms(qq, vv)
        return _q_dot(system, qq, vv), acc

    def invariants(qq, vv):
        kinetic, potential, momentum, ang = _mechanics(system, qq, vv)
        return (kinetic + potential).detach(), momentum[..., :2].detach(), ang.detach()

    e0, p0, l0 = invariants(q, v)
    h = 2e-4
    for _ in range(250):
        k1q, k1v = deriv(q, v)
        k2q, k2v = deriv(q + 0.5 * h * k1q, v + 0.5 * h * k1v)
        k3q, k3v = deriv(q + 0.5 * h * k2q, v + 0.5 * h * k2v)
        k4q, k4v = deriv(q + h * k3q, v + h * k3v)
        q = q + h / 6 * (k1q + 2 * k2q + 2 * k3q + k4q)
        v = v + h / 6 * (k1v + 2 * k2v + 2 * k3v + k4v)
    e1, p1, l1 = invariants(q, v)
    assert (e1 - e0).abs().max() < 1e-10
    assert (p1 - p0).abs().max() < 1e-11
    assert (l1 - l0).abs().max() < 1e-11


@pytest.mark.parametrize('urdf', CONTACTS)
def test_contact_jacobian_matches_finite_differences(urdf):
    """d phi / dt = J_n v for every contact, including a body-body contact (its signed distance is the maximum of the
    separation over the direction, so its time derivative is the one at the maximising direction held fixed -- what
    the reference assumes when it treats fcl's direction as piecewise constant, geometry.py:597-600)"""
    system = _system(urdf)
    q, v = _random_state(system, torch.Generator().manual_seed(2), 4)
    if system.spec['pairs']:  # fold the joints so that the pair is near (and, in some rows, overlapping)
        gen = torch.Generator().manual_seed(4)
        trial = 6.2 * (torch.rand((400, system.n_joints), generator=gen) - 0.5)
        q_trial = torch.cat((q[:1, :7].expand(400, 7), trial), -1)
        gap = system.contact_terms(q_trial)[0][:, -1]
        near = torch.argsort(gap.abs())[:4]
        q = torch.cat((q[:, :7], trial[near]), -1)
    phi, J = system.contact_terms(q)
    k = phi.shape[-1]
    h = 1e-6
    q_dot = _q_dot(system, q, v)
    phi_p, _ = system.contact_terms(q + h * q_dot)
    phi_m, _ = system.contact_terms(q - h * q_dot)
    dphi = (phi_p - phi_m) / (2 * h)
    normal_speed = (J[..., :k, :] @ v.unsqueeze(-1)).squeeze(-1)
    ground = k - len(system.spec['pairs'])
    assert torch.allclose(dphi[..., :ground], normal_speed[..., :ground], atol=1e-7)
    # A body-body contact's Jacobian is NOT the time derivative of its signed distance for polytopes, in the reference
    # as here: the witness points are support VERTICES (geometry.py:627-629), not the nearest points of the two faces or
    # edges, and a vertex off the common normal moves along it when its body turns.  What does hold: phi is the exact
    # signed distance -- for rows that are apart, the distance between the two convex hulls by an independent solver
    for a_index, b_index in system.spec['pairs']:
        import numpy as np
        from scipy.optimize import minimize
        R_WC, p_WoCo_W, _ = O.geometry_kinematics(system.spec, q)
        va, ma = system.vertex_set(a_index)
        vb, mb = system.vertex_set(b_index)
        for row in range(q.shape[0]):
            if phi[row, -1] <= 1e-4:
                continue
            a_w = (va @ R_WC[row, a_index].T + p_WoCo_W[row, a_index]).numpy()
            b_w = (vb @ R_WC[row, b_index].T + p_WoCo_W[row, b_index]).numpy()
            na, nb = len(a_w), len(b_w)

            def gap(z):
                return ((z[:na] @ a_w - z[na:] @ b_w) ** 2).sum()
            z0 = np.concatenate((np.full(na, 1 / na), np.full(nb, 1 / nb)))
            cons = [{'type': 'eq', 'fun': lambda z: z[:na].sum() - 1}, {'type': 'eq', 'fun': lambda z: z[na:].sum() - 1}]
            best = minimize(gap, z0, bounds=[(0, 1)] * (na + nb), constraints=cons, method='SLSQP', options={'ftol': 1e-16, 'maxiter': 500})
            assert abs(np.sqrt(best.fun) - float(ma) - float(mb) - phi[row, -1].item()) < 1e-6


def test_reference_closed_forms_for_the_cube():
    """SURVEY 8a closed forms: M = [[I_Bo, m S(p) R^T], [m R S(p)^T, m I]],
    F = [-w x (I_Bo w) + m p x (R^T g); -m R (w x (w x p)) + m g]."""
    system = _system('cube.urdf')
    q, v = _random_state(system, torch.Generator().manual_seed(5), 5)
    mass, com, i_cm = _body_inertias(system)
    m, p = mass[0], com[0]
    i_o = i_cm[0] - m * O.skew(p) @ O.skew(p)
    R = O.quat_to_rot(q[:, :4])
    w = v[:, :3]
    g = torch.tensor([0., 0., O.GRAVITY_Z])
    top = torch.cat((i_o.expand(5, 3, 3), m * O.skew(p) @ R.transpose(-1, -2)), -1)
    bottom = torch.cat((m * R @ O.skew(p).t(), m * torch.eye(3).expand(5, 3, 3)), -1)
    M_closed = torch.cat((top, bottom), -2)
    rtg = (R.transpose(-1, -2) @ g.unsqueeze(-1)).squeeze(-1)
    f_ang = -torch.cross(w, (i_o @ w.unsqueeze(-1)).squeeze(-1), dim=-1) + m * torch.cross(p.expand(5, 3), rtg, dim=-1)
    wwp = torch.cross(w, torch.cross(w, p.expand(5, 3), dim=-1), dim=-1)
    f_lin = -m * (R @ wwp.unsqueeze(-1)).squeeze(-1) + m * g
    inertia = O.theta_to_spatial_inertia(system.theta).expand(5, 1, 10)
    assert torch.allclose(O.mass_matrix(system.spec, q, inertia, 'physical'), M_closed, atol=1e-14)
    assert torch.allclose(O.lagrangian_forces(system.spec, q, v, inertia, 'physical'),
                          torch.cat((f_ang, f_lin), -1), atol=1e-13)


@pytest.mark.parametrize('eps', [1e-3, 1e-4])
def test_solver_kkt_on_random_problems(eps):
    gen = torch.Generator().manual_seed(7)
    J = torch.randn((200, 12, 6), generator=gen)
    q = torch.randn((200, 12), generator=gen)
    f = O.sap_solve(J, q, eps)
    kkt = O.kkt_residuals(J, q, eps, f)
    size = 1 + f.abs().amax(-1)  # residuals relative to the force magnitude
    assert (kkt['primal'] / size).max() < 1e-13
    assert (kkt['dual'] / size).max() < 1e-10
    assert (kkt['complementarity'] / size**2).max() < 1e-10


def test_theta_round_trip():
    gen = torch.Generator().manual_seed(11)
    theta = torch.randn((16, 10), generator=gen) * 0.5
    pi_o = O.theta_to_pi_o(theta)
    assert torch.allclose(O.pi_o_to_theta(pi_o), theta, atol=1e-10)
    assert torch.allclose(O.pi_cm_to_pi_o(O.pi_o_to_pi_cm(pi_o)), pi_o, atol=1e-12)


@pytest.mark.reference
@pytest.mark.parametrize('ours,theirs', [('cube.urdf', 'contactnets_cube.urdf'), ('elbow.urdf', 'contactnets_elbow.urdf'),
                                         ('cube_mesh.urdf', 'contactnets_cube_mesh.urdf'),
                                         ('elbow_mesh.urdf', 'contactnets_elbow_mesh.urdf')])
def test_repo_assets_describe_the_reference_models(ours, theirs):
    mine = O.parse_urdf(os.path.join(ASSET_DIR, ours))
    ref = O.parse_urdf(os.path.join(REFERENCE_DIR, 'assets', theirs))
    assert mine['n_joints'] == ref['n_joints']
    for a, b in zip(mine['bodies'], ref['bodies']):
        for key in ('mass', 'com', 'inertia_cm', 'parent', 'joint_origin', 'joint_axis'):
            assert a[key] == b[key]
        for ga, gb in zip(a['geoms'], b['geoms']):
            assert ga['kind'] == gb['kind'] and ga['mu'] == gb['mu'] and ga['origin'] == gb['origin']
            if ga['kind'] == 'box':
                assert ga['half'] == gb['half']
            else:
                va, vb = torch.tensor(ga['vertices']), torch.tensor(gb['vertices'])
                assert torch.equal(va.max(0).values - va.min(0).values, vb.max(0).values - vb.min(0).values)
