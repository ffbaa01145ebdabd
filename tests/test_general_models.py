"""Models beyond the cube / elbow topologies: a three-link serial chain, a branching tree, several
geometries on one body, spheres, polygons (learnable vertex sets), body-body contact (clasp: box against box; clasp_ball:
a sphere against a polygon; vee_pair: the two arms of a branching tree) -- the GENERAL build (csrc/dpll_general.hip, GeneralDesc branches of csrc/dpll_core.hpp).

Fixtures `{chain3, vee, ballcube, mace, gripper, crank, slider, polycube, wedge}_literal.npz` were recorded by running the reference's own
MultibodyTerms / contactnets_loss / forward_dynamics / simulate (and its Sphere / Polygon classes) on these URDFs
(oracle/gen_golden.py record_general_cases, record_polygon_cases, record_pair_cases -- the last through the reference's
collide_mesh_mesh with fcl's direction supplied by the oracle's exact search).  CPU tests:
the oracle and the host build of the per-item math against them; GPU tests (`-m gpu`): the kernels through the C ABI."""
import os

import numpy as np
import pytest
import torch

import hostsim
from conftest import ASSET_DIR
from dair_pll_amd._capi import make_desc
from dair_pll_amd.urdf import parse_urdf
from oracle import dpll_oracle as O

MODELS = ['chain3', 'vee', 'ballcube', 'mace', 'polycube', 'wedge', 'clasp', 'clasp_ball', 'vee_pair', 'gripper', 'crank',
          'pincer', 'grasp', 'slider']
# model -> (URDF under assets/, what a <mesh> element is read as)
SOURCES = {'polycube': ('cube_mesh.urdf', 'polygon'), 'wedge': ('wedge.urdf', 'polygon'), 'clasp_ball': ('clasp_ball.urdf', 'polygon')}
# assets/welded_arm.urdf: links welded on by `fixed` joints, each with its own row of inertial_parameters (tests/test_welded_links.py
# holds the rest of its tests; the host build's tests here take the kernels' per-body rows)
WITH_WELDS = MODELS + ['welded_arm']
P = 'multibody_terms.'
STRIDE = 24  # numbers per geometry in the general build's lengths block (DPLL_GEOM_BLOCK)
SLOTS = 4    # geometry slots of the general build: three geometries + a body-body pair (DPLL_GEN_SLOTS)


def source(name):
    urdf, representation = SOURCES.get(name, (name + '.urdf', 'deep_support'))
    return os.path.join(ASSET_DIR, urdf), representation


def spec_of(name):
    urdf, representation = source(name)
    return parse_urdf(urdf, representation)


def oracle_from(g, name) -> O.OracleSystem:
    urdf, representation = source(name)
    system = O.OracleSystem(urdf, float(g['dt']), mesh_representation=representation)
    system.theta = torch.tensor(g['param/' + P + 'lagrangian_terms.inertial_parameters'])
    system.friction = torch.tensor(g['param/' + P + 'contact_terms.friction_params'])
    for index, params in enumerate(system.geom_params):
        for key in list((params or {}).keys()):
            params[key] = torch.tensor(g['param/' + P + f'contact_terms.geometries.{index}.{key}'])
    return system


def reference_gradient(g, spec):
    """the reference run's gradients in the kernels' flat layout [theta | friction (1 + 4 slots) | lengths (4 slots, 24)]"""
    n_b = spec.n_joints + 1
    out = np.zeros(10 * n_b + 1 + SLOTS + SLOTS * STRIDE)
    out[:10 * n_b] = g['grad/' + P + 'lagrangian_terms.inertial_parameters'].ravel()
    friction = g['grad/' + P + 'contact_terms.friction_params']
    out[10 * n_b:10 * n_b + len(friction)] = friction
    for index, (_, geom) in enumerate(spec.geoms()):
        key = {'box': 'length_params', 'sphere': 'length_param', 'polygon': 'vertices'}[geom.kind]
        value = g['grad/' + P + f'contact_terms.geometries.{index + 1}.{key}'].ravel()
        out[10 * n_b + 1 + SLOTS + STRIDE * index:10 * n_b + 1 + SLOTS + STRIDE * index + len(value)] = value
    return out


def fixture_params(g, spec):
    """(theta, friction, lengths) of the fixture in the kernels' layout (hostsim.general_params with the recorded values)"""
    theta, friction, lengths = hostsim.general_params(spec)
    theta = g['param/' + P + 'lagrangian_terms.inertial_parameters'].copy()
    recorded = g['param/' + P + 'contact_terms.friction_params']
    friction[:len(recorded)] = recorded
    for index, (_, geom) in enumerate(spec.geoms()):
        key = {'box': 'length_params', 'sphere': 'length_param', 'polygon': 'vertices'}[geom.kind]
        value = g['param/' + P + f'contact_terms.geometries.{index + 1}.{key}'].ravel()
        lengths[index, :len(value)] = value
    return theta, friction, lengths


def canonical(spec, phi, J=None, D=None):
    """contacts of one geometry sorted by signed distance, then -- corners of a box lying flat are equally far -- by their
    normal Jacobian row (the reference's top-k order is unspecified, quirk Q3)"""
    k = phi.shape[-1]
    order = np.zeros(phi.shape, dtype=np.int64)
    start = 0
    for _, geom in spec.geoms():
        n = 1 if geom.kind == 'sphere' else 4
        for item in range(phi.shape[0]):
            keys = [np.round(phi[item, start:start + n], 9)]
            if J is not None:
                keys = [np.round(J[item, start + c, :], 7) for c in range(0)] + \
                       [np.round(J[item, start:start + n, col], 7) for col in range(J.shape[-1] - 1, -1, -1)] + keys
            order[item, start:start + n] = start + np.lexsort(keys)
        start += n
    for _ in spec.pairs:  # one contact per body-body candidate, behind the geometries' contacts
        order[:, start] = start
        start += 1
    rows = np.arange(phi.shape[0])[:, None]
    idx = np.concatenate((order, k + 2 * np.repeat(order, 2, -1) + np.tile([0, 1], k)), -1)
    out = [phi[rows, order]]
    if J is not None:
        out.append(J[rows, idx])
    if D is not None:
        out.append(D[rows[:, :, None], idx[:, :, None], idx[:, None, :]])
    return out


def align_pair_frames(spec, J, D, J_ref):
    """The tangent axes of a body-body contact frame are one choice among the rotations about the normal
    (rotation_matrix_from_one_vector keys on the smallest component of the direction, tensor_utils.py:337-341: a direction
    that is a box's face normal has two components that are zero up to rounding, so the choice is decided by noise).
    Rotates (proper rotation: handedness still has to agree) each pair's tangent rows of J and rows / columns of D onto
    the reference's."""
    k = spec.n_contacts
    J, D = J.copy(), D.copy()
    for p in range(len(spec.pairs)):
        c = k - len(spec.pairs) + p
        tx, ty = k + 2 * c, k + 2 * c + 1
        for item in range(J.shape[0]):
            m = J_ref[item, [tx, ty]] @ J[item, [tx, ty]].T
            angle = np.arctan2(m[1, 0] - m[0, 1], m[0, 0] + m[1, 1])
            rot = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
            J[item, [tx, ty]] = rot @ J[item, [tx, ty]]
            D[item, [tx, ty], :] = rot @ D[item, [tx, ty], :]
            D[item, :, [tx, ty]] = rot @ D[item, :, [tx, ty]]
    return J, D


@pytest.mark.parametrize('name', WITH_WELDS)
def test_oracle_reproduces_the_reference_run(golden, name):
    g = golden(name + '_literal')
    system = oracle_from(g, name).requires_grad_()
    spec = spec_of(name)
    x, xp = torch.tensor(g['x']), torch.tensor(g['x_plus'])
    loss = system.contactnets_loss(x, xp)
    assert (loss.detach() - torch.tensor(g['loss'])).abs().max() < 1e-12
    loss.mean().backward()
    for key, value in system.named_parameters().items():
        ref = g['grad/' + key]
        assert np.abs(value.grad.numpy() - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), key
    with torch.no_grad():
        q, v = system.q_v(xp)
        D, M, J, phi, a = system.multibody_terms(q, v)
        assert (M - torch.tensor(g['terms/M'])).abs().max() < 1e-12 and (a - torch.tensor(g['terms/a'])).abs().max() < 1e-9
        mine = canonical(spec, phi.numpy(), J.numpy(), D.numpy())
        ref = canonical(spec, g['terms/phi'], g['terms/J'], g['terms/D'])
        for m, r in zip(mine, ref):
            assert np.abs(m - r).max() < 1e-9
        ref_next = torch.tensor(g['dynamics/x_next'])
        assert (system.step(x) - ref_next).abs().max() < 1e-9 * max(1.0, ref_next.abs().max().item())


@pytest.mark.parametrize('name', MODELS)
def test_host_build_of_the_kernel_math(golden, name):
    """csrc/dpll_core.hpp compiled for the host with GeneralDesc (tree + geometry table branches): loss, every
    gradient, next state against the reference run; float32 with the double-accumulated residual within 1e-4."""
    g = golden(name + '_literal')
    spec = spec_of(name)
    desc = make_desc(spec, float(g['dt']))
    theta, friction, lengths = fixture_params(g, spec)
    out = hostsim.loss(desc, theta, friction, lengths, g['x'], g['x_plus'])
    assert np.abs(out['loss'] - g['loss']).max() < 1e-12
    ref = reference_gradient(g, spec)
    assert np.abs(out['grad'] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    x_next, iters = hostsim.step(desc, theta, friction, lengths, g['x'])
    scale = max(1.0, np.abs(g['dynamics/x_next']).max())  # (joint rates reach 10 rad/s where a body-body pair closes)
    assert np.abs(x_next - g['dynamics/x_next']).max() < 1e-10 * scale and iters.max() < 100
    out32 = hostsim.loss(desc, theta, friction, lengths, g['x'], g['x_plus'], dtype=np.float32)
    assert np.abs(out32['loss'] - g['loss']).max() < 2e-5
    x_next32, _ = hostsim.step(desc, theta, friction, lengths, g['x'], dtype=np.float32)
    assert np.abs(x_next32 - g['dynamics/x_next']).max() < 1e-4 * scale


@pytest.mark.parametrize('name', MODELS)
def test_host_build_step_backward_against_oracle_autograd(golden, name):
    """the backward of one step (step_item_backward + step_state_adjoint of csrc/dpll_core.hpp, host build): parameter
    gradient and state adjoint of a seeded linear functional of the next state against torch autograd through the
    oracle's step (differentiable cone solve); the state adjoint on the unit-quaternion tangent space (Q2)"""
    g = golden(name + '_literal')
    spec = spec_of(name)
    desc = make_desc(spec, float(g['dt']))
    theta, friction, lengths = fixture_params(g, spec)
    rows = np.linspace(0, g['x'].shape[0] - 1, 12).astype(int)
    x_np = g['x'][rows]
    w = torch.rand(x_np.shape, generator=torch.Generator().manual_seed(11), dtype=torch.float64) - 0.5
    oracle = oracle_from(g, name).requires_grad_()
    x_ref = torch.tensor(x_np).requires_grad_(True)
    (oracle.step(x_ref) * w).sum().backward()
    grad, xbar = hostsim.step_backward(desc, theta, friction, lengths, x_np, w.numpy(), want_state=True)
    named = {key: value.grad.numpy() for key, value in oracle.named_parameters().items()}
    ref = reference_gradient({'grad/' + key: value for key, value in named.items()}, spec)
    assert np.abs(grad - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-3)
    diff = xbar - x_ref.grad.numpy()
    q = x_np[:, :4] / np.linalg.norm(x_np[:, :4], axis=-1, keepdims=True)
    diff[:, :4] -= (diff[:, :4] * q).sum(-1, keepdims=True) * q
    assert np.abs(diff).max() <= 1e-7 * np.abs(x_ref.grad.numpy()).max()


# ---- GPU ------------------------------------------------------------------------------------------------------------
def gpu_system(g, name, dtype):
    from dair_pll_amd import MultibodyLearnableSystem
    urdf, representation = source(name)
    system = MultibodyLearnableSystem({name: urdf}, float(g['dt']), dtype=dtype, device='cuda:0', mesh_representation=representation)
    system.load_state_dict({key: torch.tensor(g['param/' + key]) for key, _ in system.named_parameters()})
    return system


@pytest.mark.gpu
@pytest.mark.parametrize('name', WITH_WELDS)
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_gpu_loss_gradients_dynamics(golden, name, dtype):
    g = golden(name + '_literal')
    system = gpu_system(g, name, dtype)
    assert not system.spec.is_fast()
    x = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
    u = torch.zeros(x.shape[:-1] + (0,), device='cuda:0')
    f64 = dtype == torch.float64
    loss = system.contactnets_loss(x, u, xp)
    # (float32: north_star's budget is 1e-4; the loss evaluated without its big cancellation -- round 4 -- is within 1e-5 on every model)
    assert np.abs(loss.detach().cpu().double().numpy() - g['loss']).max() < (1e-10 if f64 else 1e-5)
    loss.mean().backward()
    for key, param in system.named_parameters():
        ref = g['grad/' + key]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 5e-3) * max(np.abs(ref).max(), 1.0 if f64 else 1e-3), (key, err, np.abs(ref).max())
    system.zero_grad()
    total = system.contactnets_loss_and_grad(x, xp)
    assert abs(total.item() - float(g['loss_mean'])) < (1e-12 if f64 else 1e-6)
    if f64:
        for key, param in system.named_parameters():
            ref = g['grad/' + key]
            assert np.abs(param.grad.cpu().numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), key
    # forces: one row per real contact, feasible
    _, force, iters = system.contact_forces(x, xp)
    k = system.spec.n_contacts
    assert force.shape == (x.shape[0], 3 * k) and iters.max().item() < 100
    f = force.cpu().double().numpy()
    # (the float kernels project in float: feasible to a few ulp of the force itself)
    assert (np.linalg.norm(f[:, k:].reshape(-1, k, 2), axis=-1) <= f[:, :k] * (1.0 + (0.0 if f64 else 1e-5)) + 1e-6).all()
    # dynamics
    tol = (1e-10 if f64 else 1e-4) * max(1.0, np.abs(g['dynamics/x_next']).max())  # (joint rates reach 10 rad/s where a pair closes)
    q, v = system.space.q_v(x)
    v_next = system.forward_dynamics(q, v, u)
    assert np.abs(v_next.detach().cpu().double().numpy() - g['dynamics/v_next']).max() < tol
    assert np.abs(system.step(x).detach().cpu().double().numpy() - g['dynamics/x_next']).max() < tol
    rows, steps = g['simulate/rows'], int(g['simulate/steps'])
    traj, _ = system.simulate(x[rows].unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), steps)
    assert np.abs(traj.detach().cpu().double().numpy() - g['simulate/traj']).max() < (1e-9 if f64 else 5e-4) * max(1.0, np.abs(g['simulate/traj']).max())


@pytest.mark.gpu
@pytest.mark.parametrize('name', WITH_WELDS)
def test_gpu_terms(golden, name):
    g = golden(name + '_literal')
    system = gpu_system(g, name, torch.float64)
    xp = torch.tensor(g['x_plus'], device='cuda:0')
    q, v = system.space.q_v(xp)
    D, M, J, phi, a = system.multibody_terms(q, v, torch.zeros(xp.shape[:-1] + (0,), device='cuda:0'))
    k = system.spec.n_contacts
    assert phi.shape == (xp.shape[0], k) and J.shape == (xp.shape[0], 3 * k, system.space.n_v) and D.shape[-2:] == (3 * k, 3 * k)
    assert np.abs(M.cpu().numpy() - g['terms/M']).max() < 1e-12 and np.abs(a.cpu().numpy() - g['terms/a']).max() < 1e-9
    J_np, D_np = align_pair_frames(system.spec, J.cpu().numpy(), D.cpu().numpy(), g['terms/J'])
    mine = canonical(system.spec, phi.cpu().numpy(), J_np, D_np)
    ref = canonical(system.spec, g['terms/phi'], g['terms/J'], g['terms/D'])
    for m, r in zip(mine, ref):  # (the Delassus entries of the clasp's light arm reach 1e4)
        assert np.abs(m - r).max() < 1e-9 * max(1.0, np.abs(r).max())


@pytest.mark.gpu
@pytest.mark.parametrize('name', MODELS)
def test_gpu_step_backward_against_oracle_autograd(golden, name):
    """parameter gradient and state adjoint of a 2-step rollout: dpll_step_backward of the general build against torch
    autograd through the oracle (differentiable cone solve); state gradients on the unit-quaternion tangent space (Q2)"""
    g = golden(name + '_literal')
    system = gpu_system(g, name, torch.float64)
    rows = np.linspace(0, g['x'].shape[0] - 1, 24).astype(int)
    x_np = g['x'][rows]
    gen = torch.Generator().manual_seed(5)
    w = torch.rand((len(rows), 2, x_np.shape[1]), generator=gen, dtype=torch.float64) - 0.5
    oracle = oracle_from(g, name).requires_grad_()
    x_ref = torch.tensor(x_np).requires_grad_(True)
    traj_ref = oracle.simulate(x_ref, 2)
    (traj_ref[:, 1:] * w).sum().backward()
    x = torch.tensor(x_np, device='cuda:0').requires_grad_(True)
    traj, _ = system.simulate(x.unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), 2)
    assert (traj.detach().cpu() - traj_ref.detach()).abs().max() < 1e-9
    (traj[:, 1:] * w.cuda()).sum().backward()
    ref_named = oracle.named_parameters()
    for key, param in system.named_parameters():
        ref = ref_named[key].grad.numpy()
        err = np.abs(param.grad.cpu().numpy() - ref).max()
        assert err <= 1e-7 * max(np.abs(ref).max(), 1e-3), (key, err, np.abs(ref).max())
    diff = (x.grad.cpu() - x_ref.grad).numpy()
    q = x_np[:, :4] / np.linalg.norm(x_np[:, :4], axis=-1, keepdims=True)
    diff[:, :4] -= (diff[:, :4] * q).sum(-1, keepdims=True) * q
    assert np.abs(diff).max() <= 1e-7 * x_ref.grad.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('name', MODELS)
def test_gpu_float32_terms_and_step_backward(golden, name):
    """the float32 instantiations of the kernels the two tests above run in float64 (every instantiation is compiled on its
    own: DESIGN.md 4a "compiler fragility"): terms against the fixture, the gradient of a 2-step rollout against the
    float64 kernels, at float32 accuracy"""
    g = golden(name + '_literal')
    s32, s64 = gpu_system(g, name, torch.float32), gpu_system(g, name, torch.float64)
    xp = torch.tensor(g['x_plus'], device='cuda:0')
    k = s32.spec.n_contacts
    q, v = s32.space.q_v(xp.float())
    D, M, J, phi, a = s32.multibody_terms(q, v, torch.zeros(xp.shape[:-1] + (0,), device='cuda:0'))
    assert np.abs(M.cpu().double().numpy() - g['terms/M']).max() < 1e-5 * max(1.0, np.abs(g['terms/M']).max())
    assert np.abs(a.cpu().double().numpy() - g['terms/a']).max() < 2e-3 * max(1.0, np.abs(g['terms/a']).max())
    J_np, D_np = align_pair_frames(s32.spec, J.cpu().double().numpy(), D.cpu().double().numpy(), g['terms/J'])
    mine = canonical(s32.spec, np.round(phi.cpu().double().numpy(), 5), J_np, D_np)
    ref = canonical(s32.spec, np.round(g['terms/phi'], 5), g['terms/J'], g['terms/D'])
    assert np.abs(canonical(s32.spec, phi.cpu().double().numpy())[0] - canonical(s32.spec, g['terms/phi'])[0]).max() < 1e-5  # (sorted: Q3)
    assert np.abs(np.sort(np.abs(mine[1]).sum(-1), axis=-1) - np.sort(np.abs(ref[1]).sum(-1), axis=-1)).max() < 1e-3 * max(1.0, np.abs(ref[1]).max())
    rows = np.linspace(0, g['x'].shape[0] - 1, 24).astype(int)
    w = torch.rand((len(rows), 2, g['x'].shape[1]), generator=torch.Generator().manual_seed(5), dtype=torch.float64) - 0.5
    grads = []
    for system, dtype in ((s32, torch.float32), (s64, torch.float64)):
        x = torch.tensor(g['x'][rows], dtype=dtype, device='cuda:0').requires_grad_(True)
        traj, _ = system.simulate(x.unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), 2)
        (traj[:, 1:] * w.to(dtype).cuda()).sum().backward()
        grads.append((torch.cat([p.grad.reshape(-1).double() for _, p in system.named_parameters()]).cpu(), x.grad.double().cpu()))
    (p32, x32), (p64, x64) = grads
    assert torch.isfinite(p32).all() and torch.isfinite(x32).all()
    # (float32 rollouts through contact: the two precisions agree to a few per cent of the largest entry)
    assert (p32 - p64).abs().max() <= 5e-2 * p64.abs().max() and (x32 - x64).abs().max() <= 5e-2 * x64.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize('name', MODELS)
def test_gpu_matches_the_host_build_on_many_random_states(golden, name):
    """2048 seeded states per model near the ground (contacts engaged, sliding, airborne; joints anywhere, so body-body
    candidates apart and overlapping), far more solver paths than the 96-120 pairs of a fixture: next state, loss and
    every gradient of the kernels against the SAME templates compiled for the host (one lane per item) in float64 --
    what would catch an instantiation that the device compiler got wrong only on a path the fixtures do not take."""
    g = golden(name + '_literal')
    spec = spec_of(name)
    desc = make_desc(spec, float(g['dt']))
    theta, friction, lengths = fixture_params(g, spec)
    system = gpu_system(g, name, torch.float64)
    gen = torch.Generator().manual_seed(23)
    n, n_j = 2048, spec.n_joints
    sliding = torch.tensor([b.joint_kind == 'prismatic' for b in spec.bodies[1:]], dtype=torch.bool)
    quat = torch.randn((n, 4), generator=gen, dtype=torch.float64)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    q = torch.cat((quat, 0.05 * torch.randn((n, 2), generator=gen, dtype=torch.float64),
                   0.03 + 0.08 * torch.rand((n, 1), generator=gen, dtype=torch.float64),
                   torch.where(sliding, 0.03, 2.0) * torch.randn((n, n_j), generator=gen, dtype=torch.float64)), -1)
    v = torch.cat((3 * torch.randn((n, 3), generator=gen, dtype=torch.float64), 0.5 * torch.randn((n, 3), generator=gen, dtype=torch.float64),
                   torch.where(sliding, 0.3, 3.0) * torch.randn((n, n_j), generator=gen, dtype=torch.float64)), -1)
    x = torch.cat((q, v), -1)
    x_next_host, iters = hostsim.step(desc, theta, friction, lengths, x.numpy())
    assert iters.max() < 100
    xd = x.cuda()
    x_next = system.step(xd).detach().cpu().numpy()
    # Deeply overlapping boxes have several directions of equal penetration up to rounding (the device contracts
    # multiply-adds, the host compiler does not): the direction search treats separations within kPairTie (1e-12 m) as a tie
    # and gives it to the candidate with the lowest number -- in a lane's own sequence and in the butterfly over the lanes --
    # so device and host pick the SAME contact (round 3 tolerated 8 of 2048 items on different, equally deep contacts)
    row_scale = np.maximum(1.0, np.abs(x_next_host).max(axis=1))
    same = np.abs(x_next - x_next_host).max(axis=1) < 1e-8 * row_scale
    assert same.all(), (~same).sum()
    # the loss of the transition to a perturbed next state (so that its solve is not the dynamics' own)
    xp = torch.tensor(x_next_host)
    xp[:, -(6 + n_j):] += 0.05 * torch.randn((n, 6 + n_j), generator=gen, dtype=torch.float64)
    host = hostsim.loss(desc, theta, friction, lengths, x.numpy(), xp.numpy())
    loss = system.contactnets_loss(xd, torch.zeros((n, 0), device='cuda:0'), xp.cuda())
    same_loss = np.abs(loss.detach().cpu().numpy() - host['loss']) < 1e-9 * np.maximum(1.0, np.abs(host['loss']))
    assert same_loss.all(), (~same_loss).sum()
    loss.mean().backward()
    ref = host['grad']  # (kernel layout: [theta | friction (1 + 4 slots) | lengths (4 slots, 24)]; the module's parameters in order)
    mine = reference_gradient({'grad/' + key: p.grad.cpu().numpy() for key, p in system.named_parameters()}, spec)
    assert np.abs(mine - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())
    # The float32 kernels on the same states, at north_star's tolerance (1e-4 on next state and loss) for EVERY item whose answer
    # float32 inputs can resolve at all: both precisions get the float32-rounded states (no input rounding in the comparison),
    # and an item is excused only if the float64 kernels themselves move by more than a quarter of the tolerance when its
    # inputs are nudged by two float32 ulps -- its active set sits within float32 resolution of an edge (a corner about to
    # touch, a contact between sticking and sliding), the one thing no float32 kernel can decide; at most 3 % of the states (measured: 0 - 1.3 %).
    s32 = gpu_system(g, name, torch.float32)
    tol = 1e-4
    x32, xp32 = xd.float(), xp.cuda().float()
    xr, xpr = x32.double(), xp32.double()
    u0 = torch.zeros((n, 0), device='cuda:0')
    with torch.no_grad():
        next32 = s32.step(x32).double().cpu().numpy()
        loss32 = s32.contactnets_loss(x32, u0, xp32).double().cpu().numpy()
        next64, loss64 = system.step(xr).cpu().numpy(), system.contactnets_loss(xr, u0, xpr).cpu().numpy()
        scale_next = np.maximum(1.0, np.abs(next64).max(axis=1))
        scale_loss = np.maximum(1.0, np.abs(loss64))
        sens_next, sens_loss = np.zeros(n), np.zeros(n)
        nudge = torch.Generator().manual_seed(7)
        for _ in range(3):
            sx = (1 + 2.4e-7 * (2.0 * torch.randint(0, 2, xr.shape, generator=nudge) - 1.0)).cuda()
            sp = (1 + 2.4e-7 * (2.0 * torch.randint(0, 2, xr.shape, generator=nudge) - 1.0)).cuda()
            sens_next = np.maximum(sens_next, np.abs(system.step(xr * sx).cpu().numpy() - next64).max(axis=1) / scale_next)
            sens_loss = np.maximum(sens_loss, np.abs(system.contactnets_loss(xr * sx, u0, xpr * sp).cpu().numpy() - loss64) / scale_loss)
    assert np.isfinite(next32).all() and np.isfinite(loss32).all()
    err_next = np.abs(next32 - next64).max(axis=1) / scale_next
    err_loss = np.abs(loss32 - loss64) / scale_loss
    edge_next, edge_loss = sens_next > tol / 4, sens_loss > tol / 4
    assert ((err_next <= tol) | edge_next).all(), (int(((err_next > tol) & ~edge_next).sum()), err_next[~edge_next].max())
    assert ((err_loss <= tol) | edge_loss).all(), (int(((err_loss > tol) & ~edge_loss).sum()), err_loss[~edge_loss].max())
    assert edge_next.mean() <= 0.03 and edge_loss.mean() <= 0.03, (edge_next.mean(), edge_loss.mean())


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['chain3', 'grasp', 'crank'])
def test_gpu_batch_sizes_of_the_general_build(golden, name):
    """Size-independent properties (the reference's batch is any leading shape): an item's loss, next state and rollout
    do not depend on the batch it sits in -- ragged sizes that leave lanes of the last wave idle (1, 3, 5, 4097 items at
    four items per wave) and 40,000 items, where the capped grid loops over the items -- and the
    gradient of a mean is the mean of the gradients (fixed-order partial rows: rows of one wave, folded rows, looped grid)."""
    g = golden(name + '_literal')
    system = gpu_system(g, name, torch.float64)
    n = g['x'].shape[0]
    x = torch.tensor(g['x'], device='cuda:0')
    xp = torch.tensor(g['x_plus'], device='cuda:0')
    u = lambda t: torch.zeros(t.shape[:-1] + (0,), device='cuda:0')
    with torch.no_grad():
        base_loss, base_next = system.contactnets_loss(x, u(x), xp), system.step(x)
    assert (base_loss.cpu().numpy() - g['loss']).__abs__().max() < 1e-10
    gen = torch.Generator().manual_seed(4)
    for size in (1, 3, 5, 4097, 40000):  # (an empty batch is refused loudly: tests/test_hip_edges.py)
        pick = torch.randint(0, n, (size,), generator=gen).cuda()
        with torch.no_grad():
            loss, nxt = system.contactnets_loss(x[pick], u(x[pick]), xp[pick]), system.step(x[pick])
        assert loss.shape == (size,) and nxt.shape == (size, x.shape[1])
        assert torch.equal(loss, base_loss[pick]) and torch.equal(nxt, base_next[pick])  # bit for bit, wherever the item sits
    # gradients: the batch repeated k times under .mean() gives the same gradient; one item alone gives its own
    def grads(xb, xpb):
        system.zero_grad()
        system.contactnets_loss(xb, u(xb), xpb).mean().backward()
        return torch.cat([p.grad.reshape(-1) for _, p in system.named_parameters()]).clone()
    whole = grads(x, xp)
    tiled = grads(x.repeat(334, 1), xp.repeat(334, 1))  # (> 8192 waves of four items: the looped grid)
    assert (tiled - whole).abs().max() <= 1e-12 * max(1.0, whole.abs().max().item())
    per_item = torch.stack([grads(x[i:i + 1], xp[i:i + 1]) for i in range(0, n, max(1, n // 6))])
    subset = grads(x[::max(1, n // 6)], xp[::max(1, n // 6)])
    assert (per_item.mean(0) - subset).abs().max() <= 1e-12 * max(1.0, subset.abs().max().item())
    # rollouts: trajectories of a ragged batch equal those of the items alone
    with torch.no_grad():
        traj, _ = system.simulate(x[:7].unsqueeze(-2), torch.zeros((7, 1), device='cuda:0'), 3)
        one, _ = system.simulate(x[6:7].unsqueeze(-2), torch.zeros((1, 1), device='cuda:0'), 3)
    assert torch.equal(traj[6], one[0])


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['clasp', 'clasp_ball', 'vee_pair', 'pincer', 'grasp'])
def test_gpu_pair_models_on_random_states(golden, name):
    """Body-body contact away from the rollouts of the fixtures: 192 seeded states with the joints anywhere (the pair far
    apart, touching, and overlapping by centimetres), parameters as recorded -- loss, every gradient and the next state of
    the kernels against the oracle (whose direction comes from the convex hull of the Minkowski difference)."""
    g = golden(name + '_literal')
    system = gpu_system(g, name, torch.float64)
    oracle = oracle_from(g, name).requires_grad_()
    gen = torch.Generator().manual_seed(11)
    n, n_j = 192, system.spec.n_joints
    quat = torch.randn((n, 4), generator=gen, dtype=torch.float64)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    q = torch.cat((quat, 0.05 * torch.randn((n, 2), generator=gen, dtype=torch.float64),
                   0.06 + 0.05 * torch.rand((n, 1), generator=gen, dtype=torch.float64),
                   2.5 * torch.randn((n, n_j), generator=gen, dtype=torch.float64)), -1)
    v = torch.cat((3 * torch.randn((n, 3), generator=gen, dtype=torch.float64), 0.5 * torch.randn((n, 3), generator=gen, dtype=torch.float64),
                   3 * torch.randn((n, n_j), generator=gen, dtype=torch.float64)), -1)
    x = torch.cat((q, v), -1)
    with torch.no_grad():
        x_next_ref = oracle.step(x)
        phi_pair = oracle.contact_terms(oracle.q_v(x_next_ref)[0])[0][:, -len(system.spec.pairs):]
    for p in range(phi_pair.shape[1]):  # every candidate occurs overlapping and apart
        assert (phi_pair[:, p] < -0.002).sum() >= 3 and (phi_pair[:, p] > 0.01).sum() >= 50
    assert (phi_pair.min(dim=-1).values < -0.002).sum() >= 5
    loss_ref = oracle.contactnets_loss(x, x_next_ref)
    loss_ref.mean().backward()
    xd, xpd = x.cuda(), x_next_ref.cuda()
    x_next = system.step(xd)
    assert (x_next.detach().cpu() - x_next_ref).abs().max() < 1e-9 * max(1.0, x_next_ref.abs().max().item())
    loss = system.contactnets_loss(xd, torch.zeros((n, 0), device='cuda:0'), xpd)
    assert (loss.detach().cpu() - loss_ref.detach()).abs().max() < 1e-10 * max(1.0, loss_ref.abs().max().item())
    loss.mean().backward()
    ref_named = oracle.named_parameters()
    for key, param in system.named_parameters():
        ref = ref_named[key].grad.numpy()
        err = np.abs(param.grad.cpu().numpy() - ref).max()
        assert err <= 1e-8 * max(np.abs(ref).max(), 1e-3), (key, err, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['ballcube', 'chain3', 'gripper', 'clasp', 'grasp', 'slider'])
def test_gpu_workspace_is_exactly_what_dpll_workspace_bytes_says(golden, name):
    """dpll_workspace_bytes of the general build left out a block of the chain matrix, so the last
    folded row landed past the end.  The workspace here is EXACTLY that many bytes inside a guard band of canaries; loss +
    gradients and the step backward run over ragged batch sizes (0/1/2/3-joint models, with and without a pair)."""
    import ctypes
    from dair_pll_amd import _capi
    g = golden(name + '_literal')
    lib = _capi.library()
    for dtype in (torch.float64, torch.float32):
        system = gpu_system(g, name, dtype)
        code = _capi.F64 if dtype == torch.float64 else _capi.F32
        flat = system._packed()
        params = system._params_struct(flat)
        rng = np.random.default_rng(3)
        for batch in (1, 3, 64, 257, 512, 1000, 1024, 8192 + 5):
            pick = rng.integers(0, g['x'].shape[0], size=batch)
            x = torch.tensor(g['x'][pick], dtype=dtype, device='cuda:0')
            xp = torch.tensor(g['x_plus'][pick], dtype=dtype, device='cuda:0')
            need = lib.dpll_workspace_bytes(system._model(), batch)
            guard = 4096
            arena = torch.full((need + 2 * guard,), 0xA5, dtype=torch.uint8, device='cuda:0')
            ws = arena[guard:guard + need]
            assert ws.data_ptr() % 16 == 0
            grad = torch.zeros(flat.numel(), dtype=dtype, device='cuda:0')
            total = torch.zeros(1, dtype=dtype, device='cuda:0')
            _capi.check(lib.dpll_contactnets_loss(system._model(), code, ctypes.byref(params), x.data_ptr(), x.stride(0),
                                                  xp.data_ptr(), xp.stride(0), batch, None, 1.0 / batch, None, grad.data_ptr(),
                                                  total.data_ptr(), None, None, ws.data_ptr(), need, system._stream()))
            gx = torch.ones_like(x)
            grad2 = torch.zeros_like(grad)
            _capi.check(lib.dpll_step_backward(system._model(), code, ctypes.byref(params), x.data_ptr(), x.stride(0),
                                               gx.data_ptr(), gx.stride(0), batch, grad2.data_ptr(), None, 0, ws.data_ptr(), need,
                                               system._stream()))
            torch.cuda.synchronize()
            assert (arena[:guard] == 0xA5).all() and (arena[guard + need:] == 0xA5).all(), (name, dtype, batch)
            assert torch.isfinite(grad).all() and torch.isfinite(grad2).all() and torch.isfinite(total).all()
            # one byte less is refused, not overrun
            rc = lib.dpll_contactnets_loss(system._model(), code, ctypes.byref(params), x.data_ptr(), x.stride(0), xp.data_ptr(),
                                           xp.stride(0), batch, None, 1.0 / batch, None, grad.data_ptr(), total.data_ptr(), None, None,
                                           ws.data_ptr(), need - 1, system._stream())
            assert rc != 0


@pytest.mark.gpu
def test_poisoned_locals_build():
    """The regression gate of DESIGN.md 4a / csrc/Makefile (GENERAL_EXTRA): a build of the general translation units in which
    EVERY automatic variable is pre-filled with clang's poison pattern (``make -C dair_pll_amd/csrc poison-check``, also built by
    ``__graft_entry__.build()``) must pass the step-backward tests of the 3-joint models -- the four tests that failed for the
    round-4 flags, where the backend's promote-alloca pass decided what the kernel wrote for the length gradients.  Runs them
    in a child process against that library; skips LOUDLY when the variant has not been built."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(repo, 'tools', 'diag', 'variants', 'libdpll_hip_poison.so')
    if not os.path.exists(lib):
        pytest.skip('POISON CHECK NOT RUN: tools/diag/variants/libdpll_hip_poison.so is missing -- build it with '
                    '`make -C dair_pll_amd/csrc poison-check` (or __graft_entry__.build())')
    env = dict(os.environ, DPLL_HIP_LIBRARY=lib)
    run = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-m', 'gpu', '-q', '-x', '-p', 'no:cacheprovider',
                          '-k', 'step_backward and (gripper or grasp)'], env=env, cwd=repo, capture_output=True, text=True, timeout=900)
    tail = run.stdout.strip().splitlines()[-1] if run.stdout.strip() else run.stderr[-400:]
    assert run.returncode == 0 and ' passed' in tail and 'failed' not in tail, tail
