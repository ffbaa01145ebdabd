"""Actuation inputs u: the reference's ``lagrangian_forces(q, v, u, inertia)`` carries ``B u``
(multibody_terms.py:142-146, 235-236) and ``contactnets_loss(x, u, x_plus)`` / ``forward_dynamics(q, v, u)`` hand it through
(multibody_learnable_system.py:104, 199-203).  None of the reference's URDFs has an actuator; ``assets/elbow_actuated.urdf`` is
its elbow with a ``<transmission>`` on the hinge, and ``elbow_actuated_literal.npz`` was recorded by running the reference's own
code on it with seeded torques (oracle/gen_golden.py record_actuated_elbow).  CPU: the oracle and the host build of the per-item
math; GPU (``-m gpu``): the general build through the C ABI."""
import os

import numpy as np
import pytest
import torch

import hostsim
from conftest import ASSET_DIR
from dair_pll_amd import _capi
from dair_pll_amd._capi import make_desc
from dair_pll_amd.urdf import parse_urdf
from oracle import dpll_oracle as O
from test_general_models import fixture_params, reference_gradient

P = 'multibody_terms.'
URDF = os.path.join(ASSET_DIR, 'elbow_actuated.urdf')


def oracle_of(g) -> O.OracleSystem:
    system = O.OracleSystem(URDF, float(g['dt']))
    system.theta = torch.tensor(g['param/' + P + 'lagrangian_terms.inertial_parameters'])
    system.friction = torch.tensor(g['param/' + P + 'contact_terms.friction_params'])
    for index, params in enumerate(system.geom_params):
        for key in list((params or {}).keys()):
            params[key] = torch.tensor(g['param/' + P + f'contact_terms.geometries.{index}.{key}'])
    return system


def test_urdf_transmission_becomes_an_actuator():
    spec = parse_urdf(URDF)
    assert spec.actuators == [0] and spec.n_u == 1 and not spec.is_fast()  # (B u lives in the general build)
    desc = make_desc(spec, 0.0068)
    assert desc.n_u == 1 and desc.act_joint[0] == 0 and desc.n_geoms == 2
    plain = parse_urdf(os.path.join(ASSET_DIR, 'elbow.urdf'))
    assert plain.actuators == [] and plain.is_fast() and make_desc(plain, 0.0068).n_u == 0
    assert O.parse_urdf(URDF)['actuators'] == [0]


def test_oracle_with_actuation_reproduces_the_reference_run(golden):
    g = golden('elbow_actuated_literal')
    system = oracle_of(g).requires_grad_()
    x, xp, u = torch.tensor(g['x']), torch.tensor(g['x_plus']), torch.tensor(g['u'])
    loss = system.contactnets_loss(x, xp, u=u)
    assert np.abs(loss.detach().numpy() - g['loss']).max() < 1e-12
    loss.mean().backward()
    for name, param in system.named_parameters().items():
        ref = g['grad/' + name]
        assert np.abs(param.grad.numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), name
    with torch.no_grad():
        q, v = system.q_v(x)
        assert np.abs(system.forward_dynamics(q, v, u=u).numpy() - g['dynamics/v_next']).max() < 1e-10
        assert np.abs(system.multibody_terms(*system.q_v(xp), u)[4].numpy() - g['terms/a']).max() < 1e-9
        # the inputs matter: without them the acceleration is off by hundreds of rad / s^2
        assert np.abs(system.multibody_terms(*system.q_v(xp))[4].numpy() - g['terms/a']).max() > 10.0
        # step / simulate run unactuated, as the reference's sim_step does (multibody_learnable_system.py:311)
        assert np.abs(system.step(x).numpy() - g['dynamics/x_next']).max() < 1e-10


def test_host_build_with_actuation(golden):
    g = golden('elbow_actuated_literal')
    spec = parse_urdf(URDF)
    desc = make_desc(spec, float(g['dt']), str(g['inertia_mode']))
    theta, friction, lengths = fixture_params(g, spec)
    ref_grad = reference_gradient(g, spec)
    try:
        hostsim.set_actuation(g['u'])
        for dtype, tol_loss, tol_grad, tol_v in ((np.float64, 1e-10, 1e-9, 1e-9), (np.float32, 1e-5, 2e-3, 1e-4)):
            out = hostsim.loss(desc, theta, friction, lengths, g['x'], g['x_plus'], dtype=dtype)
            assert np.abs(out['loss'] - g['loss']).max() < tol_loss * max(1.0, np.abs(g['loss']).max())
            assert np.abs(out['grad'] - ref_grad).max() < tol_grad * max(1.0, np.abs(ref_grad).max())
            x_next, _ = hostsim.step(desc, theta, friction, lengths, g['x'], dtype=dtype)
            n_q = 8
            assert np.abs(x_next[:, n_q:] - g['dynamics/v_next']).max() < tol_v * max(1.0, np.abs(g['dynamics/v_next']).max())
    finally:
        hostsim.set_actuation(None)
    x_next, _ = hostsim.step(desc, theta, friction, lengths, g['x'])  # without inputs: the reference's unactuated step
    assert np.abs(x_next - g['dynamics/x_next']).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_gpu_actuated_elbow_against_the_reference_run(golden, dtype):
    from dair_pll_amd import MultibodyLearnableSystem
    g = golden('elbow_actuated_literal')
    f64 = dtype == torch.float64
    system = MultibodyLearnableSystem({'elbow': URDF}, float(g['dt']), dtype=dtype, device='cuda:0')
    system.load_state_dict({name: torch.tensor(g['param/' + name]) for name, _ in system.named_parameters()})
    assert system.spec.n_u == 1
    x, xp = torch.tensor(g['x'], dtype=dtype, device='cuda:0'), torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
    u = torch.tensor(g['u'], dtype=dtype, device='cuda:0')
    # loss through the caller's path and its gradients
    loss = system.contactnets_loss(x, u, xp)
    assert np.abs(loss.detach().cpu().double().numpy() - g['loss']).max() < (1e-10 if f64 else 1e-5)
    total = system.contactnets_loss_and_grad(x, xp, u=u)  # (the fused path: same launch, mean and gradients in one pass)
    assert abs(total.item() - float(g['loss_mean'])) < (1e-12 if f64 else 1e-6)
    # gradients.  float32: the items at the |phi| kink (resting contacts: the sign of phi, and with it the gradient, flips with
    # float32 input rounding) are left out and the rest compared with the oracle's autograd on the same items -- as
    # tests/test_hip_parity.py::test_gradients_float32 does for the unactuated elbow
    keep = np.ones(x.shape[0], dtype=bool) if f64 else np.abs(g['terms/phi']).min(-1) >= 1e-6
    assert keep.mean() > 0.5
    if f64:
        expected = {name: g['grad/' + name] for name, _ in system.named_parameters()}
    else:
        oracle = oracle_of(g).requires_grad_()
        oracle.contactnets_loss(torch.tensor(g['x'][keep]), torch.tensor(g['x_plus'][keep]), u=torch.tensor(g['u'][keep])).mean().backward()
        expected = {name: value.grad.numpy() for name, value in oracle.named_parameters().items()}
    kept = torch.tensor(keep, device='cuda:0')
    system.zero_grad()
    system.contactnets_loss(x[kept], u[kept], xp[kept]).mean().backward()  # the caller's path (autograd.Function)
    for name, param in system.named_parameters():
        ref = expected[name]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 2e-3) * max(np.abs(ref).max(), 1.0 if f64 else 1e-6), (name, err)
    system.zero_grad()
    system.contactnets_loss_and_grad(x[kept], xp[kept], u=u[kept])
    for name, param in system.named_parameters():
        ref = expected[name]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 2e-3) * max(np.abs(ref).max(), 1.0 if f64 else 1e-6), (name, err)
    # dynamics and terms with the inputs; step / simulate without (sim_step passes a u of width 0)
    q, v = system.space.q_v(x)
    v_next = system.forward_dynamics(q, v, u).detach().cpu().double().numpy()
    assert np.abs(v_next - g['dynamics/v_next']).max() < (1e-9 if f64 else 1e-4) * max(1.0, np.abs(g['dynamics/v_next']).max())
    a = system.multibody_terms(*system.space.q_v(xp), u)[4].cpu().double().numpy()
    assert np.abs(a - g['terms/a']).max() < (1e-8 if f64 else 2e-3) * max(1.0, np.abs(g['terms/a']).max())
    x_next = system.step(x).detach().cpu().double().numpy()
    assert np.abs(x_next - g['dynamics/x_next']).max() < (1e-10 if f64 else 1e-4)
    # wrong widths are refused; an unactuated model refuses any input
    with pytest.raises(_capi.DpllError, match='actuation'):
        system.contactnets_loss(x, torch.zeros((x.shape[0], 2), dtype=dtype, device='cuda:0'), xp)


@pytest.mark.gpu
def test_gpu_gradient_through_actuated_dynamics(golden):
    """d/d parameters and d/d state of forward_dynamics(q, v, u) (dpll_step_backward with the inputs) against torch autograd
    through the oracle's restatement of the reference's forward_dynamics with the same inputs."""
    from dair_pll_amd import MultibodyLearnableSystem
    g = golden('elbow_actuated_literal')
    rows = np.arange(0, g['x'].shape[0], 4)
    system = MultibodyLearnableSystem({'elbow': URDF}, float(g['dt']), dtype=torch.float64, device='cuda:0')
    system.load_state_dict({name: torch.tensor(g['param/' + name]) for name, _ in system.named_parameters()})
    x = torch.tensor(g['x'][rows], device='cuda:0', requires_grad=True)
    u = torch.tensor(g['u'][rows], device='cuda:0')
    w = torch.tensor(np.random.default_rng(5).standard_normal((len(rows), 7)), device='cuda:0')
    q, v = system.space.q_v(x)
    (system.forward_dynamics(q, v, u) * w).sum().backward()
    oracle = oracle_of(g).requires_grad_()
    xo = torch.tensor(g['x'][rows], requires_grad=True)
    qo, vo = oracle.q_v(xo)
    (oracle.forward_dynamics(qo, vo, u=u.cpu()) * w.cpu()).sum().backward()
    named = oracle.named_parameters()
    for name, param in system.named_parameters():
        ref = named[name].grad.numpy()
        assert np.abs(param.grad.cpu().numpy() - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), name
    # state gradient on the tangent space of the unit quaternions (quirk Q2: along q itself the two differ by design)
    gx, gref = x.grad.cpu().numpy(), xo.grad.numpy()
    quat = g['x'][rows][:, :4]
    project = lambda grad: grad[:, :4] - (grad[:, :4] * quat).sum(-1, keepdims=True) * quat
    assert np.abs(project(gx) - project(gref)).max() < 1e-7 * max(1.0, np.abs(gref).max())
    assert np.abs(gx[:, 4:] - gref[:, 4:]).max() < 1e-7 * max(1.0, np.abs(gref).max())


# ---- actuators beyond one short tree: the forest build (csrc/dpll_forest.hpp / .hip) ----------------------------------------------------
# `chain6_actuated` (motors on three of five hinges, listed out of joint order) and `pendulum_actuated_cube` (a fixed-base model
# with a motor next to a free cube): fixtures recorded through the reference's own code with seeded torques
# (oracle/gen_golden.py record_actuated_forest).
FOREST_ACTUATED = {'chain6_actuated': {'chain6_actuated': 'chain6_actuated.urdf'},
                   'pendulum_actuated_cube': {'pendulum': 'pendulum_actuated.urdf', 'cube': 'cube.urdf'}}


def forest_urdfs(name):
    return {key: os.path.join(ASSET_DIR, value) for key, value in FOREST_ACTUATED[name].items()}


def forest_oracle(g, name) -> O.OracleSystem:
    system = O.OracleSystem(forest_urdfs(name), float(g['dt']))
    system.theta = torch.tensor(g['param/' + P + 'lagrangian_terms.inertial_parameters'])
    system.friction = torch.tensor(g['param/' + P + 'contact_terms.friction_params'])
    for index, params in enumerate(system.geom_params):
        for key in list((params or {}).keys()):
            params[key] = torch.tensor(g['param/' + P + f'contact_terms.geometries.{index}.{key}'])
    return system


def test_forest_actuators_in_the_plants_order():
    from dair_pll_amd.urdf import build_system_spec, check_forest_supported
    chain = build_system_spec({'c': parse_urdf(os.path.join(ASSET_DIR, 'chain6_actuated.urdf'))})
    check_forest_supported(chain)
    desc = _capi.make_forest_desc(chain, 0.0068)
    # transmissions listed h34, h12, h56: joints 2, 0, 4 -> bodies 3, 1, 5
    assert chain.n_u == 3 and desc.n_u == 3 and [desc.act_body[k] for k in range(3)] == [3, 1, 5]
    mixed = build_system_spec({key: parse_urdf(path) for key, path in forest_urdfs('pendulum_actuated_cube').items()})
    desc = _capi.make_forest_desc(mixed, 0.0068)
    assert mixed.n_u == 1 and desc.n_u == 1 and desc.act_body[0] == 1 and desc.v_index[1] == 0  # the arm on the pivot: velocity 0
    assert O.system_spec(forest_urdfs('pendulum_actuated_cube'))['actuators'] == [0]
    # the library validates what a C caller hands it
    import ctypes
    lib, handle = _capi.library(), ctypes.c_void_p()
    bad = _capi.ForestDesc.from_buffer_copy(desc)
    bad.act_body[0] = 0  # the mast: welded to the world, no joint to drive
    assert lib.dpll_forest_model_create(ctypes.byref(bad), ctypes.byref(handle)) != 0 and b'act_body' in lib.dpll_last_error()
    bad = _capi.ForestDesc.from_buffer_copy(_capi.make_forest_desc(chain, 0.0068))
    bad.act_body[1] = bad.act_body[0]
    assert lib.dpll_forest_model_create(ctypes.byref(bad), ctypes.byref(handle)) != 0 and b'two actuators' in lib.dpll_last_error()


@pytest.mark.parametrize('name', list(FOREST_ACTUATED))
def test_forest_oracle_and_host_build_with_actuation(golden, name):
    from hostsim import forest
    from test_forest import reference_gradient as forest_reference_gradient, fixture_params as forest_fixture_params
    from dair_pll_amd.urdf import build_system_spec
    g = golden(name + '_literal')
    u = torch.tensor(g['u'])
    oracle = forest_oracle(g, name).requires_grad_()
    x, xp = torch.tensor(g['x']), torch.tensor(g['x_plus'])
    loss = oracle.contactnets_loss(x, xp, u=u)
    assert np.abs(loss.detach().numpy() - g['loss']).max() < 1e-11 * max(1.0, np.abs(g['loss']).max())
    loss.mean().backward()
    for key, param in oracle.named_parameters().items():
        ref = g['grad/' + key]
        assert np.abs(param.grad.numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), key
    with torch.no_grad():
        a_with = oracle.multibody_terms(*oracle.q_v(xp), u)[4].numpy()
        a_without = oracle.multibody_terms(*oracle.q_v(xp))[4].numpy()
    assert np.abs(a_with - g['terms/a']).max() < 1e-9 * max(1.0, np.abs(g['terms/a']).max())
    assert np.abs(a_without - g['terms/a']).max() > 1.0  # the inputs matter
    # the forest program on the host (team of one lane), with the inputs
    system = build_system_spec({key: parse_urdf(path) for key, path in forest_urdfs(name).items()})
    desc = _capi.make_forest_desc(system, float(g['dt']), str(g['inertia_mode']))
    theta, friction, lengths = forest_fixture_params(g, system)
    ref_grad = forest_reference_gradient(g, system)
    try:
        forest.set_actuation(g['u'])
        for dtype, tol_loss, tol_grad, tol_v in ((np.float64, 1e-10, 1e-9, 1e-9), (np.float32, 1e-4, 2e-3, 1e-4)):
            out = forest.loss(desc, theta, friction, lengths, g['x'], g['x_plus'], dtype=dtype)
            assert np.abs(out['loss'] - g['loss']).max() < tol_loss * max(1.0, np.abs(g['loss']).max())
            assert np.abs(out['grad'] - ref_grad).max() < tol_grad * max(1.0, np.abs(ref_grad).max())
            x_next, _ = forest.step(desc, theta, friction, lengths, g['x'], dtype=dtype)
            assert np.abs(x_next[:, system.n_q:] - g['dynamics/v_next']).max() < tol_v * max(1.0, np.abs(g['dynamics/v_next']).max())
        M, a, phi, J = forest.terms(desc, theta, friction, lengths, g['x_plus'])
        assert np.abs(a - g['terms/a']).max() < 1e-9 * max(1.0, np.abs(g['terms/a']).max())
        # the backward of an actuated step: against torch autograd through the oracle's forward_dynamics with the inputs
        rows = np.linspace(0, g['x'].shape[0] - 1, 4).astype(int)
        w = np.random.default_rng(2).standard_normal((len(rows), system.n_q + system.n_v))
        w[:, :system.n_q] = 0.0  # (a functional of the next velocities: forward_dynamics)
        forest.set_actuation(g['u'][rows])
        grad, _ = forest.step_backward(desc, theta, friction, lengths, g['x'][rows], w, want_state=True)
        fresh = forest_oracle(g, name).requires_grad_()
        q, v = fresh.q_v(torch.tensor(g['x'][rows]))
        (fresh.forward_dynamics(q, v, u=u[rows]) * torch.tensor(w[:, system.n_q:])).sum().backward()
        named = {'grad/' + key: value.grad.numpy() for key, value in fresh.named_parameters().items()}
        ref = forest_reference_gradient(named, system)
        assert np.abs(grad - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max())
    finally:
        forest.set_actuation(None)
    x_next, _ = forest.step(desc, theta, friction, lengths, g['x'])  # without inputs: the reference's unactuated step
    assert np.abs(x_next - g['dynamics/x_next']).max() < 1e-9 * max(1.0, np.abs(g['dynamics/x_next']).max())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('name', list(FOREST_ACTUATED))
def test_gpu_forest_build_with_actuation(golden, name, dtype):
    from dair_pll_amd import MultibodyLearnableSystem
    g = golden(name + '_literal')
    f64 = dtype == torch.float64
    system = MultibodyLearnableSystem(forest_urdfs(name), float(g['dt']), dtype=dtype, device='cuda:0')
    system.load_state_dict({key: torch.tensor(g['param/' + key]) for key, _ in system.named_parameters()})
    assert system.forest and system.spec.n_u == g['u'].shape[1]
    x, xp, u = (torch.tensor(g[key], dtype=dtype, device='cuda:0') for key in ('x', 'x_plus', 'u'))
    scale = max(1.0, np.abs(g['loss']).max())
    loss = system.contactnets_loss(x, u, xp)
    assert np.abs(loss.detach().cpu().double().numpy() - g['loss']).max() < (1e-10 if f64 else 1e-4) * scale
    total = system.contactnets_loss_and_grad(x, xp, u=u)
    assert abs(total.item() - float(g['loss_mean'])) < (1e-11 if f64 else 1e-5) * scale
    for key, param in system.named_parameters():
        ref = g['grad/' + key]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 2e-3) * max(1.0, np.abs(ref).max()), (key, err)
    q, v = system.space.q_v(x)
    v_next = system.forward_dynamics(q, v, u).detach().cpu().double().numpy()
    assert np.abs(v_next - g['dynamics/v_next']).max() < (1e-9 if f64 else 1e-4) * max(1.0, np.abs(g['dynamics/v_next']).max())
    a = system.multibody_terms(*system.space.q_v(xp), u)[4].cpu().double().numpy()
    assert np.abs(a - g['terms/a']).max() < (1e-8 if f64 else 2e-3) * max(1.0, np.abs(g['terms/a']).max())
    x_next = system.step(x).detach().cpu().double().numpy()  # unactuated, as the reference's sim_step
    assert np.abs(x_next - g['dynamics/x_next']).max() < (1e-9 if f64 else 1e-4) * max(1.0, np.abs(g['dynamics/x_next']).max())
    with pytest.raises(_capi.DpllError, match='actuation'):
        system.contactnets_loss(x, torch.zeros((x.shape[0], system.spec.n_u + 1), dtype=dtype, device='cuda:0'), xp)
    if f64:  # gradient through the actuated dynamics (dpll_step_backward with the inputs) against autograd through the oracle
        rows = np.linspace(0, x.shape[0] - 1, 6).astype(int)
        w = torch.tensor(np.random.default_rng(4).standard_normal((len(rows), system.space.n_v)), device='cuda:0')
        system.zero_grad()
        xr = x[rows].clone().requires_grad_(True)
        qr, vr = system.space.q_v(xr)
        (system.forward_dynamics(qr, vr, u[rows]) * w).sum().backward()
        oracle = forest_oracle(g, name).requires_grad_()
        qo, vo = oracle.q_v(torch.tensor(g['x'][rows]))
        (oracle.forward_dynamics(qo, vo, u=torch.tensor(g['u'][rows])) * w.cpu()).sum().backward()
        for key, param in system.named_parameters():
            ref = oracle.named_parameters()[key].grad.numpy()
            assert np.abs(param.grad.cpu().numpy() - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), key
