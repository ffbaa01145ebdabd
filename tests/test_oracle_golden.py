"""The oracle restatement vs the fixtures recorded from the reference's own Python
(oracle/gen_golden.py).  CPU only."""
import itertools
import os

import numpy as np
import pytest
import torch

from conftest import ASSET_DIR
from oracle import dpll_oracle as O

torch.set_default_dtype(torch.float64)

CASES = ['cube_box_literal', 'cube_box_physical', 'cube_box_config1', 'elbow_box_literal', 'cube_mesh_literal',
         'elbow_mesh_literal', 'clasp_mesh_literal']
URDF = {'contactnets_cube.urdf': 'cube.urdf', 'contactnets_elbow.urdf': 'elbow.urdf',
        'contactnets_cube_mesh.urdf': 'cube_mesh.urdf', 'contactnets_elbow_mesh.urdf': 'elbow_mesh.urdf',
        # the reference's own body-body case, two DeepSupportConvex shapes through collide_mesh_mesh (geometry.py:585-643):
        # oracle and fixture only so far, the kernels take box / sphere / polygon pairs (tests/test_general_models.py)
        'clasp_mesh.urdf': 'clasp_mesh.urdf'}
PREFIX = 'multibody_terms.contact_terms.geometries.'


def oracle_from_golden(g) -> O.OracleSystem:
    system = O.OracleSystem(os.path.join(ASSET_DIR, URDF[str(g['urdf'])]), float(g['dt']),
                            inertia_mode=str(g['inertia_mode']))
    system.theta = torch.tensor(g['param/multibody_terms.lagrangian_terms.inertial_parameters'])
    system.friction = torch.tensor(g['param/multibody_terms.contact_terms.friction_params'])
    for index, params in enumerate(system.geom_params):
        if params is None:
            continue
        for key in list(params.keys()):
            name = PREFIX + f'{index}.' + (key if key in ('length_params', 'perturbations') else 'network.' + key)
            params[key] = torch.tensor(g['param/' + name])
    return system


def match_contacts(J_ref, J_mine, k):
    """per item and per geometry (groups of 4) the permutation of my contacts that matches the
    reference's unspecified top-k order (quirk Q3)."""
    perms = np.array(list(itertools.permutations(range(4))))
    order = np.tile(np.arange(k), (J_ref.shape[0], 1))  # (a body-body contact behind the groups of four keeps its place)
    for group in range(k // 4):
        rows = np.arange(4) + 4 * group
        ref = J_ref[:, rows, :]
        cost = np.stack([np.abs(ref - J_mine[:, rows[p], :]).sum((-1, -2)) for p in perms], -1)
        order[:, rows] = rows[perms[cost.argmin(-1)]]
    return order


def permute_terms(order, k, J, phi, D):
    idx = np.concatenate((order, k + 2 * np.repeat(order, 2, -1) + np.tile([0, 1], k)), -1)
    rows = np.arange(J.shape[0])[:, None]
    return J[rows, idx], phi[rows, order], D[rows[:, :, None], idx[:, :, None], idx[:, None, :]]


@pytest.mark.parametrize('case', CASES)
def test_terms_match_reference_run(golden, case):
    g = golden(case)
    system = oracle_from_golden(g)
    x_plus = torch.tensor(g['x_plus'])
    q, v = system.q_v(x_plus)
    with torch.no_grad():
        D, M, J, phi, a = [t.numpy() for t in system.multibody_terms(q, v)]
    k = phi.shape[-1]
    order = match_contacts(g['terms/J'][:, :k], J[:, :k], k)
    J, phi, D = permute_terms(order, k, J, phi, D)
    assert np.abs(M - g['terms/M']).max() < 1e-13
    assert np.abs(a - g['terms/a']).max() < 1e-10
    assert np.abs(phi - g['terms/phi']).max() < 1e-13
    assert np.abs(J - g['terms/J']).max() < 1e-13
    assert np.abs(D - g['terms/D']).max() < 1e-9 * max(1.0, np.abs(g['terms/D']).max())


@pytest.mark.parametrize('case', CASES)
def test_loss_and_gradients_match_reference_run(golden, case):
    g = golden(case)
    system = oracle_from_golden(g).requires_grad_()
    loss = system.contactnets_loss(torch.tensor(g['x']), torch.tensor(g['x_plus']))
    assert np.abs(loss.detach().numpy() - g['loss']).max() < 1e-12
    loss.mean().backward()
    assert abs(loss.mean().item() - float(g['loss_mean'])) < 1e-13
    for name, param in system.named_parameters().items():
        ref = g['grad/' + name]
        assert np.abs(param.grad.numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), name


@pytest.mark.parametrize('case', CASES)
def test_dynamics_match_reference_run(golden, case):
    g = golden(case)
    system = oracle_from_golden(g)
    x = torch.tensor(g['x'])
    with torch.no_grad():
        q, v = system.q_v(x)
        assert np.abs(system.forward_dynamics(q, v).numpy() - g['dynamics/v_next']).max() < 1e-10
        assert np.abs(system.step(x).numpy() - g['dynamics/x_next']).max() < 1e-10
        rows = g['simulate/rows']
        traj = system.simulate(x[rows], int(g['simulate/steps']))
        assert np.abs(traj.numpy() - g['simulate/traj']).max() < 1e-9


@pytest.mark.parametrize('case', CASES)
def test_recorded_solver_problems_satisfy_kkt(golden, case):
    """the fixtures carry the exact (J_M, P^T q, eps) the reference handed to its solver."""
    g = golden(case)
    for tag in ('solver_loss', 'solver_dynamics'):
        J, q, eps = torch.tensor(g[tag + '/J']), torch.tensor(g[tag + '/q']), float(g[tag + '/eps'])
        f = O.sap_solve(J, q, eps)
        assert (f - torch.tensor(g[tag + '/f'])).abs().max() < 1e-10
        kkt = O.kkt_residuals(J, q, eps, f)
        size = 1 + f.abs().amax(-1)
        assert (kkt['primal'] / size).max() < 1e-13
        assert (kkt['dual'] / size).max() < 1e-10
        assert (kkt['complementarity'] / size**2).max() < 1e-10


def test_mesh_benchmark_batch_fixture_subset(golden):
    """cube_mesh_4096 (BASELINE configs[3], reference run in chunks of 256): the oracle on every 16th pair reproduces the
    recorded per-item losses, and the fixture's network is the one of cube_mesh_literal."""
    g = golden('cube_mesh_4096')
    pairs = golden(str(g['pairs_from']))
    system = oracle_from_golden(g)
    rows = np.arange(0, g['loss'].shape[0], 16)
    with torch.no_grad():
        loss = system.contactnets_loss(torch.tensor(pairs['x'][rows]), torch.tensor(pairs['x_plus'][rows]))
    assert np.abs(loss.numpy() - g['loss'][rows]).max() < 1e-12
    assert abs(float(g['loss_mean']) - g['loss'].mean()) < 1e-15
    lit = golden('cube_mesh_literal')
    key = 'param/' + PREFIX + '1.network.hidden_weights.0'
    assert np.array_equal(g[key], lit[key])


def test_elbow_benchmark_batch_kept_subset(golden):
    """elbow_box_4096: the reference-run gradients over the items off the |phi| kink (what the float32 kernels are held
    to) are the oracle's autograd on the same subset."""
    g = golden('elbow_box_4096')
    keep = g['keep']
    assert 0.5 < keep.mean() < 1.0 and (g['terms/phi_min'][keep] >= 1e-6).all()
    system = oracle_from_golden(g).requires_grad_()
    loss = system.contactnets_loss(torch.tensor(g['x'][keep]), torch.tensor(g['x_plus'][keep]))
    loss.mean().backward()
    assert abs(loss.mean().item() - float(g['loss_mean_keep'])) < 1e-13
    for name, param in system.named_parameters().items():
        ref = g['grad_keep/' + name]
        assert np.abs(param.grad.numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), name
