"""Mesh geometry (DeepSupportConvex / HomogeneousICNN, BASELINE configs[3]) through the C ABI against the
reference-run fixture `cube_mesh_literal` and the oracle.  Needs the MI355X: `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch

from conftest import ASSET_DIR
from oracle import dpll_oracle as O

pytestmark = pytest.mark.gpu
GEOM = 'multibody_terms.contact_terms.geometries.'
NET = GEOM + '1.'
URDF = {'contactnets_cube_mesh.urdf': 'cube_mesh.urdf', 'contactnets_elbow_mesh.urdf': 'elbow_mesh.urdf'}
CASES = ['cube_mesh_literal', 'elbow_mesh_literal']


def build(g, dtype):
    from dair_pll_amd import MultibodyLearnableSystem
    system = MultibodyLearnableSystem({'model': os.path.join(ASSET_DIR, URDF[str(g['urdf'])])}, float(g['dt']), dtype=dtype,
                                      device='cuda:0')
    system.load_state_dict({name: torch.tensor(g['param/' + name]) for name, _ in system.named_parameters()})
    for index, geometry in enumerate(system.multibody_terms.contact_terms.geometries):
        if index > 0:  # the fixed perturbation directions are a buffer drawn at construction (geometry.py:382-385)
            geometry.perturbations = torch.tensor(g[f'param/{GEOM}{index}.perturbations'], dtype=dtype, device='cuda:0')
    return system


def oracle_for(g):
    n_nets = sum(1 for key in g.files if key.endswith('.perturbations'))
    meshes = {}
    for index in range(1, n_nets + 1):
        meshes[index] = {'perturbations': torch.tensor(g[f'param/{GEOM}{index}.perturbations'])}
        for key in ('hidden_weights.0', 'input_weights.0', 'input_weights.1', 'output_weight'):
            meshes[index][key] = torch.tensor(g[f'param/{GEOM}{index}.network.{key}'])
    system = O.OracleSystem(os.path.join(ASSET_DIR, URDF[str(g['urdf'])]), float(g['dt']), mesh_params=meshes)
    system.theta = torch.tensor(g['param/multibody_terms.lagrangian_terms.inertial_parameters'])
    system.friction = torch.tensor(g['param/multibody_terms.contact_terms.friction_params'])
    return system


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_mesh_loss_gradients_dynamics(golden, dtype, case):
    g = golden(case)
    system = build(g, dtype)
    x = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
    f64 = dtype == torch.float64
    # support points vs the oracle's DeepSupportConvex restatement (order-free: sort rows), one network per body
    oracle = oracle_for(g)
    R_WC = O.geometry_kinematics(oracle.spec, oracle.q_v(torch.tensor(g['x_plus']))[0])[0]
    pts = system.support_points(xp).cpu().double().numpy()
    key = lambda a: np.sort(a.reshape(a.shape[0], -1), axis=-1)
    for index in range(1, R_WC.shape[-3]):
        ref_pts = oracle.support_points(index, -R_WC[:, index, 2, :]).numpy()
        mine = pts[:, 4 * (index - 1):4 * index]
        assert np.abs(key(mine) - key(ref_pts)).max() < (1e-12 if f64 else 2e-6), index
    # loss: autograd.Function path
    loss = system.contactnets_loss(x, torch.zeros((x.shape[0], 0), device='cuda:0'), xp)
    assert np.abs(loss.detach().cpu().double().numpy() - g['loss']).max() < (1e-10 if f64 else 1e-4)
    loss.mean().backward()
    for name, param in system.named_parameters():
        ref = g['grad/' + name]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 2e-3) * max(np.abs(ref).max(), 1e-12 if not f64 else 1.0), (name, err)
    # fused path
    system.zero_grad()
    total = system.contactnets_loss_and_grad(x, xp)
    assert abs(total.item() - float(g['loss_mean'])) < (1e-12 if f64 else 1e-6)
    for name, param in system.named_parameters():
        ref = g['grad/' + name]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 2e-3) * max(np.abs(ref).max(), 1e-12 if not f64 else 1.0), (name, err)
    # dynamics
    x_next = system.step(x).detach()
    assert np.abs(x_next.cpu().double().numpy() - g['dynamics/x_next']).max() < (1e-10 if f64 else 1e-4)
    rows = g['simulate/rows']
    traj, _ = system.simulate(x[rows].unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), int(g['simulate/steps']))
    assert np.abs(traj.detach().cpu().double().numpy() - g['simulate/traj']).max() < (1e-9 if f64 else 5e-4)


@pytest.mark.parametrize('form', ['f64', 'f32', 'f32 8-wave kernels', 'bf16 2 planes', 'bf16 3 planes', 'fp16 2 planes'])
def test_benchmark_batch_mesh_4096(golden, form):
    """BASELINE configs[3] at its stated size: the 4096 benchmark pairs (16,384 support queries) through the reference's own
    DeepSupportConvex / HomogeneousICNN loss (oracle/gen_golden.py: record_mesh_bench_batch) -- per-item loss, batch mean and
    every gradient incl. the 67,328 network weights -- against the float64 kernels, the exact float32 MFMA kernels (pipelined
    and 8-wave), both split-bf16 forms and the two-fp16-plane form.  (Round 4 held this size only against this repository's own float64 kernels.)"""
    g = golden('cube_mesh_4096')
    pairs = golden(str(g['pairs_from']))
    f64 = form == 'f64'
    dtype = torch.float64 if f64 else torch.float32
    system = build(g, dtype)
    if not f64:
        system.set_solver(mesh_gemm={'f32': 0, 'f32 8-wave kernels': 1, 'bf16 2 planes': 2, 'bf16 3 planes': 3, 'fp16 2 planes': 4}[form])
    x, xp = torch.tensor(pairs['x'], dtype=dtype, device='cuda:0'), torch.tensor(pairs['x_plus'], dtype=dtype, device='cuda:0')
    loss = system.contact_forces(x, xp)[0].cpu().double().numpy()
    err = np.abs(loss - g['loss'])
    if f64:
        assert err.max() < 1e-10, err.max()
    else:
        # float32: a hidden unit within rounding of zero can take the other LeakyReLU branch (another vertex of the learned
        # shape): a handful of items may sit at 1e-4, the rest far below
        assert err.max() < 1e-4 and np.quantile(err, 0.995) < 1e-5, (err.max(), np.quantile(err, 0.995))
    keep = torch.tensor(g['keep'], device='cuda:0')
    assert bool(keep.all())  # (no item of this batch sits on the |phi| kink: the whole-batch gradients are the float32 reference too)
    total = system.contactnets_loss_and_grad(x, xp)
    assert abs(total.item() - float(g['loss_mean'])) < (1e-12 if f64 else 1e-6)
    for name, param in system.named_parameters():
        ref = g['grad/' + name]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 2e-3) * max(np.abs(ref).max(), 1.0 if f64 else 1e-12), (name, err, np.abs(ref).max())


def test_mesh_large_batch_support_points_float32_vs_float64(golden):
    """The support points of the 16,384 queries: float32 kernels against float64 kernels -- the fraction of queries whose
    float32 evaluation lands on another vertex (a mask flip) stays tiny."""
    g = golden('cube_mesh_literal')
    big = golden('cube_box_4096')
    xp64 = torch.tensor(big['x_plus'], device='cuda:0')
    s64, s32 = build(g, torch.float64), build(g, torch.float32)
    p64 = s64.support_points(xp64).cpu().numpy()
    p32 = s32.support_points(xp64.float()).cpu().double().numpy()
    err = np.abs(p64 - p32).max(-1).max(-1)
    assert (err > 1e-5).mean() < 2e-3


def test_learned_shape_export(golden, tmp_path):
    """scalars_and_meshes / generate_updated_urdfs for a DeepSupportConvex body (multibody_terms.py:536-582,
    urdf_utils.py:244-252): the mesh vertices are the network's support points over the reference's 296
    surface directions (evaluated by the HIP kernels, checked against the oracle's network)."""
    from dair_pll_amd import export
    from dair_pll_amd.urdf import parse_urdf
    g = golden('cube_mesh_literal')
    system = build(g, torch.float64)
    system.output_urdfs_dir = str(tmp_path)
    scalars, meshes = system.scalars_and_meshes()
    vertices, faces = meshes['body']
    weights = {key: torch.tensor(g['param/' + NET + 'network.' + key]) for key in
               ('hidden_weights.0', 'input_weights.0', 'input_weights.1', 'output_weight')}
    directions = torch.tensor(export.surface_directions())
    expect = O.icnn_support_point(weights, directions).numpy()
    # every vertex is one of the oracle's support points, and every support point is a vertex
    dist = np.abs(vertices[:, None, :] - expect[None, :, :]).max(axis=2)
    assert dist.min(axis=1).max() < 1e-12 and dist.min(axis=0).max() < 1e-12
    # support function of the hull = the network's value f(d) = d . grad f(d) (positive homogeneity)
    value = O.icnn_value(weights, directions).numpy()
    assert np.abs((vertices @ directions.numpy().T).max(axis=0) - value).max() < 1e-10
    normals, backwards, _ = export.outward_normals(vertices, faces)
    assert not backwards.any()
    for axis, lo, hi in zip('xyz', vertices.min(axis=0), vertices.max(axis=0)):
        assert scalars[f'body_diameter_{axis}'] == pytest.approx(hi - lo) and scalars[f'body_center_{axis}'] == pytest.approx((hi + lo) / 2)
    new = system.generate_updated_urdfs()
    spec = parse_urdf(new['model'])
    assert spec.bodies[0].geoms[0].kind == 'mesh' and spec.bodies[0].geoms[0].mesh_file == 'test.obj'
    assert np.allclose(np.array(spec.bodies[0].geoms[0].vertices), vertices, atol=0, rtol=1e-15)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_racing_copies_in_the_mesh_loss_launch(golden, dtype):
    """the loss launch of a single body with a learned shape races four copies of the cone solve like the box cube (<= 4096
    pairs): the copies read the same support points, the winner writes the witness adjoints -- loss, every gradient incl. the
    network's weights and the forces agree with the launch without copies to the solver's tolerance, bitwise from run to run"""
    g = golden('cube_box_4096')
    from dair_pll_amd import MultibodyLearnableSystem
    x = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
    out = {}
    for copies in (0, 1):
        torch.manual_seed(0)
        system = MultibodyLearnableSystem({'model': os.path.join(ASSET_DIR, 'cube_mesh.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
        system.set_solver(portfolio=copies)
        assert system.racing_copies(4096) == (4 if copies == 0 else 1) and system.racing_copies(4097) == 1
        assert system.racing_copies(4096, rollout=True) == 1
        total = system.contactnets_loss_and_grad(x, xp).item()
        grads = torch.cat([p.grad.reshape(-1).double() for p in system.parameters()]).cpu()
        again = system.contactnets_loss_and_grad(x, xp).item()
        assert again == total and torch.equal(grads, torch.cat([p.grad.reshape(-1).double() for p in system.parameters()]).cpu())
        with torch.no_grad():
            per_item = system.contactnets_loss(x, torch.zeros((x.shape[0], 0), device='cuda:0'), xp).double().cpu()
        out[copies] = (total, grads, per_item)
    tol = 2e-5 if dtype == torch.float32 else 1e-9
    assert abs(out[0][0] - out[1][0]) <= tol * abs(out[1][0])
    assert (out[0][2] - out[1][2]).abs().max() <= tol * out[1][2].abs().max()
    assert (out[0][1] - out[1][1]).abs().max() <= 50 * tol * out[1][1].abs().max()


def test_learned_shape_turned_in_its_body_exports_the_same_hull(tmp_path):
    """a learned shape whose collision <origin> carries an rpy (the reference exports any pose, urdf_utils.py:255-384): the
    hull is extracted in the geometry's own frame -- the same network gives the same vertices whether the geometry sits
    turned in its body (general build with a learned shape) or not (the specialised mesh build) -- and the written URDF
    keeps the origin"""
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.urdf import parse_urdf
    meshes = {}
    for name in ('cube_mesh', 'cube_mesh_turned'):
        torch.manual_seed(3)  # (the network is drawn at construction: the same weights for both)
        system = MultibodyLearnableSystem({'model': os.path.join(ASSET_DIR, name + '.urdf')}, 0.0068, dtype=torch.float64, device='cuda:0',
                                          output_urdfs_dir=str(tmp_path / name))
        os.makedirs(system.output_urdfs_dir, exist_ok=True)
        meshes[name] = system.extract_meshes()['body']
        written = parse_urdf(system.generate_updated_urdfs()['model'])
        assert written.bodies[0].geoms[0].kind == 'mesh'
        if name == 'cube_mesh_turned':
            assert system.spec.is_fast() is False
            assert np.allclose(written.bodies[0].geoms[0].rotation, system.spec.geoms()[0][1].rotation, atol=1e-12)
            assert np.allclose(written.bodies[0].geoms[0].origin, [0.004, -0.003, 0.002], atol=1e-15)
    plain, turned = meshes['cube_mesh'][0], meshes['cube_mesh_turned'][0]
    dist = np.abs(plain[:, None, :] - turned[None, :, :]).max(axis=2)
    assert dist.min(axis=1).max() < 1e-10 and dist.min(axis=0).max() < 1e-10


@pytest.mark.parametrize('case', CASES)
def test_mesh_terms_match_reference_run(golden, case):
    """MultibodyTerms.forward (D, M, J, phi, a) with the learned shape; the reference's top-k leaves the order of
    the four witness points unspecified, so contacts are put in a canonical order first (as for the boxes)."""
    from test_hip_parity import _canonical
    g = golden(case)
    system = build(g, torch.float64)
    xp = torch.tensor(g['x_plus'], dtype=torch.float64, device='cuda:0')
    q, v = system.space.q_v(xp)
    D, M, J, phi, a = [t.cpu().numpy() for t in system.multibody_terms(q, v, torch.zeros(q.shape[:-1] + (0,)))]
    k = phi.shape[-1]
    assert np.abs(M - g['terms/M']).max() < 1e-12 and np.abs(a - g['terms/a']).max() < 1e-9
    assert np.abs(np.sort(phi, -1) - np.sort(g['terms/phi'], -1)).max() < 1e-12
    Jm, pm, Dm = _canonical(J, phi, D, k)
    Jr, pr, Dr = _canonical(g['terms/J'], g['terms/phi'], g['terms/D'], k)
    good = np.abs(Jm - Jr).reshape(J.shape[0], -1).max(-1) < 1e-9
    assert good.mean() > 0.9
    assert np.abs(pm[good] - pr[good]).max() < 1e-12
    assert np.abs(Dm[good] - Dr[good]).max() < 1e-8 * max(1.0, np.abs(Dr).max())


@pytest.mark.parametrize('case', CASES)
def test_mesh_step_gradients_match_finite_differences(golden, case):
    """dpll_step_backward_mesh: d(sum w . x_next)/d(theta, friction, network weights) and /dx for the learned-shape
    body against central differences of dpll_step_mesh, float64 (the support point is piecewise constant in the
    state and piecewise linear in every weight tensor, so differences are exact away from mask flips)."""
    g = golden(case)
    system = build(g, torch.float64)
    x = torch.tensor(g['x'][::2], dtype=torch.float64, device='cuda:0').clone().requires_grad_(True)
    w = torch.randn(x.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64).to(x.device)
    system.zero_grad()
    (system.step(x) * w).sum().backward()

    def total(xs):
        with torch.no_grad():
            return (system.step(xs) * w).sum(-1)
    h = 1e-6
    fd = torch.zeros_like(x)
    for k in range(x.shape[1]):
        e = torch.zeros_like(x)
        e[:, k] = h
        fd[:, k] = (total(x.detach() + e) - total(x.detach() - e)) / (2 * h)
    rel = ((x.grad - fd).abs().max(-1).values / (fd.abs().max(-1).values + 1e-9)).cpu().numpy()
    assert np.median(rel) < 1e-6 and (rel < 1e-4).sum() >= len(rel) - 2, rel
    # parameters: all of theta and friction, a sample of every network tensor
    gen = torch.Generator().manual_seed(4)
    for name, p in system.named_parameters():
        flat, grad = p.data.view(-1), p.grad.reshape(-1)
        picks = range(flat.numel()) if flat.numel() <= 16 else torch.randint(0, flat.numel(), (6,), generator=gen).tolist()
        for k in picks:
            old = flat[k].item()
            flat[k] = old + 1e-6
            up = total(x.detach()).sum().item()
            flat[k] = old - 1e-6
            down = total(x.detach()).sum().item()
            flat[k] = old
            fdk = (up - down) / 2e-6
            assert abs(grad[k].item() - fdk) <= 1e-5 * max(1.0, abs(fdk)) + 1e-7 * grad.abs().max().item(), (name, k, grad[k].item(), fdk)
    # a 2-step rollout carries the graph through both steps
    system.zero_grad()
    x2 = x.detach().clone().requires_grad_(True)
    traj, _ = system.simulate(x2.unsqueeze(-2), torch.zeros((x2.shape[0], 1), device='cuda:0'), 2)
    (traj[:, -1] * w).sum().backward()
    assert x2.grad is not None and torch.isfinite(x2.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in system.parameters())


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('batch', [1, 5, 33])
def test_mesh_ragged_batches(golden, batch, case):
    """Batches that do not fill a 32-row query tile or a slab: per-item losses equal those of the full fixture
    batch, and the gradient of their mean equals the weighted full-batch gradient (float32 MFMA path and float64)."""
    g = golden(case)
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        system = build(g, dtype)
        x = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
        xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
        u = torch.zeros((x.shape[0], 0), device='cuda:0')
        full = system.contactnets_loss(x, u, xp).detach()
        part = system.contactnets_loss(x[:batch], u[:batch], xp[:batch])
        assert (part.detach() - full[:batch]).abs().max().item() <= tol * max(1.0, full.abs().max().item())
        system.zero_grad()
        part.mean().backward()
        g_part = [p.grad.clone() for p in system.parameters()]
        system.zero_grad()
        weights = torch.zeros(x.shape[0], dtype=dtype, device='cuda:0')
        weights[:batch] = 1.0 / batch
        (system.contactnets_loss(x, u, xp) * weights).sum().backward()
        for a, (name, p) in zip(g_part, system.named_parameters()):
            scale = max(p.grad.abs().max().item(), 1e-12)
            assert (a - p.grad).abs().max().item() <= (1e-9 if dtype == torch.float64 else 5e-3) * scale, (name, batch, dtype)


def test_two_learned_shapes_export(golden, tmp_path):
    """contactnets_elbow_mesh: a network per link -> a mesh per body in scalars_and_meshes, each the hull of its own
    network's support points, and an OBJ file per body beside the updated URDF."""
    from dair_pll_amd import export
    from dair_pll_amd.urdf import parse_urdf
    g = golden('elbow_mesh_literal')
    system = build(g, torch.float64)
    system.output_urdfs_dir = str(tmp_path)
    scalars, meshes = system.scalars_and_meshes()
    assert sorted(meshes) == ['elbow_1', 'elbow_2']
    directions = torch.tensor(export.surface_directions())
    for index, body in ((1, 'elbow_1'), (2, 'elbow_2')):
        weights = {key: torch.tensor(g[f'param/{GEOM}{index}.network.{key}']) for key in
                   ('hidden_weights.0', 'input_weights.0', 'input_weights.1', 'output_weight')}
        expect = O.icnn_support_point(weights, directions).numpy()
        vertices, faces = meshes[body]
        dist = np.abs(vertices[:, None, :] - expect[None, :, :]).max(axis=2)
        assert dist.min(axis=1).max() < 1e-12 and dist.min(axis=0).max() < 1e-12
        assert not export.outward_normals(vertices, faces)[1].any()
        assert scalars[f'{body}_diameter_x'] == pytest.approx(vertices[:, 0].max() - vertices[:, 0].min())
    spec = parse_urdf(system.generate_updated_urdfs()['model'])
    files = [body.geoms[0].mesh_file for body in spec.bodies]
    assert files == ['elbow_1.obj', 'elbow_2.obj']
    for body, name in zip(spec.bodies, ('elbow_1', 'elbow_2')):
        assert np.allclose(np.array(body.geoms[0].vertices), meshes[name][0], atol=0, rtol=1e-15)
        assert body.geoms[0].origin == ([0.0, 0.0, 0.0] if name == 'elbow_1' else [0.035, 0.0, 0.0])


# ---- the reference's own body-body case: two learned shapes (GeometryCollider.collide_mesh_mesh, geometry.py:585-643) --------
CLASP = 'clasp_mesh_literal'


def build_general(g, dtype):
    from dair_pll_amd import MultibodyLearnableSystem
    system = MultibodyLearnableSystem({'model': os.path.join(ASSET_DIR, 'clasp_mesh.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
    assert not system.spec.is_fast() and system.spec.pairs == [(0, 1)]
    system.load_state_dict({name: torch.tensor(g['param/' + name]) for name, _ in system.named_parameters()})
    for index, geometry in enumerate(system.multibody_terms.contact_terms.geometries):
        if index > 0:
            geometry.perturbations = torch.tensor(g[f'param/{GEOM}{index}.perturbations'], dtype=dtype, device='cuda:0')
    return system


def clasp_oracle(g):
    meshes = {}
    for index in (1, 2):
        meshes[index] = {'perturbations': torch.tensor(g[f'param/{GEOM}{index}.perturbations'])}
        for key in ('hidden_weights.0', 'input_weights.0', 'input_weights.1', 'output_weight'):
            meshes[index][key] = torch.tensor(g[f'param/{GEOM}{index}.network.{key}'])
    system = O.OracleSystem(os.path.join(ASSET_DIR, 'clasp_mesh.urdf'), float(g['dt']), mesh_params=meshes)
    system.theta = torch.tensor(g['param/multibody_terms.lagrangian_terms.inertial_parameters'])
    system.friction = torch.tensor(g['param/multibody_terms.contact_terms.friction_params'])
    return system


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_two_learned_shapes_against_the_reference_run(golden, dtype):
    """assets/clasp_mesh.urdf: a DeepSupportConvex on the base and on the tip of a two-joint arm, the pair a collision
    candidate -- the fixture was recorded from the reference's unmodified DeepSupportConvex / collide_mesh_mesh /
    extract_mesh (fcl's direction: the oracle's exact search).  On the device: vertex sets through the ICNN kernels, the
    direction by GJK / EPA in LDS, the witnesses by the networks at +-d.  Loss, every gradient incl. both networks'
    weights, step, rollout, terms."""
    g = golden(CLASP)
    system = build_general(g, dtype)
    f64 = dtype == torch.float64
    x = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
    u = torch.zeros((x.shape[0], 0), device='cuda:0')
    loss = system.contactnets_loss(x, u, xp)
    assert np.abs(loss.detach().cpu().double().numpy() - g['loss']).max() < (1e-10 if f64 else 1e-4)
    loss.mean().backward()
    for name, param in system.named_parameters():
        ref = g['grad/' + name]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 2e-3) * max(np.abs(ref).max(), 1e-12 if not f64 else 1.0), (name, err, np.abs(ref).max())
    system.zero_grad()
    total = system.contactnets_loss_and_grad(x, xp)
    assert abs(total.item() - float(g['loss_mean'])) < (1e-12 if f64 else 1e-6)
    for name, param in system.named_parameters():
        ref = g['grad/' + name]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-9 if f64 else 2e-3) * max(np.abs(ref).max(), 1e-12 if not f64 else 1.0), (name, err)
    # dynamics
    tol = (1e-10 if f64 else 1e-4) * max(1.0, np.abs(g['dynamics/x_next']).max())
    assert np.abs(system.step(x).detach().cpu().double().numpy() - g['dynamics/x_next']).max() < tol
    rows = g['simulate/rows']
    traj, _ = system.simulate(x[rows].unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), int(g['simulate/steps']))
    assert np.abs(traj.detach().cpu().double().numpy() - g['simulate/traj']).max() < (1e-9 if f64 else 5e-4) * max(1.0, np.abs(g['simulate/traj']).max())
    # terms: signed distances and Jacobian rows per geometry (order-free inside a geometry: Q3), the pair's row as it is
    q, v = system.space.q_v(xp)
    D, M, J, phi, a = system.multibody_terms(q, v, u)
    k = system.spec.n_contacts
    assert k == 9 and phi.shape == (x.shape[0], 9) and J.shape == (x.shape[0], 27, 8)
    tight = 1e-9 if f64 else 2e-4
    assert np.abs(M.cpu().double().numpy() - g['terms/M']).max() < (1e-12 if f64 else 1e-5) * max(1.0, np.abs(g['terms/M']).max())
    phi_np, ref_phi = phi.cpu().double().numpy(), g['terms/phi']
    for lo in (0, 4):
        assert np.abs(np.sort(phi_np[:, lo:lo + 4], -1) - np.sort(ref_phi[:, lo:lo + 4], -1)).max() < tight
    assert np.abs(phi_np[:, 8] - ref_phi[:, 8]).max() < tight
    J_np = J.cpu().double().numpy()
    assert np.abs(J_np[:, 8] - g['terms/J'][:, 8]).max() < tight * max(1.0, np.abs(g['terms/J']).max())  # the pair's normal row


def test_two_learned_shapes_on_random_states_against_the_oracle(golden):
    """states the fixture does not hold: the arm folded INTO the base (overlapping shapes: the EPA branch) and around
    touching, networks perturbed; float64 kernels against the oracle (hull of the Minkowski difference)."""
    g = golden(CLASP)
    system = build_general(g, torch.float64)
    oracle = clasp_oracle(g)
    gen = torch.Generator().manual_seed(11)
    n = 48
    quat = torch.randn((n, 4), generator=gen, dtype=torch.float64)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    pos = torch.cat((0.05 * torch.randn((n, 2), generator=gen, dtype=torch.float64), 0.08 + 0.05 * torch.rand((n, 1), generator=gen, dtype=torch.float64)), -1)
    # joint angles around the configurations of the fixture (where the pair is about to meet), spread so that a good part overlaps
    base = torch.tensor(g['x'][:, 7:9])
    joints = base[torch.randint(0, base.shape[0], (n,), generator=gen)] + 0.25 * torch.randn((n, 2), generator=gen, dtype=torch.float64)
    vel = torch.cat((2.0 * torch.randn((n, 3), generator=gen, dtype=torch.float64), 0.3 * torch.randn((n, 3), generator=gen, dtype=torch.float64),
                     2.0 * torch.randn((n, 2), generator=gen, dtype=torch.float64)), -1)
    xs = torch.cat((quat, pos, joints, vel), -1)
    with torch.no_grad():
        phi_ref, _ = oracle.contact_terms(xs[:, :9])
        x_next_ref = oracle.step(xs)
    assert (phi_ref[:, 8] < 0).sum() >= 5 and (phi_ref[:, 8] > 0).sum() >= 5  # both branches of the search
    xd = xs.cuda()
    q, v = system.space.q_v(xd)
    phi = system.multibody_terms(q, v, torch.zeros((n, 0), device='cuda:0'))[3].cpu()
    assert (phi[:, 8] - phi_ref[:, 8]).abs().max() < 1e-9
    x_next = system.step(xd).cpu()
    assert (x_next - x_next_ref).abs().max() < 1e-8 * max(1.0, x_next_ref.abs().max().item())
    oracle.requires_grad_()
    loss_ref = oracle.contactnets_loss(xs, x_next_ref.detach())
    loss_ref.mean().backward()
    loss = system.contactnets_loss(xd, torch.zeros((n, 0), device='cuda:0'), x_next_ref.detach().cuda())
    assert (loss.detach().cpu() - loss_ref.detach()).abs().max() < 1e-9
    loss.mean().backward()
    ref_named = oracle.named_parameters()
    for name, param in system.named_parameters():
        ref = ref_named[name].grad
        err = (param.grad.cpu() - ref).abs().max().item()
        assert err <= 1e-8 * max(ref.abs().max().item(), 1e-3), (name, err)


@pytest.mark.parametrize('case', CASES + [CLASP])
@pytest.mark.parametrize('mode', [2, 3, 4])
def test_split_bf16_gemm_forms_meet_the_float32_tolerances(golden, case, mode):
    """dpll_solver_opts_t.mesh_gemm: the ICNN GEMMs on the bf16 matrix cores with the operands split into 2 planes (three
    products per k-step, "bf16 x 3"), 3 planes (six products, f32-grade) or -- mode 4 -- 2 fp16 planes with the low one scaled by
    2^11 (three products, f32-grade: fp16 keeps 11 significand bits) -- csrc/dpll_mesh_bf16.hpp.  The float32
    tolerances of the default (exact f32 MFMA) kernels hold for both against the reference-run fixtures."""
    g = golden(case)
    system = build_general(g, torch.float32) if case == CLASP else build(g, torch.float32)
    system.set_solver(mesh_gemm=mode)
    x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
    loss = system.contactnets_loss(x, torch.zeros((x.shape[0], 0), device='cuda:0'), xp)
    assert np.abs(loss.detach().cpu().double().numpy() - g['loss']).max() < 1e-4
    loss.mean().backward()
    for name, param in system.named_parameters():
        ref = g['grad/' + name]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= 2e-3 * max(np.abs(ref).max(), 1e-12), (name, err, np.abs(ref).max())
    tol = 1e-4 * max(1.0, np.abs(g['dynamics/x_next']).max())
    assert np.abs(system.step(x).detach().cpu().double().numpy() - g['dynamics/x_next']).max() < tol


def test_split_bf16_gemm_forms_on_the_benchmark_batch(golden):
    """4096 pairs: support points that land on another vertex of the learned shape (a LeakyReLU mask flipped) and the loss /
    gradient differences against the float64 kernels, for the three GEMM forms (the numbers of DESIGN.md 5a)"""
    g = golden('cube_mesh_literal')
    big = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'cube_box_4096.npz'))
    x64, xp64 = torch.tensor(big['x'], device='cuda:0'), torch.tensor(big['x_plus'], device='cuda:0')
    s64 = build(g, torch.float64)
    p64 = s64.support_points(xp64).cpu().numpy()
    t64 = s64.contactnets_loss_and_grad(x64, xp64).item()
    g64 = {n: p.grad.cpu().numpy().copy() for n, p in s64.named_parameters()}
    # (mode 4, two fp16 planes, is held to the bars of the exact f32 kernels: its products are f32-grade)
    for mode, flips_allowed, grad_tol in ((0, 2e-4, 2e-5), (3, 2e-4, 2e-5), (4, 2e-4, 2e-5), (2, 1e-3, 5e-4)):
        s32 = build(g, torch.float32)
        s32.set_solver(mesh_gemm=mode)  # (0 is not the library's default: the fp16-plane form, 4, is)
        p32 = s32.support_points(xp64.float()).cpu().double().numpy()
        assert (np.abs(p64 - p32).max(-1) > 1e-5).mean() <= flips_allowed, mode
        t32 = s32.contactnets_loss_and_grad(x64.float(), xp64.float()).item()
        assert abs(t64 - t32) < 1e-6 * max(1.0, abs(t64)), mode
        for n, p in s32.named_parameters():
            assert np.abs(p.grad.cpu().double().numpy() - g64[n]).max() <= grad_tol * max(np.abs(g64[n]).max(), 1e-12), (mode, n)


def test_two_learned_shapes_rollout_gradients_against_oracle_autograd(golden):
    """dpll_step_backward_mesh of the general build: the gradient of a 2-step rollout with respect to every parameter (both
    networks' weights included) and to the initial state, against torch autograd through the oracle (differentiable cone
    solve; the candidate's direction a constant, as in the reference); state gradients on the unit-quaternion tangent (Q2)"""
    g = golden(CLASP)
    system = build_general(g, torch.float64)
    oracle = clasp_oracle(g).requires_grad_()
    rows = np.linspace(0, g['x'].shape[0] - 1, 16).astype(int)
    x_np = g['x'][rows]
    w = torch.rand((len(rows), 2, x_np.shape[1]), generator=torch.Generator().manual_seed(5), dtype=torch.float64) - 0.5
    x_ref = torch.tensor(x_np).requires_grad_(True)
    traj_ref = oracle.simulate(x_ref, 2)
    (traj_ref[:, 1:] * w).sum().backward()
    x = torch.tensor(x_np, device='cuda:0').requires_grad_(True)
    traj, _ = system.simulate(x.unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), 2)
    assert (traj.detach().cpu() - traj_ref.detach()).abs().max() < 1e-9
    (traj[:, 1:] * w.cuda()).sum().backward()
    ref_named = oracle.named_parameters()
    for name, param in system.named_parameters():
        ref = ref_named[name].grad.numpy()
        err = np.abs(param.grad.cpu().numpy() - ref).max()
        assert err <= 1e-7 * max(np.abs(ref).max(), 1e-3), (name, err, np.abs(ref).max())
    diff = (x.grad.cpu() - x_ref.grad).numpy()
    q = x_np[:, :4] / np.linalg.norm(x_np[:, :4], axis=-1, keepdims=True)
    diff[:, :4] -= (diff[:, :4] * q).sum(-1, keepdims=True) * q
    assert np.abs(diff).max() <= 1e-7 * x_ref.grad.abs().max().item()


@pytest.mark.parametrize('case', ['cube_mesh_literal', CLASP])
def test_mesh_workspace_is_exactly_what_the_library_asks_for(golden, case):
    """dpll_mesh_workspace_bytes: the mesh pipelines (specialised and general build) run inside EXACTLY that many bytes -- guard
    bands of canaries on both sides stay intact over ragged batch sizes, one byte less is refused -- and ragged batches give
    the per-item losses of the full batch"""
    import ctypes
    from dair_pll_amd import _capi
    g = golden(case)
    lib = _capi.library()
    for dtype in (torch.float32, torch.float64):
        system = build_general(g, dtype) if case == CLASP else build(g, dtype)
        code = _capi.F64 if dtype == torch.float64 else _capi.F32
        flat = system._packed()
        params, mesh = system._params_struct(flat), system._mesh_struct(flat)
        x_all = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
        xp_all = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
        full = system.contactnets_loss(x_all, torch.zeros((x_all.shape[0], 0), device='cuda:0'), xp_all).detach()
        for batch in (1, 3, 33, x_all.shape[0]):
            x, xp = x_all[:batch].contiguous(), xp_all[:batch].contiguous()
            need = lib.dpll_mesh_workspace_bytes(system._model(), batch, code)
            assert need > 0
            guard = 4096
            arena = torch.full((need + 2 * guard,), 0xA5, dtype=torch.uint8, device='cuda:0')
            ws = arena[guard:guard + need]
            loss = torch.zeros(batch, dtype=dtype, device='cuda:0')
            grad = torch.zeros(flat.numel(), dtype=dtype, device='cuda:0')
            total = torch.zeros(1, dtype=dtype, device='cuda:0')
            args = [system._model(), code, ctypes.byref(params), mesh, x.data_ptr(), x.stride(0), xp.data_ptr(), xp.stride(0), batch, None,
                    1.0 / batch, loss.data_ptr(), grad.data_ptr(), total.data_ptr(), None, None, ws.data_ptr()]
            _capi.check(lib.dpll_contactnets_loss_mesh(*args, need, system._stream()))
            torch.cuda.synchronize()
            assert (arena[:guard] == 0xA5).all() and (arena[guard + need:] == 0xA5).all(), (case, dtype, batch)
            assert torch.isfinite(grad).all()
            assert (loss - full[:batch]).abs().max().item() <= (1e-12 if dtype == torch.float64 else 1e-6)
            assert abs(total.item() - loss.double().mean().item()) <= (1e-12 if dtype == torch.float64 else 1e-6)
            assert lib.dpll_contactnets_loss_mesh(*args, need - 1, system._stream()) != 0
        assert lib.dpll_mesh_workspace_bytes(system._model(), 0, code) == -1


def test_fp16_planes_refuse_weights_beyond_their_range(golden):
    """mesh_gemm = 4 runs the ICNN GEMMs on two fp16 planes: fp16 ends at 65504.  set_solver checks the weights once on the host;
    weights that grow past the bound AFTERWARDS make the prep kernel turn |wout| into NaN, so that no item has a valid solve --
    every per-item loss is then NaN or masked to zero like any failed solve (multibody_learnable_system.py:186-192): never a
    finite wrong number."""
    from dair_pll_amd import _capi
    g = golden('cube_mesh_literal')
    system = build(g, torch.float32)
    net = system.multibody_terms.contact_terms.geometries[1].network
    x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
    with torch.no_grad():
        keep = net.hidden_weights[0][3, 5].item()
        net.hidden_weights[0][3, 5] = 3.0e4
    with pytest.raises(_capi.DpllError, match='fp16'):
        system.set_solver(mesh_gemm=4)
    with torch.no_grad():
        net.hidden_weights[0][3, 5] = keep
    system.set_solver(mesh_gemm=4)
    good = system.contactnets_loss(x, torch.zeros((x.shape[0], 0), device='cuda:0'), xp).detach()
    assert np.abs(good.cpu().double().numpy() - g['loss']).max() < 1e-4
    with torch.no_grad():
        net.hidden_weights[0][3, 5] = 3.0e4  # grown after the check
    bad = system.contactnets_loss(x, torch.zeros((x.shape[0], 0), device='cuda:0'), xp).detach()
    assert bool((~torch.isfinite(bad) | (bad == 0)).all())
    # a checkpoint with such a weight: load_state_dict switches the system to the f32 MFMA kernels, loudly
    state = {name: value.clone() for name, value in system.state_dict().items()}
    fresh = build(g, torch.float32)
    with pytest.warns(UserWarning, match='fp16'):
        fresh.load_state_dict(state)
    with torch.no_grad():  # (the weight is absurd, the arithmetic is not: finite losses from the f32 MFMA kernels)
        out = fresh.contactnets_loss(x, torch.zeros((x.shape[0], 0), device='cuda:0'), xp)
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) > 0.0


@pytest.mark.parametrize('case', ['cube_mesh_literal', CLASP])
def test_float32_rollout_gradients_of_both_gemm_forms_against_float64(golden, case):
    """dpll_step_backward_mesh in float32: the gradient of a 2-step rollout with respect to every parameter (network weights
    included) with the default fp16-plane form of the ICNN GEMMs and with the f32 MFMA kernels, both against the float64 kernels on
    the same states -- the adjoints of a rollout are another magnitude than those of the loss (no 1 / batch), which the scaled
    planes must not care about"""
    g = golden(case)
    make = build_general if case == CLASP else build
    rows = np.linspace(0, g['x'].shape[0] - 1, 24).astype(int)
    w = torch.rand((len(rows), 2, g['x'].shape[1]), generator=torch.Generator().manual_seed(9), dtype=torch.float64) - 0.5

    def rollout_gradients(dtype, mode):
        system = make(g, dtype)
        if mode is not None:
            system.set_solver(mesh_gemm=mode)
        x = torch.tensor(g['x'][rows], dtype=dtype, device='cuda:0')
        traj, _ = system.simulate(x.unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), 2)
        (traj[:, 1:] * w.to(dtype).cuda()).sum().backward()
        return {name: param.grad.cpu().double().numpy().copy() for name, param in system.named_parameters()}
    ref = rollout_gradients(torch.float64, None)
    for mode in (4, 0):
        mine = rollout_gradients(torch.float32, mode)
        for name, value in ref.items():
            err = np.abs(mine[name] - value).max()
            assert err <= 5e-3 * max(np.abs(value).max(), 1e-6), (mode, name, err, np.abs(value).max())
