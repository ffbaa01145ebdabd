"""Links welded together by `fixed` joints keep a row of `inertial_parameters` each.

Drake keeps a welded link as a body of its own, so the reference's parameter tree has one theta row per LINK
(multibody_terms.py:161-207, drake_utils.py:129-146); the kernels' bodies are the links that move against each other.
`assets/welded_arm.urdf` has five links (base + bracket welded on, turned; arm on a hinge off the bracket; tip welded to the
arm; sensor welded to the tip) = five rows, two kernel bodies.  Fixtures `welded_arm_{literal, physical}.npz` were recorded by
running the reference's own MultibodyTerms / contactnets_loss / forward_dynamics / simulate on it (oracle/gen_golden.py
record_welded_case: LagrangianTerms.forward converts every row and hands all five to the closures, which sum over the rows
with a welded link riding on its host).  The product path composes the rows into the bodies' inertial vectors on the device
(csrc/dpll_weld.hip) and chains the gradient back.  tests/test_general_models.py runs the fixture through the oracle, the loss
/ gradients / dynamics / terms GPU tests; here: the host logic, the composition itself, the contracts around it."""
import os

import numpy as np
import pytest
import torch

from conftest import ASSET_DIR
from dair_pll_amd import MultibodyLearnableSystem, _capi
from dair_pll_amd.inertia import pi_cm_to_pi_o, pi_cm_to_theta, theta_to_pi_cm
from dair_pll_amd.urdf import parse_urdf
from oracle import dpll_oracle as O

URDF = os.path.join(ASSET_DIR, 'welded_arm.urdf')
P = 'multibody_terms.'
THETA = P + 'lagrangian_terms.inertial_parameters'


def test_rows_are_the_links_in_document_order():
    spec = parse_urdf(URDF)
    assert [body.name for body in spec.bodies] == ['base', 'arm'] and spec.has_welded_rows() and not spec.is_fast()
    rows = spec.inertia_rows()
    assert [(row.name, row.body) for row in rows] == [('base', 0), ('bracket', 0), ('arm', 1), ('tip', 1), ('sensor', 1)]
    assert [row.mass for row in rows] == [0.3, 0.1, 0.05, 0.02, 0.01]  # every link's OWN mass, nothing folded in
    # a link welded to a welded link: the two joint <origin>s composed
    from dair_pll_amd.urdf import _rotation
    import xml.etree.ElementTree as ET
    r_tip = np.array(_rotation(ET.fromstring('<origin rpy="0 0.3 0"/>')))
    r_sensor = np.array(_rotation(ET.fromstring('<origin rpy="0.1 0 0"/>')))
    assert np.abs(np.array(rows[4].origin) - (np.array([0.06, 0, 0.005]) + r_tip @ np.array([0.01, 0, 0]))).max() < 1e-15
    assert np.abs(np.array(rows[4].rotation) - r_tip @ r_sensor).max() < 1e-15
    # the same reading in the oracle (its own parser)
    mine = O.inertia_rows(O.parse_urdf(URDF))
    assert [(row['name'], row['body']) for row in mine] == [(row.name, row.body) for row in rows]
    for a, b in zip(mine, rows):
        assert np.abs(np.array(a['origin']) - b.origin).max() < 1e-15 and np.abs(np.array(a['rot']) - b.rotation).max() < 1e-15
    # the base's box and the arm's box: not adjacent in Drake (the hinge joins BRACKET and arm) -> a collision candidate;
    # the bracket's sphere and the arm's box are filtered by the hinge
    assert spec.pairs == [(0, 2)] and O.parse_urdf(URDF)['pairs'] == [(1, 3)]
    # a model without welded links: one row per body, as ever
    plain = parse_urdf(os.path.join(ASSET_DIR, 'chain3.urdf'))
    assert not plain.has_welded_rows() and [row.name for row in plain.inertia_rows()] == [body.name for body in plain.bodies]


def iota_of(pi_cm_row, physical: bool):
    """[m, m c, I about the origin] the kernels work with (csrc/dpll_terms.hpp theta_to_iota): physical, or the reference's
    literal reading (the rotational inertia handed over is I_cm / m, DESIGN Q1)"""
    pi = np.array(pi_cm_row, dtype=np.float64)
    if not physical:
        pi = pi.copy()
        pi[4:] = pi[4:] / pi[0]
    return pi_cm_to_pi_o(pi)


@pytest.mark.parametrize('physical', [True, False])
def test_composition_is_the_sum_of_the_transformed_rows(physical):
    """X_r (dair_pll_amd/_capi.py weld_transform) against first principles: the inertia about the body's origin of point
    masses that realise each link's (m, c, I_cm), placed in the body's frame -- and, in physical mode, against the composite the
    parser folds at parse time (parallel axes, dair_pll_amd/urdf.py _weld_fixed_joints)"""
    spec = parse_urdf(URDF)
    rows = spec.inertia_rows()
    composed = np.zeros((len(spec.bodies), 10))
    direct = np.zeros((len(spec.bodies), 10))
    for row in rows:
        pi_cm = [row.mass] + [row.mass * c for c in row.com] + list(row.inertia_cm)
        composed[row.body] += _capi.weld_transform(row.rotation, row.origin) @ iota_of(pi_cm, physical)
        # first principles: I_o' = R (I_cm [/ m]) R^T + m (|c'|^2 1 - c' c'^T), c' = R c + t
        R, t = np.array(row.rotation), np.array(row.origin)
        ixx, iyy, izz, ixy, ixz, iyz = row.inertia_cm
        I_cm = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]]) / (1.0 if physical else row.mass)
        c = R @ np.array(row.com) + t
        I_o = R @ I_cm @ R.T + row.mass * (c @ c * np.eye(3) - np.outer(c, c))
        direct[row.body] += np.concatenate(([row.mass], row.mass * c, [I_o[0, 0], I_o[1, 1], I_o[2, 2], I_o[0, 1], I_o[0, 2], I_o[1, 2]]))
    assert np.abs(composed - direct).max() < 1e-17
    if physical:
        for index, body in enumerate(spec.bodies):  # the parse-time composite of the body
            folded = pi_cm_to_pi_o(np.array([body.mass] + [body.mass * c for c in body.com] + list(body.inertia_cm)))
            assert np.abs(composed[index] - folded).max() < 1e-17


def test_parameter_tree_has_the_reference_shape(golden, tmp_path):
    """`load_state_dict` of a reference-shaped checkpoint succeeds: (5, 10) inertial_parameters, the reference's key names;
    the scalar summary and the written URDF carry every link"""
    g = golden('welded_arm_literal')
    system = MultibodyLearnableSystem({'welded_arm': URDF}, float(g['dt']), dtype=torch.float64, device='cpu',
                                      output_urdfs_dir=str(tmp_path))
    names = dict(system.named_parameters())
    assert names[THETA].shape == (5, 10) and set(names) == {key[len('param/'):] for key in g.files if key.startswith('param/')}
    # the initial rows are the links' own URDF values in the reference's theta format
    assert np.abs(names[THETA].detach().numpy() - g['param/' + THETA]).max() < 1e-12
    checkpoint = {key: torch.tensor(g['param/' + key]) + 0.01 for key in names}
    system.load_state_dict(checkpoint)
    assert torch.equal(system.state_dict()[THETA], checkpoint[THETA])
    scalars = system.scalars()
    pi_cm = theta_to_pi_cm(checkpoint[THETA][3].numpy())
    assert scalars['tip_m'] == pytest.approx(pi_cm[0]) and scalars['sensor_I_zz'] == pytest.approx(theta_to_pi_cm(checkpoint[THETA][4].numpy())[6])
    assert 'base_len_x' in scalars and 'arm_mu' in scalars and 'bracket_m' in scalars
    written = system.generate_updated_urdfs()['welded_arm']
    again = parse_urdf(written)
    for before, after in zip(theta_to_pi_cm_rows(checkpoint[THETA].numpy()), again.inertia_rows()):
        assert after.mass == pytest.approx(before[0], rel=1e-12)
        assert np.abs(np.array(after.com) - before[1:4] / before[0]).max() < 1e-12
        assert np.abs(np.array(after.inertia_cm) - before[4:]).max() < 1e-12
    # the bracket's sphere stays the bracket's <collision>, the base keeps one box
    import xml.etree.ElementTree as ET
    links = {link.get('name'): link for link in ET.parse(written).getroot().findall('link')}
    assert len(links['base'].findall('collision')) == 1 and links['bracket'].find('collision/geometry/sphere') is not None
    assert links['tip'].find('collision') is None


def theta_to_pi_cm_rows(theta):
    return [theta_to_pi_cm(row) for row in theta]


def test_library_exports_the_weld_calls():
    lib = _capi.library()
    assert hasattr(lib, 'dpll_weld_compose') and hasattr(lib, 'dpll_weld_compose_backward')
    # (argument checks need no GPU)
    assert lib.dpll_weld_compose(0, _capi.INERTIA_COMPOSED, 2, 1, None, None, None, None, None) == -1
    assert b'inertia_mode' in lib.dpll_last_error()
    assert lib.dpll_weld_compose(0, 0, 1, 2, None, None, None, None, None) == -1  # fewer rows than bodies


# ---- GPU ------------------------------------------------------------------------------------------------------------
def gpu_system(g, dtype, inertia_mode='reference_literal', build='auto'):
    system = MultibodyLearnableSystem({'welded_arm': URDF}, float(g['dt']), dtype=dtype, device='cuda:0', inertia_mode=inertia_mode,
                                      build=build)
    system.load_state_dict({key: torch.tensor(g['param/' + key]) for key, _ in system.named_parameters()})
    return system


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['reference_literal', 'physical'])
def test_gpu_composed_vectors_and_their_chain(golden, mode):
    """dpll_weld_compose on the device against the numpy composition above; dpll_weld_compose_backward against torch autograd
    through a float64 restatement of theta -> iota -> X iota"""
    g = golden('welded_arm_literal' if mode == 'reference_literal' else 'welded_arm_physical')
    for dtype in (torch.float64, torch.float32):
        system = gpu_system(g, dtype, mode)
        flat = system._packed()
        theta = torch.tensor(g['param/' + THETA], requires_grad=True)
        rows = system.spec.inertia_rows()
        inertia = O.theta_to_spatial_inertia(theta)  # [m, c, I_cm / m]
        composed = []
        for b in range(2):
            total = 0
            for r, row in enumerate(rows):
                if row.body != b:
                    continue
                m, c, i_cm = inertia[r, 0:1], inertia[r, 1:4], inertia[r, 4:]
                if mode == 'physical':
                    i_cm = i_cm * m
                s = O.skew(c)
                i_o = O._inertia_matrix(i_cm) - m * (s @ s)
                iota = torch.cat((m, m * c, torch.stack((i_o[0, 0], i_o[1, 1], i_o[2, 2], i_o[0, 1], i_o[0, 2], i_o[1, 2]))))
                total = total + torch.tensor(_capi.weld_transform(row.rotation, row.origin)) @ iota
            composed.append(total)
        composed = torch.stack(composed)
        mine = flat[:20].view(2, 10).double().cpu()
        assert (mine - composed.detach()).abs().max() < (1e-15 if dtype == torch.float64 else 1e-7)
        w = torch.rand((2, 10), dtype=torch.float64, generator=torch.Generator().manual_seed(3)) - 0.5
        (composed * w).sum().backward()
        chained = system._weld_call(True, w.to(dtype).cuda().reshape(-1).contiguous())
        assert (chained.double().cpu() - theta.grad).abs().max() < (1e-14 if dtype == torch.float64 else 1e-6) * theta.grad.abs().max()


@pytest.mark.gpu
def test_gpu_physical_mode_fixture(golden):
    g = golden('welded_arm_physical')
    system = gpu_system(g, torch.float64, 'physical')
    x, xp = torch.tensor(g['x'], device='cuda:0'), torch.tensor(g['x_plus'], device='cuda:0')
    total = system.contactnets_loss_and_grad(x, xp)
    assert abs(total.item() - float(g['loss_mean'])) < 1e-12
    for key, param in system.named_parameters():
        ref = g['grad/' + key]
        assert param.grad.shape == ref.shape
        assert np.abs(param.grad.cpu().numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), key
    assert np.abs(system.step(x).detach().cpu().numpy() - g['dynamics/x_next']).max() < 1e-10 * max(1.0, np.abs(g['dynamics/x_next']).max())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_gpu_forest_build_takes_the_same_rows(golden, dtype):
    """the forest build (csrc/dpll_forest.hip) with composed rows against the reference run"""
    g = golden('welded_arm_literal')
    system = gpu_system(g, dtype, build='forest')
    assert system.forest
    f64 = dtype == torch.float64
    x, xp = torch.tensor(g['x'], dtype=dtype, device='cuda:0'), torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
    total = system.contactnets_loss_and_grad(x, xp)
    assert abs(total.item() - float(g['loss_mean'])) < (1e-11 if f64 else 1e-6)
    for key, param in system.named_parameters():
        ref = g['grad/' + key]
        err = np.abs(param.grad.cpu().double().numpy() - ref).max()
        assert err <= (1e-8 if f64 else 5e-3) * max(np.abs(ref).max(), 1.0 if f64 else 1e-3), (key, err)


@pytest.mark.gpu
def test_gpu_training_moves_every_row_and_the_fused_step_refuses(golden):
    g = golden('welded_arm_literal')
    system = gpu_system(g, torch.float64)
    x, xp = torch.tensor(g['x'], device='cuda:0'), torch.tensor(g['x_plus'], device='cuda:0')
    theta = system.multibody_terms.lagrangian_terms.inertial_parameters
    before = theta.detach().clone()
    optimizer = torch.optim.Adam(system.parameters(), lr=1e-3)
    losses = []
    for _ in range(5):
        optimizer.zero_grad()
        losses.append(system.contactnets_loss_and_grad(x, xp).item())
        optimizer.step()
    assert losses[-1] < losses[0] and ((theta.detach() - before).abs().max(dim=1).values > 1e-4).all()
    # accumulate=True adds to the rows' gradient as to every other parameter's
    system.zero_grad()
    system.contactnets_loss_and_grad(x, xp)
    once = theta.grad.clone()
    system.contactnets_loss_and_grad(x, xp, accumulate=True)
    assert (theta.grad - 2 * once).abs().max() < 1e-12 * once.abs().max()
    from dair_pll_amd.system import FusedAdamState
    with pytest.raises(NotImplementedError):
        system.contactnets_train_step(x, xp, FusedAdamState())
    # rollouts with gradients through the steps (dpll_step_backward -> chain): against torch autograd through the oracle's step
    rows = np.linspace(0, x.shape[0] - 1, 8).astype(int)
    system.load_state_dict({key: torch.tensor(g['param/' + key]) for key, _ in system.named_parameters()})
    system.zero_grad()
    x0 = x[rows].clone()
    w = torch.rand(x0.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(5)) - 0.5
    (system.step(x0) * w.cuda()).sum().backward()
    oracle = O.OracleSystem(URDF, float(g['dt']))
    oracle.theta = torch.tensor(g['param/' + THETA])
    oracle.friction = torch.tensor(g['param/' + P + 'contact_terms.friction_params'])
    for index, params in enumerate(oracle.geom_params):
        for key in list((params or {}).keys()):
            params[key] = torch.tensor(g['param/' + P + f'contact_terms.geometries.{index}.{key}'])
    oracle.requires_grad_()
    (oracle.step(torch.tensor(g['x'][rows])) * w).sum().backward()
    ref = oracle.named_parameters()[THETA].grad
    assert (theta.grad.cpu() - ref).abs().max() <= 1e-7 * max(ref.abs().max().item(), 1e-3)


def test_host_build_of_the_composition_and_of_the_composed_model(golden):
    """The code the device runs, on the host (tests/hostsim compiles csrc/dpll_weld.hpp and csrc/dpll_core.hpp): (1) the
    composition against the numpy restatement above and its backward against central differences; (2) the per-item math of the
    general build with a DPLL_INERTIA_COMPOSED description -- loss, the gradient chained back to the five rows, the next
    state -- against the reference run."""
    import hostsim
    from test_general_models import fixture_params, reference_gradient
    for mode, physical, fixture in ((0, False, 'welded_arm_literal'), (1, True, 'welded_arm_physical')):
        g = golden(fixture)
        spec = parse_urdf(URDF)
        rows = spec.inertia_rows()
        host = [row.body for row in rows]
        X = np.stack([_capi.weld_transform(row.rotation, row.origin) for row in rows])
        theta_rows = g['param/' + THETA]
        iota = hostsim.weld_compose(mode, host, X, theta_rows, len(spec.bodies))
        expected = np.zeros_like(iota)
        for r, row in enumerate(rows):
            expected[row.body] += X[r] @ iota_of(theta_to_pi_cm(theta_rows[r]), physical)
        assert np.abs(iota - expected).max() < 1e-15
        w = np.random.default_rng(0).standard_normal(iota.shape)
        chained = hostsim.weld_backward(mode, host, X, theta_rows, w)
        for r in range(len(rows)):
            for c in range(10):
                e = np.zeros_like(theta_rows)
                e[r, c] = 1e-6
                fd = ((hostsim.weld_compose(mode, host, X, theta_rows + e, 2) - hostsim.weld_compose(mode, host, X, theta_rows - e, 2)) * w).sum() / 2e-6
                assert abs(fd - chained[r, c]) <= 1e-7 * max(1.0, np.abs(chained).max()), (r, c)
        # the composed model through the kernels' per-item math
        desc = make_desc_of(spec, float(g['dt']), str(g['inertia_mode']))
        assert desc.inertia_mode == _capi.INERTIA_COMPOSED
        _, friction, lengths = fixture_params(g, spec)
        out = hostsim.loss(desc, iota, friction, lengths, g['x'], g['x_plus'])
        assert np.abs(out['loss'] - g['loss']).max() < 1e-12
        n_b = len(spec.bodies)
        grad_rows = hostsim.weld_backward(mode, host, X, theta_rows, out['grad'][:10 * n_b].reshape(n_b, 10))
        ref = g['grad/' + THETA]
        assert np.abs(grad_rows - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
        named = {key: g[key] for key in g.files if key.startswith('grad/')}
        named['grad/' + THETA] = np.zeros((n_b, 10))  # (the theta block of the flat layout is d / d iota here: compared above)
        rest = reference_gradient(named, spec)
        assert np.abs(out['grad'][10 * n_b:] - rest[10 * n_b:]).max() <= 1e-9 * max(1.0, np.abs(rest).max())
        x_next, iters = hostsim.step(desc, iota, friction, lengths, g['x'])
        assert iters.max() < 100 and np.abs(x_next - g['dynamics/x_next']).max() < 1e-10 * max(1.0, np.abs(g['dynamics/x_next']).max())


def make_desc_of(spec, dt, inertia_mode):
    from dair_pll_amd._capi import make_desc
    return make_desc(spec, dt, inertia_mode)


def test_a_welded_frame_without_mass_has_no_row(tmp_path):
    """a `fixed` joint to a link without mass and inertia (a frame: a sensor mount, a tool centre point) -- the reference's theta
    has no finite value for it (log m), so it carries no row; its geometry still rides on its host.  A massless link WITH inertia
    is refused, and so is it by the oracle's parser."""
    link = ('<link name="{name}"><inertial><origin xyz="0 0 0"/><mass value="{m}"/>'
            '<inertia ixx="{i}" iyy="{i}" izz="{i}" ixy="0" ixz="0" iyz="0"/></inertial>{col}</link>')
    box = ('<collision><geometry><box size="0.1 0.06 0.04"/></geometry><drake:proximity_properties><drake:mu_static value="0.3"/>'
           '</drake:proximity_properties></collision>')
    ball = ('<collision><origin xyz="0.01 0 0"/><geometry><sphere radius="0.01"/></geometry><drake:proximity_properties>'
            '<drake:mu_static value="0.2"/></drake:proximity_properties></collision>')

    def urdf(frame_inertia):
        return ('<?xml version="1.0"?><robot name="framed" xmlns:drake="https://drake.mit.edu/">'
                + link.format(name='base', m=0.3, i=2e-4, col=box) + link.format(name='frame', m=0.0, i=frame_inertia, col=ball)
                + link.format(name='lump', m=0.05, i=1e-5, col='')
                + '<joint name="a" type="fixed"><parent link="base"/><child link="frame"/><origin xyz="0.05 0 0.02" rpy="0 0 0.3"/></joint>'
                + '<joint name="b" type="fixed"><parent link="frame"/><child link="lump"/><origin xyz="0 0.01 0"/></joint></robot>')
    path = tmp_path / 'framed.urdf'
    path.write_text(urdf(0.0))
    spec = parse_urdf(str(path))
    assert [body.name for body in spec.bodies] == ['base'] and spec.welded == {'frame': 'base', 'lump': 'base'}
    assert [(row.name, row.body) for row in spec.inertia_rows()] == [('base', 0), ('lump', 0)] and spec.has_welded_rows()
    assert [geom.kind for geom in spec.bodies[0].geoms] == ['box', 'sphere'] and spec.bodies[0].geoms[1].link == 'frame'
    # the lump sits in the base through both joints
    from dair_pll_amd.urdf import _rotation
    import xml.etree.ElementTree as ET
    R = np.array(_rotation(ET.fromstring('<origin rpy="0 0 0.3"/>')))
    assert np.abs(np.array(spec.inertia_rows()[1].origin) - (np.array([0.05, 0, 0.02]) + R @ np.array([0, 0.01, 0]))).max() < 1e-15
    assert [(row['name'], row['body']) for row in O.inertia_rows(O.parse_urdf(str(path)))] == [('base', 0), ('lump', 0)]
    system = MultibodyLearnableSystem({'framed': str(path)}, 0.0068, device='cpu')
    assert system.multibody_terms.lagrangian_terms.inertial_parameters.shape == (2, 10) and 'lump_m' in system.scalars()
    path.write_text(urdf(1e-6))
    with pytest.raises(ValueError, match='welded link'):
        parse_urdf(str(path)).inertia_rows()
    with pytest.raises(AssertionError):
        O.parse_urdf(str(path))
