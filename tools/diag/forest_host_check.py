"""Host build of the forest program (tests/hostsim/forestsim.cpp) against every reference-run fixture: loss, every gradient,
next state, terms -- float64 and float32.  Run from the repository root."""
import os
import sys

import numpy as np

sys.path.insert(0, 'tests')
sys.path.insert(0, '.')
from hostsim import forest  # noqa: E402
from dair_pll_amd import _capi  # noqa: E402
from dair_pll_amd.urdf import build_system_spec, parse_urdf  # noqa: E402

P = 'multibody_terms.'
SOURCES = {'polycube': ('cube_mesh.urdf', 'polygon'), 'wedge': ('wedge.urdf', 'polygon'), 'clasp_ball': ('clasp_ball.urdf', 'polygon'),
           'cube_box': ('cube.urdf', 'deep_support'), 'elbow_box': ('elbow.urdf', 'deep_support')}
MODELS = ['cube_box', 'elbow_box', 'chain3', 'vee', 'ballcube', 'mace', 'polycube', 'wedge', 'clasp', 'clasp_ball', 'vee_pair', 'gripper', 'crank',
          'pincer', 'grasp', 'slider']


def fixture_params(g, system):
    theta, friction, lengths = forest.params_of(system)
    theta = g['param/' + P + 'lagrangian_terms.inertial_parameters'].copy()
    friction = g['param/' + P + 'contact_terms.friction_params'].copy()
    for index, (_, geom) in enumerate(system.geoms()):
        key = {'box': 'length_params', 'sphere': 'length_param', 'polygon': 'vertices'}[geom.kind]
        value = g['param/' + P + f'contact_terms.geometries.{index + 1}.{key}'].ravel()
        lengths[index, :len(value)] = value
    return theta, friction, lengths


def reference_gradient(g, system):
    n_b, n_g = system.n_bodies, len(system.geoms())
    out = np.zeros(10 * n_b + 1 + n_g + 24 * n_g)
    out[:10 * n_b] = g['grad/' + P + 'lagrangian_terms.inertial_parameters'].ravel()
    out[10 * n_b:10 * n_b + 1 + n_g] = g['grad/' + P + 'contact_terms.friction_params']
    for index, (_, geom) in enumerate(system.geoms()):
        key = {'box': 'length_params', 'sphere': 'length_param', 'polygon': 'vertices'}[geom.kind]
        value = g['grad/' + P + f'contact_terms.geometries.{index + 1}.{key}'].ravel()
        at = 10 * n_b + 1 + n_g + 24 * index
        out[at:at + len(value)] = value
    return out


FOREST = {'chain6': {'chain6': 'chain6.urdf'}, 'rake': {'rake': 'rake.urdf'}, 'two_cubes': {'cube_a': 'cube.urdf', 'cube_b': 'cube.urdf'},
          'pendulum_cube': {'pendulum': 'pendulum.urdf', 'cube': 'cube.urdf'}}
MODELS += list(FOREST)
worst = {}
for name in (sys.argv[1:] or MODELS):
    urdf, representation = SOURCES.get(name, (name + '.urdf', 'deep_support'))
    g = np.load(os.path.join('tests', 'golden', name + '_literal.npz'))
    models = FOREST.get(name, {name: urdf})
    system = build_system_spec({key: parse_urdf(os.path.join('assets', value), representation) for key, value in models.items()})
    desc = _capi.make_forest_desc(system, float(g['dt']), str(g['inertia_mode']))
    theta, friction, lengths = fixture_params(g, system)
    out = forest.loss(desc, theta, friction, lengths, g['x'], g['x_plus'])
    ref_grad = reference_gradient(g, system)
    e_loss = np.abs(out['loss'] - g['loss']).max()
    e_grad = np.abs(out['grad'] - ref_grad).max() / max(1.0, np.abs(ref_grad).max())
    x_next, iters = forest.step(desc, theta, friction, lengths, g['x'])
    ref_next = g['dynamics/x_next']
    e_step = np.abs(x_next - ref_next).max() / max(1.0, np.abs(ref_next).max())
    M, a, phi, J = forest.terms(desc, theta, friction, lengths, g['x_plus'])
    e_M, e_a = np.abs(M - g['terms/M']).max(), np.abs(a - g['terms/a']).max() / max(1.0, np.abs(g['terms/a']).max())
    e_phi = np.abs(np.sort(phi, -1) - np.sort(g['terms/phi'], -1)).max()
    o32 = forest.loss(desc, theta, friction, lengths, g['x'], g['x_plus'], dtype=np.float32)
    n32, _ = forest.step(desc, theta, friction, lengths, g['x'], dtype=np.float32)
    e_loss32 = np.abs(o32['loss'] - g['loss']).max()
    e_grad32 = np.abs(o32['grad'] - ref_grad).max() / max(1.0, np.abs(ref_grad).max())
    e_step32 = np.abs(n32 - ref_next).max() / max(1.0, np.abs(ref_next).max())
    print(f'{name:11s} nb {desc.n_bodies} nv {desc.n_v} K {desc.n_contacts:2d} | f64 loss {e_loss:.1e} grad {e_grad:.1e} step {e_step:.1e} M {e_M:.1e} a {e_a:.1e} phi {e_phi:.1e} '
          f'iters {out["iters"].max()}/{iters.max()} | f32 loss {e_loss32:.1e} grad {e_grad32:.1e} step {e_step32:.1e}')
    worst[name] = (e_loss, e_grad, e_step)
bad = {k: v for k, v in worst.items() if max(v) > 1e-8}
print('FAIL' if bad else 'ok', bad)
