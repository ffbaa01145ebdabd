"""Diagnostic (CPU, tests/hostsim): racing schedules for the DYNAMICS solve (forward_dynamics, eps = 1e-4) of fused rollouts.
States: the 4096 fixture states rolled out with the default solver, sampled every 10th step.  A wave of the rollout kernel
pays, per step, for its slowest item; with P copies per item a wave holds 16 / P items and an item needs the minimum
over its copies.

    python tools/diag/race_schedules_dynamics.py [cube_box_4096|elbow_box_4096] [copies] [horizon]
"""
import itertools
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import hostsim  # noqa: E402
from dair_pll_amd._capi import make_desc  # noqa: E402
from dair_pll_amd.urdf import parse_urdf  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else 'cube_box_4096'
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 4
horizon = int(sys.argv[3]) if len(sys.argv) > 3 else 80
dtype = np.float32
g = np.load(os.path.join(REPO, 'tests', 'golden', case + '.npz'))
urdf = {'contactnets_cube.urdf': 'cube.urdf', 'contactnets_elbow.urdf': 'elbow.urdf'}[str(g['urdf'])]
spec = parse_urdf(os.path.join(REPO, 'assets', urdf))
desc = make_desc(spec, float(g['dt']), str(g['inertia_mode']))
P = 'param/multibody_terms.'
theta = g[P + 'lagrangian_terms.inertial_parameters']
friction = g[P + 'contact_terms.friction_params']
lengths = np.concatenate([g[P + f'contact_terms.geometries.{i + 1}.length_params'] for i in range(spec.n_joints + 1)])
lanes = 4 * (spec.n_joints + 1)
LS = {'full': {}, 'none': dict(ls_tol=1e30, stage_ls_tol=1e30)}

x = g['x'].astype(np.float32)
states = []
for s in range(horizon):
    if s % 10 == 0:
        states.append(x.copy())
    x, _ = hostsim.step(desc, theta, friction, lengths, x, dtype=dtype)
states = np.stack(states)  # (samples, 4096, n_x)
print('sampled', states.shape, flush=True)


def run(ns, sf, ls):
    opts = hostsim.default_opts(dtype)
    opts.n_stages, opts.stage_factor = ns, sf
    for k, v in LS[ls].items():
        setattr(opts, k, v)
    return np.stack([hostsim.step(desc, theta, friction, lengths, st, dtype=dtype, opts=opts)[1] for st in states])


default = (6, 3.0, 'full')
cands = [default] + [(ns, sf, ls) for ns, sf in ((1, 1.0), (2, 10.0), (2, 30.0), (2, 100.0), (3, 5.0), (3, 10.0), (4, 3.0), (4, 5.0), (5, 2.0), (5, 3.0), (6, 2.0), (7, 2.5), (8, 2.0))
                     for ls in ('none', 'full')]
res = {}
for c in cands:
    res[c] = np.minimum(run(*c), 60)
    print(c, 'max', res[c].max(), 'mean %.2f' % res[c].mean(), flush=True)


def wave_cost(item_it, per_wave):
    n = (item_it.shape[1] // per_wave) * per_wave
    return item_it[:, :n].reshape(item_it.shape[0], -1, per_wave).max(2)  # (samples, waves)


base = wave_cost(res[default], 64 // lanes)
print('alone: mean wave-max %.2f, slowest wave (sum over samples) %d, mean item %.2f' % (base.mean(), base.sum(0).max(), res[default].mean()))
best = []
others = [c for c in cands if c != default and c[2] == 'none']
for combo in itertools.combinations(others, copies - 1):
    m = np.minimum.reduce([res[default]] + [res[c] for c in combo])
    w = wave_cost(m, 64 // (lanes * copies))
    best.append((float(w.mean()), int(w.sum(0).max()), float(m.mean()), combo))
best.sort()
for r in best[:8]:
    print(r)
