"""Diagnostic (CPU, tests/hostsim): picks the racing schedules of the loss solve (dpll_solver_opts_t.race_*).  Every candidate
schedule (warm start, stages, factor, line-search mode) is run alone on a fixture; a portfolio's iteration count per item is
the minimum over its copies (the copies' paths are independent: only the stopping time is shared), and a wave pays for the
fall-back step of its line search whenever any live copy of any of its items rejects the full step.

    python tools/diag/race_schedules.py [cube_box_4096|elbow_box_4096] [copies]
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
dtype = np.float32
g = np.load(os.path.join(REPO, 'tests', 'golden', case + '.npz'))
urdf = {'contactnets_cube.urdf': 'cube.urdf', 'contactnets_elbow.urdf': 'elbow.urdf'}[str(g['urdf'])]
spec = parse_urdf(os.path.join(REPO, 'assets', urdf))
desc = make_desc(spec, float(g['dt']), str(g['inertia_mode']))
P = 'param/multibody_terms.'
theta = g[P + 'lagrangian_terms.inertial_parameters']
friction = g[P + 'contact_terms.friction_params']
lengths = np.concatenate([g[P + f'contact_terms.geometries.{i + 1}.length_params'] for i in range(spec.n_joints + 1)])
items_per_wave = 64 // (4 * (spec.n_joints + 1) * copies)
LS = {'full': {}, 'capped': dict(max_ls=1, stage_max_ls=1), 'none': dict(ls_tol=1e30, stage_ls_tol=1e30)}


def run(ws, ns, sf, ls):
    opts = hostsim.default_opts(dtype)
    opts.warm_start, opts.n_stages, opts.stage_factor, opts.loss_n_stages = ws, ns, sf, 0
    for k, v in LS[ls].items():
        setattr(opts, k, v)
    out = hostsim.loss(desc, theta, friction, lengths, g['x'], g['x_plus'], dtype=dtype, opts=opts)
    it = out['iters'].copy()
    masks = np.zeros(len(it), dtype=np.uint64)
    hostsim.lib().hostsim_reject_masks(masks.ctypes.data_as(hostsim.c_void_p), hostsim.c_int64(len(it)))
    bits = ((masks[:, None] >> np.arange(48, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
    err = np.abs(out['loss'] - g['loss'])
    return it, bits, err


default = (0, 5, 2.5, 'full') if spec.n_joints == 1 else (0, 6, 3.0, 'full')
cands = [default]
for ws in (0, 1):
    for ns, sf in ((1, 1.0), (2, 5.0), (2, 10.0), (2, 30.0), (3, 3.0), (3, 5.0), (3, 10.0), (4, 3.0), (4, 4.0), (5, 2.0), (5, 2.5), (6, 2.5), (7, 2.5), (8, 2.0)):
        for ls in ('full', 'capped', 'none'):
            cands.append((ws, ns, sf, ls))
res = {}
for c in cands:
    it, bits, err = run(*c)
    ok = err.max() < 2e-6 and it.max() < 60
    res[c] = (it, bits, ok)
    print(c, 'max', it.max(), 'mean %.2f' % it.mean(), 'err %.1e' % err.max(), 'rejects %.3f' % (bits.sum() / it.sum()), '' if ok else ' (not converged everywhere: may only race, never run alone)', flush=True)


def cost(combo):
    its = np.stack([res[c][0] for c in combo])            # (copies, items)
    item_it = its.min(0)
    live = np.arange(48)[None, :] < item_it[:, None]      # (items, iteration): the item is still iterating
    rej = np.zeros_like(live)
    for c in combo:
        rej |= res[c][1][:, :48]
    rej &= live
    n = (len(item_it) // items_per_wave) * items_per_wave
    w_it = item_it[:n].reshape(-1, items_per_wave).max(1)
    w_fb = rej[:n].reshape(-1, items_per_wave, 48).any(1).sum(1)
    ticks = w_it * 2700 + w_fb * 600
    return int(ticks.max()), float(ticks.mean()), int(item_it.max()), float(item_it.mean()), float(w_fb.mean())


print('alone:', cost((default,)) if copies == 1 else (res[default][0].max(), res[default][0].mean()))
others = [c for c in cands if c != default]
best = []
for combo in itertools.combinations(others, copies - 1):
    best.append((cost((default,) + combo), combo))
best.sort(key=lambda r: (r[0][0], r[0][1]))
for r in best[:12]:
    print(r)
