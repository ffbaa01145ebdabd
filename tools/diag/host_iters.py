"""Diagnostic (CPU, tests/hostsim): Newton-iteration histogram of the loss solve on a fixture for a list of solver
settings -- the experiment bench for warm starts / continuation schedules before they go to the GPU.

    python tools/diag/host_iters.py [cube_box_4096|elbow_box_4096] [f32|f64]
"""
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
dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == 'f32') else np.float64
g = np.load(os.path.join(REPO, 'tests', 'golden', case + '.npz'))
urdf = {'contactnets_cube.urdf': 'cube.urdf', 'contactnets_elbow.urdf': 'elbow.urdf'}[str(g['urdf'])]
spec = parse_urdf(os.path.join(REPO, 'assets', urdf))
desc = make_desc(spec, float(g['dt']), str(g['inertia_mode']))
P = 'param/multibody_terms.'
theta = g[P + 'lagrangian_terms.inertial_parameters']
friction = g[P + 'contact_terms.friction_params']
lengths = np.concatenate([g[P + f'contact_terms.geometries.{i + 1}.length_params'] for i in range(spec.n_joints + 1)])


def run(**kw):
    opts = hostsim.default_opts(dtype)
    for k, v in kw.items():
        setattr(opts, k, v)
    out = hostsim.loss(desc, theta, friction, lengths, g['x'], g['x_plus'], dtype=dtype, opts=opts)
    it = out['iters']
    err = np.abs(out['loss'] - g['loss']).max()
    ipw = 64 // (4 * (spec.n_joints + 1))
    waves = it[:(len(it) // ipw) * ipw].reshape(-1, ipw).max(-1)  # what a 64-lane wave pays: its slowest item
    hist = np.bincount(it, minlength=it.max() + 1)
    masks = np.zeros(len(it), dtype=np.uint64)
    hostsim.lib().hostsim_reject_masks(masks.ctypes.data_as(hostsim.c_void_p), hostsim.c_int64(len(it)))
    bits = ((masks[:, None] >> np.arange(40, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)  # (items, iteration)
    live = np.arange(40)[None, :] < it[:, None]
    n_w = (len(it) // ipw) * ipw
    wave_live = live[:n_w].reshape(-1, ipw, 40).any(1)
    wave_rej = bits[:n_w].reshape(-1, ipw, 40).any(1)
    print(f'   partial steps: {bits.sum()} of {live.sum()} item-iterations ({bits.sum() / live.sum():.3f}); '
          f'{wave_rej.sum()} of {wave_live.sum()} wave-iterations ({wave_rej.sum() / wave_live.sum():.3f}); by iteration '
          f'{[round(float(wave_rej[:, i].sum() / max(1, wave_live[:, i].sum())), 2) for i in range(16)]}')
    print(f'{kw}: max {it.max()} mean {it.mean():.2f} wave-mean {waves.mean():.2f} err {err:.1e} hist {hist.tolist()}')
    return out


if __name__ == '__main__':
    run()
    if len(sys.argv) > 3 and sys.argv[3] == 'sweep':
        for ws in (0, 1):
            for ns in (1, 2, 3, 4, 5, 6):
                run(warm_start=ws, n_stages=ns)
    if len(sys.argv) > 3 and sys.argv[3] == 'tol':
        for tol in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
            run(tol=tol)
            run(tol=tol, stall_tol=max(tol, 1e-5))
