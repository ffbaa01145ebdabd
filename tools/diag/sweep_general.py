"""Diagnostic: continuation schedules of the general build's loss solve (n_stages, stage_factor, stage_max_iter) on a few models at
B = 4096 (tiled fixture pairs), float32: time per loss + gradient call, worst / mean iterations, largest loss error against the
fixture.  Run on the MI355X: python tools/diag/sweep_general.py [model ...]"""
import itertools, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
MODELS = {'slider': 'slider.urdf', 'clasp': 'clasp.urdf', 'gripper': 'gripper.urdf', 'grasp': 'grasp.urdf', 'crank': 'crank.urdf'}
only = sys.argv[1:] or list(MODELS)
dtype = torch.float64 if os.environ.get('DPLL_DTYPE') == 'f64' else torch.float32
grid = [dict(n_stages=n, stage_factor=f, stage_max_iter=m) for n, f, m in itertools.product((1, 3, 4, 6, 8), (2.0, 3.0, 5.0), (2, 3))]
grid = [{}] + [c for c in grid if not (c['n_stages'] == 1 and (c['stage_factor'] != 2.0 or c['stage_max_iter'] != 2))]
for name in only:
    g = np.load(os.path.join(REPO, 'tests', 'golden', name + '_literal.npz'))
    reps = -(-4096 // g['x'].shape[0])
    x = torch.tensor(np.tile(g['x'], (reps, 1))[:4096], dtype=dtype, device='cuda:0')
    xp = torch.tensor(np.tile(g['x_plus'], (reps, 1))[:4096], dtype=dtype, device='cuda:0')
    ref = np.tile(g['loss'], reps)[:4096]
    rows = []
    for cfg in grid:
        s = MultibodyLearnableSystem({name: os.path.join(REPO, 'assets', MODELS[name])}, float(g['dt']), dtype=dtype, device='cuda:0')
        s.set_solver(**cfg)
        loss, _, iters = s.contact_forces(x, xp)
        err = np.abs(loss.cpu().double().numpy() - ref).max()
        for _ in range(3):
            s.contactnets_loss_and_grad(x, xp)
        best = float('inf')
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                s.contactnets_loss_and_grad(x, xp)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 20)
        it = iters.cpu().numpy()
        rows.append((best, cfg, err, int(it.max()), float(it.mean())))
    base = rows[0][0]
    print(f'== {name} {dtype}: default {base:.1f} us, iters max {rows[0][3]} mean {rows[0][4]:.2f}, err {rows[0][2]:.1e}', flush=True)
    for best, cfg, err, imax, imean in sorted(rows[1:], key=lambda r: r[0])[:6]:
        print(f'   {best:7.1f} us ({best / base:.3f})  {cfg}  iters max {imax} mean {imean:.2f}  err {err:.1e}', flush=True)
