"""Diagnostic: loss-kernel time / iterations / error for a few solver settings given on the command line as
python dict literals, e.g.  python tools/diag/quick_opts.py "{}" "{'ls_tol': 1e30, 'stage_ls_tol': 1e30}"
Optional env: DPLL_CASE (fixture name, default cube_box_4096), DPLL_DTYPE (f32|f64)."""
import ast, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
case = os.environ.get('DPLL_CASE', 'cube_box_4096')
dtype = torch.float64 if os.environ.get('DPLL_DTYPE', 'f32') == 'f64' else torch.float32
g = np.load(os.path.join(REPO, 'tests', 'golden', case + '.npz'))
urdf = {'contactnets_cube.urdf': 'cube.urdf', 'contactnets_elbow.urdf': 'elbow.urdf'}[str(g['urdf'])]
x = torch.tensor(g['x'], dtype=dtype, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
for text in sys.argv[1:] or ['{}']:
    cfg = ast.literal_eval(text)
    s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', urdf)}, float(g['dt']), dtype=dtype, device='cuda:0')
    s.set_solver(**cfg)
    loss, _, iters = s.contact_forces(x, xp)
    err = np.abs(loss.cpu().double().numpy() - g['loss']).max()
    ts = [s.profile_loss_kernels(x, xp, reps=100) for _ in range(5)]
    t = min(a for a, _ in ts) * 1e3
    tf = min(b for _, b in ts) * 1e3
    ipw = 64 // (4 * (s.spec.n_joints + 1))
    it = iters.cpu().numpy()
    waves = it[:(len(it) // ipw) * ipw].reshape(-1, ipw).max(-1)
    print(f'{cfg}: loss kernel {t:.2f} us, finalize {tf:.2f} us, err {err:.1e}, iters max {it.max()} mean {it.mean():.2f} wave-mean {waves.mean():.2f}', flush=True)
