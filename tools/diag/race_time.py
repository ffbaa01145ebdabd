"""Diagnostic: the loss launch with 1 / 2 / 4 racing copies per item (dpll_solver_opts_t.portfolio) on the headline batch:
kernel times (HIP events), Newton iterations of the slowest item and the mean, error against the fixture.
  python tools/diag/race_time.py [case] [f32|f64]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
case = sys.argv[1] if len(sys.argv) > 1 else 'cube_box_4096'
dtype = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == 'f64') else torch.float32
g = np.load(os.path.join(REPO, 'tests', 'golden', case + '.npz'))
urdf = {'contactnets_cube.urdf': 'cube.urdf', 'contactnets_elbow.urdf': 'elbow.urdf'}[str(g['urdf'])]
x = torch.tensor(g['x'], dtype=dtype, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', urdf)}, float(g['dt']), dtype=dtype, device='cuda:0')
for copies in (1, 2, 4, 0):
    s.set_solver(portfolio=copies)
    loss, _, iters = s.contact_forces(x, xp)
    err = np.abs(loss.cpu().double().numpy() - g['loss']).max()
    ts = [s.profile_loss_kernels(x, xp, reps=200) for _ in range(5)]
    print(f'{case} {dtype} portfolio {copies}: loss kernel {min(a for a, _ in ts) * 1e3:.2f} us, finalize {min(b for _, b in ts) * 1e3:.2f} us, '
          f'err {err:.1e}, iters max {iters.max().item()} mean {iters.float().mean().item():.2f}', flush=True)
