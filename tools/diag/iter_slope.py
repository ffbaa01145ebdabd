"""Diagnostic: loss kernel time against the iteration cap, without and with racing copies: slope = cost of one Newton
iteration of the slowest wave, intercept = everything else (launch, prologue, terms, adjoint, reduction).
  python tools/diag/iter_slope.py"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube.urdf')}, float(g['dt']), dtype=torch.float32, device='cuda:0')
x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
for copies in (1, 4):
    row = []
    for cap in (1, 2, 4, 6, 8, 10, 11, 12, 14, 20):
        s.set_solver(portfolio=copies, max_iter=cap)
        t = min(a for a, _ in [s.profile_loss_kernels(x, xp, reps=100) for _ in range(3)])
        row.append(f'{cap}: {t * 1e3:.2f}')
    print(f'copies {copies}: us at max_iter ' + ', '.join(row), flush=True)
# the same wave count with and without copies: 1024 items x 4 copies = 256 waves = 4096 items without
for copies, n in ((1, 4096), (4, 1024), (4, 2048), (4, 4096), (1, 16384)):
    xb, xpb = x.repeat(4, 1)[:n], xp.repeat(4, 1)[:n]
    s.set_solver(portfolio=copies, max_iter=1)
    t = min(a for a, _ in [s.profile_loss_kernels(xb, xpb, reps=100) for _ in range(3)])
    print(f'max_iter 1, copies {copies}, {n} items ({n * 4 * copies // 64} waves): {t * 1e3:.2f} us', flush=True)
