"""Diagnostic: loss-kernel time vs solver settings on the headline batch (one device, interleaved)."""
import os, sys, itertools
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
s = MultibodyLearnableSystem({'cube': os.path.join(REPO, 'assets', 'cube.urdf')}, float(g['dt']), dtype=torch.float32, device='cuda:0')
x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
ref = g['loss']
configs = [dict(n_stages=1), dict(n_stages=6)]
for ns, fac, smi, mls, smls in itertools.product((4, 5, 6, 7), (2.0, 3.0, 5.0), (1, 2, 3), (1, 2), (1, 2)):
    configs.append(dict(n_stages=ns, stage_factor=fac, stage_max_iter=smi, ls_tol=0.9, stage_ls_tol=0.9, max_ls=mls, stage_max_ls=smls))
res = []
for cfg in configs:
    s.set_solver(**cfg)
    loss, _, iters = s.contact_forces(x, xp)
    err = np.abs(loss.cpu().double().numpy() - ref).max()
    t = min(s.profile_loss_kernels(x, xp, reps=50)[0] for _ in range(3)) * 1e3
    res.append((t, cfg, err, iters.max().item(), iters.float().mean().item()))
for t, cfg, err, mx, mean in sorted(res, key=lambda r: r[0])[:25] + res[:2]:
    print('%.1f us' % t, cfg, 'err %.1e' % err, 'iters max', mx, 'mean %.2f' % mean)
