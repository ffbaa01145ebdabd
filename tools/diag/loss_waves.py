"""Diagnostic: the loss launch (no racing copies) at 4096 ... 16384 cube pairs / 4096 ... 8192 elbow pairs -- up to one wave
per SIMD -- and the rollouts at the same sizes: per-launch kernel time.   python tools/diag/loss_waves.py"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
for w in ('cube', 'elbow'):
    g = np.load(os.path.join(REPO, 'tests', 'golden', w + '_box_4096.npz'))
    for dtype in (torch.float32,):
        s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', w + '.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
        s.set_solver(portfolio=1)
        x = torch.tensor(g['x'], dtype=dtype, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
        for mult in (1, 2, 3, 4):
            xb, xpb = x.repeat(mult, 1), xp.repeat(mult, 1)
            t = min(a for a, _ in [s.profile_loss_kernels(xb, xpb, reps=100) for _ in range(3)])
            print(f'{w} {dtype} {xb.shape[0]} pairs: loss kernel {t * 1e3:.2f} us ({xb.shape[0] / t / 1e3 / 1e6:.0f} M pair-launches/s)', flush=True)
