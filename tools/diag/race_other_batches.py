"""Diagnostic: the racing copies on OTHER 4096-pair samples of the reference's 57,812 cube-toss pairs (the schedules were
picked on the benchmark batch): Newton iterations of the slowest item / mean, with and without copies, kernel time.
  python tools/diag/race_other_batches.py"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
from dair_pll_amd.trainer import load_tosses, slice_pairs
tosses = load_tosses(os.path.join(REPO, 'assets', 'contactnets_cube_tosses.npz'))
x_all, xp_all = slice_pairs(tosses)
dt = float(np.load(os.path.join(REPO, 'assets', 'contactnets_cube_tosses.npz'))['dt'])
print('pairs', x_all.shape[0])
for dtype in (torch.float32, torch.float64):
    s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube.urdf')}, dt, dtype=dtype, device='cuda:0')
    for seed in range(6):
        pick = torch.randperm(x_all.shape[0], generator=torch.Generator().manual_seed(seed))[:4096]
        x, xp = x_all[pick].to(dtype).cuda(), xp_all[pick].to(dtype).cuda()
        out = {}
        for copies in (1, 0):
            s.set_solver(portfolio=copies)
            loss, force, iters = s.contact_forces(x, xp)
            t = min(a for a, _ in [s.profile_loss_kernels(x, xp, reps=100) for _ in range(3)])
            out[copies] = (loss.clone(), iters.clone(), t)
        d = (out[0][0] - out[1][0]).abs().max().item()
        print(f'{dtype} sample {seed}: iterations max/mean without copies {out[1][1].max().item()}/{out[1][1].float().mean().item():.2f} '
              f'with {out[0][1].max().item()}/{out[0][1].float().mean().item():.2f}; loss kernel {out[1][2] * 1e3:.2f} -> {out[0][2] * 1e3:.2f} us; '
              f'largest loss difference {d:.1e}', flush=True)
