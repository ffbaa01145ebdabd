"""Diagnostic: fused rollouts without racing copies at 4096 / 8192 / 16384 trajectories (256 / 512 / 1024 one-wave workgroups, at
most one per SIMD): does a wave get slower when its CU's other SIMDs work?   python tools/diag/sim_waves.py [f32|f64]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
dtype = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == 'f64') else torch.float32
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
s.set_solver(portfolio=1)
x = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
for mult in (1, 2, 4, 8):
    x0 = x.repeat(mult, 1).unsqueeze(-2)   # the same items: every wave's work is a copy of one of the first 256 waves'
    carry = torch.zeros((x0.shape[0], 1), device='cuda:0')
    with torch.no_grad():
        for _ in range(2): s.simulate(x0, carry, 80)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
        for _ in range(5):
            torch.cuda.synchronize(); e0.record(); s.simulate(x0, carry, 80); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f'{dtype} {x0.shape[0]} trajectories ({x0.shape[0] // 16} waves): {np.median(ts) * 1e3 / 80:.2f} us per step', flush=True)
