import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
for dtype in (torch.float32, torch.float64):
    x0 = torch.tensor(g['x'], dtype=dtype, device='cuda:0').unsqueeze(-2)
    carry = torch.zeros((4096, 1), device='cuda:0')
    s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
    for copies in (1, 0):
        s.set_solver(portfolio=copies)
        with torch.no_grad():
            traj, _ = s.simulate(x0, carry, 80)
        bad = torch.isnan(traj).any(-1)  # (4096, 81)
        items = bad.any(-1).nonzero().flatten().cpu().numpy()
        first = bad.float().argmax(-1).cpu().numpy()
        print(dtype, 'portfolio', copies, 'NaN trajectories', len(items), 'items', items[:40], 'first NaN step', first[items[:40]])
        if len(items):
            i = int(items[0]); k = int(first[i])
            print(' item', i, 'state before', traj[i, k - 1].cpu().numpy(), '\n x0', traj[i, 0].cpu().numpy())
            # single steps from the state before
            xs = traj[i:i + 1, k - 1:k].clone()
            one, _ = s.simulate(xs, carry[:1], 1)
            print(' single step from it (batch of 1):', one[0, 1].detach().cpu().numpy())
