"""Diagnostic: the general build with two learned shapes (assets/clasp_mesh.urdf, the reference's own body-body case) at
B = 4096: loss + every gradient, per call.  Run on the MI355X (under rocprofv3 for the per-kernel table)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'clasp_mesh_literal.npz'))
torch.manual_seed(0)
reps = -(-4096 // g['x'].shape[0])
for dtype in (torch.float32, torch.float64):
    s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'clasp_mesh.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
    s.load_state_dict({name: torch.tensor(g['param/' + name]) for name, _ in s.named_parameters()})
    x = torch.tensor(np.tile(g['x'], (reps, 1))[:4096], dtype=dtype, device='cuda:0')
    xp = torch.tensor(np.tile(g['x_plus'], (reps, 1))[:4096], dtype=dtype, device='cuda:0')
    for _ in range(3):
        s.contactnets_loss_and_grad(x, xp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        s.contactnets_loss_and_grad(x, xp)
    torch.cuda.synchronize()
    print(f'clasp_mesh {dtype}: {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per loss + gradient call at B = 4096', flush=True)
