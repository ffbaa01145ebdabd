"""Diagnostic: loss + gradients of a mesh system with the workspace pre-filled with NaN -- a kernel that reads a part of the
workspace that the same call did not write shows up as NaN (or as a changed value) in the outputs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from dair_pll_amd import MultibodyLearnableSystem
name = sys.argv[1] if len(sys.argv) > 1 else 'clasp_mesh'
dtype = torch.float64 if 'f64' in sys.argv else torch.float32
g = np.load(f'tests/golden/{name}_literal.npz')
x = torch.tensor(g['x'], dtype=dtype, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
out = []
for fill in (0.0, float('nan'), 1e30):
    torch.manual_seed(0)
    s = MultibodyLearnableSystem({name: f'assets/{name}.urdf'}, float(g['dt']), dtype=dtype, device='cuda:0')
    ws = s._mesh_workspace(x.shape[0], x.device)
    ws.view(torch.float32).fill_(fill)
    s.contactnets_loss_and_grad(x, xp)
    torch.cuda.synchronize()
    flat = torch.cat([p.grad.reshape(-1).double() for p in s.parameters()]).cpu()
    out.append(flat)
    print('fill', fill, 'nan in grad', int(torch.isnan(flat).sum()), 'inf', int(torch.isinf(flat).sum()), 'max diff vs fill 0', float((flat - out[0]).abs().max()))
