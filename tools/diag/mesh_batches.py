"""Diagnostic: mesh-pipeline kernel times vs batch size (what is fixed cost, what is per row tile)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import _capi
if len(sys.argv) > 1: _capi.LIB_PATH = os.path.abspath(sys.argv[1])
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
torch.manual_seed(0)
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube_mesh.urdf')}, float(g['dt']), dtype=torch.float32, device='cuda:0')
x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
for B in ((8, 2048, 4096, 8192, 16384, 65536) if len(sys.argv) < 3 else [int(b) for b in sys.argv[2:]]):
    pick = torch.arange(B, device='cuda:0') % 4096
    xb, xpb = x[pick].contiguous(), xp[pick].contiguous()
    best = None
    for _ in range(3):
        ms = s.profile_mesh_kernels(xb, xpb, reps=30)
        best = ms if best is None else {k: min(best[k], v) for k, v in ms.items()}
    print(B, 'tiles/block', max(1, 4 * B // 32 // 256), {k: round(v * 1e3, 1) for k, v in best.items()}, flush=True)
