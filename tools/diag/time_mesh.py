"""Diagnostic: per-kernel times of the mesh pipeline for ONE build of the library (argv[1]); alternate builds with
tools/diag/ab_mesh.sh on one device."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import _capi
_capi.LIB_PATH = os.path.abspath(sys.argv[1])
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
torch.manual_seed(0)
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube_mesh.urdf')}, float(g['dt']), dtype=torch.float32, device='cuda:0')
x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
best = None
for _ in range(4):
    ms = s.profile_mesh_kernels(x, xp, reps=50)
    best = ms if best is None else {k: min(best[k], v) for k, v in ms.items()}
for _ in range(3): s.contactnets_loss_and_grad(x, xp)
torch.cuda.synchronize()
import time
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): s.contactnets_loss_and_grad(x, xp)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(10): s.contactnets_loss_and_grad(x, xp)
for _ in range(3): graph.replay()
torch.cuda.synchronize()
t = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(10): graph.replay()
    torch.cuda.synchronize()
    t = min(t, (time.perf_counter() - t0) / 100 * 1e6)
gem = sum(best[k] for k in ('icnn_fwd1', 'icnn_fwd2', 'icnn_bwd1', 'icnn_bwd2')) * 1e3
print(os.path.basename(sys.argv[1]), {k: round(v * 1e3, 1) for k, v in best.items()}, f'gemms {gem:.1f} us, step (graph) {t:.1f} us', flush=True)
