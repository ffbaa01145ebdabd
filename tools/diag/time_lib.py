"""Diagnostic: kernel / step timings of ONE build of the library (path in argv[1]) on the headline batch; used by
tools/diag/ab_libs.sh to compare builds on the same device in alternation.
  python tools/diag/time_lib.py <lib.so> [case] [f32|f64]"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import _capi
_capi.LIB_PATH = os.path.abspath(sys.argv[1])
from dair_pll_amd import MultibodyLearnableSystem
case = sys.argv[2] if len(sys.argv) > 2 else 'cube_box_4096'
dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == 'f64') else torch.float32
g = np.load(os.path.join(REPO, 'tests', 'golden', case + '.npz'))
urdf = {'contactnets_cube.urdf': 'cube.urdf', 'contactnets_elbow.urdf': 'elbow.urdf'}[str(g['urdf'])]
x = torch.tensor(g['x'], dtype=dtype, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', urdf)}, float(g['dt']), dtype=dtype, device='cuda:0')
loss, _, iters = s.contact_forces(x, xp)
err = np.abs(loss.cpu().double().numpy() - g['loss']).max()
ts = [s.profile_loss_kernels(x, xp, reps=200) for _ in range(5)]
for _ in range(3): s.contactnets_loss_and_grad(x, xp)
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): s.contactnets_loss_and_grad(x, xp)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(50): s.contactnets_loss_and_grad(x, xp)
for _ in range(4): graph.replay()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(20): graph.replay()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
print(f'{os.path.basename(sys.argv[1])}: loss kernel {min(a for a, _ in ts) * 1e3:.2f} us, finalize {min(b for _, b in ts) * 1e3:.2f} us, '
      f'step (graph) {best:.2f} us, err {err:.1e}, iters max {iters.max().item()}', flush=True)
# the same kernel timing again, right after ~0.5 s of back-to-back graph replays (what bench.py's roofline object sees)
import time as _t
t_end = _t.perf_counter() + 0.5
while _t.perf_counter() < t_end:
    graph.replay()
torch.cuda.synchronize()
ts2 = [s.profile_loss_kernels(x, xp, reps=200) for _ in range(3)]
print(f'   after 0.5 s of sustained replays: loss kernel {min(a for a, _ in ts2) * 1e3:.2f} us, finalize {min(b for _, b in ts2) * 1e3:.2f} us', flush=True)
