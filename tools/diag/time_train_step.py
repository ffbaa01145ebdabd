"""Diagnostic: per-call time of the fused training step (loss + gradients + Adam in the finalize kernel) against the plain
loss + gradients call, eagerly and as a replayed hipGraph.  Run on the MI355X: python tools/diag/time_train_step.py"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
from dair_pll_amd.system import FusedAdamState
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
s = MultibodyLearnableSystem({'cube': os.path.join(REPO, 'assets', 'cube.urdf')}, float(g['dt']), device='cuda:0')
adam = FusedAdamState(lr=1e-4)


def timed(fn, n=300):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print('eager loss+grad      %.1f us' % timed(lambda: s.contactnets_loss_and_grad(x, xp)))
print('eager fused step     %.1f us' % timed(lambda: s.contactnets_train_step(x, xp, adam)))
for name, fn in (('loss+grad', lambda: s.contactnets_loss_and_grad(x, xp)), ('fused step', lambda: s.contactnets_train_step(x, xp, adam))):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(20): fn()
    print('graph (20 per replay) %-10s %.2f us per step' % (name, timed(graph.replay, 100) / 20))
print('state', adam.state.tolist(), 'params finite', bool(torch.isfinite(s._packed()).all()))
