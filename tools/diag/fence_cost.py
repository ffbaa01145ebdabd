"""Diagnostic: what the host-side fence around a 20-step graph replay costs with different ways of waiting for the GPU.
  python tools/diag/fence_cost.py"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube.urdf')}, float(g['dt']), dtype=torch.float32, device='cuda:0')
x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
for _ in range(3): s.contactnets_loss_and_grad(x, xp)
torch.cuda.synchronize()
graphs = {}
for n in (20, 50):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): s.contactnets_loss_and_grad(x, xp)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): s.contactnets_loss_and_grad(x, xp)
    graphs[n] = gr
stream = torch.cuda.current_stream()
ev = torch.cuda.Event(enable_timing=False, blocking=False)
def wait_device(): torch.cuda.synchronize()
def wait_stream(): stream.synchronize()
def wait_event(): ev.record(stream); ev.synchronize()
def wait_poll():
    ev.record(stream)
    while not ev.query(): pass
for name, wait in (('torch.cuda.synchronize', wait_device), ('stream.synchronize', wait_stream), ('event.synchronize', wait_event), ('event.query spin', wait_poll)):
    for n in (20, 50):
        gr = graphs[n]; ts = []
        for _ in range(300):
            wait(); t0 = time.perf_counter(); gr.replay(); wait(); ts.append(time.perf_counter() - t0)
        print(f'{name:24s} {n} steps per region: {np.median(ts) / n * 1e6:.3f} us per step', flush=True)
