"""Diagnostic: per-kernel times of the mesh pipeline for the forms of the ICNN GEMM kernels (dpll_solver_opts_t.mesh_gemm:
0 = one wave per SIMD, pipelined; 1 = the 8-wave kernels of rounds 1-4; 2 = bf16 planes) on one device, alternating."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
forms = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
batch = int(os.environ.get('BATCH', '4096'))
reps = batch // 4096
x = torch.tensor(np.tile(g['x'], (reps, 1)), dtype=torch.float32, device='cuda:0'); xp = torch.tensor(np.tile(g['x_plus'], (reps, 1)), dtype=torch.float32, device='cuda:0')
systems = {}
for f in forms:
    torch.manual_seed(0)
    s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', 'cube_mesh.urdf')}, float(g['dt']), dtype=torch.float32, device='cuda:0')
    s.set_solver(mesh_gemm=f)
    systems[f] = s
ref = None
for f, s in systems.items():
    loss = s.contactnets_loss_and_grad(x, xp)
    flat = torch.cat([p.grad.reshape(-1).double() for p in s.parameters()])
    if ref is None: ref = (loss.double(), flat)
    else:
        print(f'form {f} vs form {forms[0]}: loss diff {float((loss.double() - ref[0]).abs().max()):.3e}, grad diff {float((flat - ref[1]).abs().max()):.3e} of {float(ref[1].abs().max()):.3e}', flush=True)
for rnd in range(2):
    for f, s in systems.items():
        best = None
        for _ in range(3):
            ms = s.profile_mesh_kernels(x, xp, reps=50)
            best = ms if best is None else {k: min(best[k], v) for k, v in ms.items()}
        gem = sum(best[k] for k in ('icnn_fwd1', 'icnn_fwd2', 'icnn_bwd1', 'icnn_bwd2')) * 1e3
        print(f'form {f}', {k: round(v * 1e3, 1) for k, v in best.items()}, f'gemms {gem:.1f} us', flush=True)
for f, s in systems.items():
    for _ in range(3): s.contactnets_loss_and_grad(x, xp)
    torch.cuda.synchronize()
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
    print(f'form {f}: step (graph) {t:.1f} us = {batch / t:.1f} M steps/s', flush=True)
