"""Diagnostic: fused rollouts (dpll_simulate) with and without racing copies: time per step, iterations, end-state difference.
  python tools/diag/race_sim_time.py [cube|elbow] [f32|f64] [horizon]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
w = sys.argv[1] if len(sys.argv) > 1 else 'cube'
dtype = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == 'f64') else torch.float32
horizon = int(sys.argv[3]) if len(sys.argv) > 3 else 80
g = np.load(os.path.join(REPO, 'tests', 'golden', w + '_box_4096.npz'))
x0 = torch.tensor(g['x'], dtype=dtype, device='cuda:0').unsqueeze(-2)
carry = torch.zeros((4096, 1), device='cuda:0')
s = MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', w + '.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
ref = None
for copies in (1, 2, 4, 0):
    s.set_solver(portfolio=copies)
    with torch.no_grad():
        for _ in range(2): traj, _ = s.simulate(x0, carry, horizon)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); e0.record(); s.simulate(x0, carry, horizon); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    if ref is None: ref = traj.clone()
    print(f'{w} {dtype} portfolio {copies}: {np.median(ts) * 1e3 / horizon:.2f} us per step, end-state diff vs no copies {(traj - ref).abs().max().item():.2e}', flush=True)

# one step from states sampled along the rollout: iteration counts per item under each setting
import ctypes
from dair_pll_amd import _capi
lib = _capi.library()
s.set_solver(portfolio=1)
with torch.no_grad():
    traj, _ = s.simulate(x0, carry, horizon)
print('NaN trajectories', int(torch.isnan(traj).any(-1).any(-1).sum().item()))
flat = s._packed(); params = s._params_struct(flat)
code = _capi.F64 if dtype == torch.float64 else _capi.F32
for step in (0, 10, 40, 70):
    x = traj[:, step].contiguous()
    x = torch.where(torch.isnan(x).any(-1, keepdim=True), traj[:, 0], x)
    for copies in (1, 4):
        s.set_solver(portfolio=copies)
        xn = torch.empty_like(x); iters = torch.zeros(4096, dtype=torch.int32, device='cuda:0')
        args = (s._model(), code, ctypes.byref(params), x.data_ptr(), x.stride(0), 4096, xn.data_ptr(), xn.stride(0), iters.data_ptr(), s._stream())
        _capi.check(lib.dpll_step(*args))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): lib.dpll_step(*args)
        e1.record(); torch.cuda.synchronize()
        per = 16 // copies
        wmax = iters[:4096 // per * per].reshape(-1, per).max(1).values.float()
        print(f'step {step} portfolio {copies}: iters max {iters.max().item()} mean {iters.float().mean().item():.2f} wave-max mean {wmax.mean().item():.2f}; '
              f'{e0.elapsed_time(e1) * 50:.2f} us per launch', flush=True)
