"""Diagnostic only: per-wave shader-clock stamps of the loss kernel.  Needs the -DDPLL_STAMPS build:

    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DDPLL_STAMPS -shared \
        -o tools/diag/libdpll_hip_stamps.so dair_pll_amd/csrc/dpll_kernels.hip

Prints where the slowest, the median and the fastest wave spend their time."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import _capi
_capi.LIB_PATH = os.path.join(REPO, 'tools', 'diag', os.environ.get('DPLL_DIAG_LIB', 'libdpll_hip_stamps.so'))
from dair_pll_amd import MultibodyLearnableSystem
dtype = torch.float64 if 'f64' in sys.argv else torch.float32
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
s = MultibodyLearnableSystem({'cube': os.path.join(REPO, 'assets', 'cube.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
x = torch.tensor(g['x'], dtype=dtype, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
s.set_solver(portfolio=int(os.environ.get('DPLL_PORTFOLIO', '0')))
for _ in range(5): s.contactnets_loss_and_grad(x, xp)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(50): s.contactnets_loss_and_grad(x, xp)
e1.record(); torch.cuda.synchronize(); print('eager us per call', e0.elapsed_time(e1) * 20)
torch.cuda.synchronize()
lib = _capi.library()
NR = int(os.environ.get('DPLL_DIAG_ROWS', '256')); out = np.zeros((2048, 8), dtype=np.uint64)
lib.dpll_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.dpll_debug_read_stamps(out.ctypes.data_as(ctypes.c_void_p), 2048) == 0
phases = out[1024:1024 + NR].astype(np.int64)
t = out[:NR].astype(np.int64)
names = ['params(0->1)', 'loads+terms+contacts(1->4)', 'newton(4->5)', 'adjoint(5->2)', 'reduce+store(2->3)']
seg = np.stack([t[:, 1] - t[:, 0], t[:, 4] - t[:, 1], t[:, 5] - t[:, 4], t[:, 2] - t[:, 5], t[:, 3] - t[:, 2]], 1)
total = t[:, 3] - t[:, 0]
its = t[:, 6]
order = np.argsort(total)
print('stamp units: s_memtime ticks')
for label, idx in (('slowest', order[-1]), ('median', order[len(order) // 2]), ('fastest', order[0])):
    print(label, 'wave', idx, 'total', total[idx], 'max newton iters in wave', its[idx], dict(zip(names, seg[idx])))
print('mean per segment', dict(zip(names, seg.mean(0).round(0))), 'mean total', total.mean())
print('newton ticks per iteration (slowest wave)', seg[order[-1], 2] / max(1, its[order[-1]]))
print('launch skew: last start', t[:, 0].max() - t[:, 0].min(), 'last end', t[:, 3].max() - t[:, 0].min(), 'start quantiles', np.quantile(t[:, 0] - t[:, 0].min(), [0.25, 0.5, 0.75, 0.9, 1.0]))

pn = ['grad+hessian+reduce', 'cholesky+solve', 'decrement+tests', 'state at y+d', 'fallback search', 'update+stage-logic']
w = order[-1]
print('phase cycles, slowest wave (sum over its iterations):', dict(zip(pn, phases[w, :6])), 'per iteration:', dict(zip(pn, (phases[w, :6] / max(1, its[w])).round(0))))
print('phase cycles, mean over waves per iteration:', dict(zip(pn, (phases[:, :6].sum(0) / max(1, its.sum())).round(0))))
print('line-search probes per iteration: slowest wave', phases[w, 6] / max(1, its[w]), 'mean', phases[:, 6].sum() / max(1, its.sum()))
ev = out[1024:1024 + NR, 7].astype(np.int64)
n_ev, first, shortest = ev & 0xff, (ev >> 8) & 0xffffff, (ev >> 32) & 0xffffff
has = n_ev > 0
print('fallback events per wave: mean', n_ev.mean(), 'max', n_ev.max(), '| cycles of the first event: mean', first[has].mean().round(0),
      '| shortest event of a wave (waves with >= 3 events): mean', shortest[n_ev >= 3].mean().round(0) if (n_ev >= 3).any() else None,
      '| fallback cycles per event (all waves)', phases[:, 4].sum() / max(1, n_ev.sum()))
