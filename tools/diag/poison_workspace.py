"""Diagnostic: one dpll_step_backward call on gripper (f64, 24 states), workspace pre-filled with a marker; dumps workspace, grad,
grad_x to gpurun_out/<tag>.npz.  Run once per library (DPLL_HIP_LIBRARY) and compare offline."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_general_models as tg
from dair_pll_amd import _capi
from dair_pll_amd.system import _DTYPES, _ptr
name, tag = sys.argv[1], sys.argv[2]
g = dict(np.load(f'tests/golden/{name}_literal.npz'))
s = tg.gpu_system(g, name, torch.float64)
rows = np.linspace(0, g['x'].shape[0] - 1, 24).astype(int)
x = torch.tensor(g['x'][rows], device='cuda:0')
gx = (torch.rand(x.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64) - 0.5).cuda()
lib = _capi.library()
flat = s._packed()
grad = torch.full((flat.numel(),), 7.0, dtype=torch.float64, device='cuda:0')
grad_x = torch.full_like(x, 7.0)
ws_bytes = lib.dpll_workspace_bytes(s._model(), x.shape[0])
ws = torch.full((ws_bytes // 8,), 12345.0, dtype=torch.float64, device='cuda:0')
params = s._params_struct(flat)
_capi.check(lib.dpll_step_backward(s._model(), _DTYPES[torch.float64], ctypes.byref(params), _ptr(x), x.stride(0), _ptr(gx), gx.stride(0),
                                   x.shape[0], _ptr(grad), _ptr(grad_x), grad_x.stride(0), _ptr(ws), ws_bytes, s._stream()))
torch.cuda.synchronize()
os.makedirs('gpurun_out', exist_ok=True)
np.savez(f'gpurun_out/poison_ws_{tag}.npz', ws=ws.cpu().numpy(), grad=grad.cpu().numpy(), grad_x=grad_x.cpu().numpy(), flat=flat.detach().cpu().numpy())
print(tag, 'ws doubles', ws.numel(), 'grad tail', grad[-8:].cpu().numpy())
