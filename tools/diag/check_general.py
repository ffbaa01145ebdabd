"""Diagnostic: per-item loss and next state of every general model, f32 and f64, against the reference-run fixtures, for the
library given in DPLL_LIB (default: the built one) -- the quick check after changing compiler options of the general
translation unit (csrc/Makefile GENERAL_EXTRA).  Run on the MI355X: python tools/diag/check_general.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from dair_pll_amd import MultibodyLearnableSystem, _capi
if os.environ.get('DPLL_LIB'):
    _capi.LIB_PATH = os.path.abspath(os.environ['DPLL_LIB'])
    if os.environ.get('DPLL_BISECT_ABI'):  # bisect builds of an older source whose struct layouts are the current ones (ABI 17 -> 18 added a geometry kind only)
        import ctypes
        _capi.ABI_VERSION = ctypes.CDLL(_capi.LIB_PATH).dpll_abi_version()
names = os.environ.get('DPLL_MODELS', 'chain3 vee gripper mace crank slider ballcube grasp clasp pincer').split()
for name in names:
    g = np.load(f'tests/golden/{name}_literal.npz')
    for dtype in ((torch.float64,) if os.environ.get('DPLL_F64_ONLY') else (torch.float32, torch.float64)):
        s = MultibodyLearnableSystem({name: f'assets/{name}.urdf'}, float(g['dt']), dtype=dtype, device='cuda:0')
        s.load_state_dict({k: torch.tensor(g['param/' + k]) for k, _ in s.named_parameters()})
        x = torch.tensor(g['x'], dtype=dtype, device='cuda:0'); xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')
        with torch.no_grad():
            loss = s.contactnets_loss(x, torch.zeros((x.shape[0], 0), device='cuda:0'), xp).cpu().double().numpy()
            xn = s.step(x).cpu().double().numpy()
        e1 = np.abs(loss - g['loss']); e2 = np.abs(xn - g['dynamics/x_next'])
        print(f'{name:8s} {str(dtype):14s} loss err {np.nanmax(e1):.2e} bad {(e1 > 1e-4).sum()} nan {np.isnan(loss).sum()} | step err {np.nanmax(e2):.2e} bad {(e2.max(1) > 1e-3).sum()} nan {np.isnan(xn).sum()}')
