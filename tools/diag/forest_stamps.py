"""Shader-clock ticks per phase of the forest build's loss program (diagnostic build -DDPLL_FOREST_STAMPS, workgroup 0).
    make the library:  hipcc ... -DDPLL_FOREST_STAMPS -c dpll_forest.hip ; link as tools/diag/libdpll_hip_fstamps.so
    DPLL_HIP_LIBRARY=tools/diag/libdpll_hip_fstamps.so python tools/diag/forest_stamps.py [system ...]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from dair_pll_amd import MultibodyLearnableSystem, _capi  # noqa: E402

NAMES = {1: 'kinematics', 2: 'mass matrix', 3: 'bias forces', 4: 'chol(M) + a', 5: 'pair directions', 6: 'contacts', 7: 'cone offsets',
         8: 'solve: loop head / init', 9: 'solve: grad + C + CJ + H', 10: 'solve: cholesky(H)', 11: 'solve: chol_solve', 12: 'solve: decrement + jd',
         13: 'solve: M d', 14: 'solve: advance', 15: 'solve: slope / search', 16: 'solve: tail', 17: 'loss value', 18: 'adjoint: solves + twists',
         19: 'adjoint: bodies + contacts', 20: 'adjoint: gather'}
SYSTEMS = {'cube': {'cube': 'cube.urdf'}, 'two_cubes': {'cube_a': 'cube.urdf', 'cube_b': 'cube.urdf'}, 'gripper': {'gripper': 'gripper.urdf'},
           'chain6': {'chain6': 'chain6.urdf'}}
FIX = {'cube': 'cube_box_literal', 'two_cubes': 'two_cubes_literal', 'gripper': 'gripper_literal', 'chain6': 'chain6_literal'}
lib = _capi.library()
for name in (sys.argv[1:] or ['two_cubes', 'chain6']):
    g = np.load(os.path.join('tests', 'golden', FIX[name] + '.npz'))
    system = MultibodyLearnableSystem({k: os.path.join('assets', v) for k, v in SYSTEMS[name].items()}, float(g['dt']), dtype=torch.float32, device='cuda:0',
                                      build='forest')
    pick = torch.randint(0, g['x'].shape[0], (4096,), generator=torch.Generator().manual_seed(0))
    x, xp = torch.tensor(g['x'], dtype=torch.float32)[pick].cuda(), torch.tensor(g['x_plus'], dtype=torch.float32)[pick].cuda()
    system.contactnets_loss_and_grad(x, xp)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    lib.dpll_debug_forest_stamps(out, 1)
    system.contactnets_loss_and_grad(x, xp)
    torch.cuda.synchronize()
    lib.dpll_debug_forest_stamps(out, 1)
    ticks = np.array(out).reshape(32, 2)
    total = ticks[:, 0].sum()
    print(f'{name}: workgroup 0, {ticks[7, 1]} items, {total} ticks in total ({total / max(1, ticks[7, 1]):.0f} per item)')
    for slot, label in NAMES.items():
        if ticks[slot, 1]:
            print(f'  {label:28s} {ticks[slot, 0]:9d} ticks  {100 * ticks[slot, 0] / total:5.1f} %  ({ticks[slot, 1]} times, {ticks[slot, 0] / ticks[slot, 1]:.0f} each)')
