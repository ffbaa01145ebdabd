"""Diagnostic: launch time of the general build's loss (+ gradients) kernel for every general model at B = 4096
(tiled fixture pairs), f32 and f64.  Run on the MI355X: python tools/diag/time_general.py"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem, _capi
if os.environ.get('DPLL_LIB'):  # another build of the library (A/B runs)
    _capi.LIB_PATH = os.path.abspath(os.environ['DPLL_LIB'])
ONLY = sys.argv[1:]
MODELS = {'chain3': ('chain3.urdf', 'deep_support'), 'gripper': ('gripper.urdf', 'deep_support'), 'mace': ('mace.urdf', 'deep_support'), 'wedge': ('wedge.urdf', 'polygon'),
          'clasp': ('clasp.urdf', 'deep_support'), 'clasp_ball': ('clasp_ball.urdf', 'polygon'), 'crank': ('crank.urdf', 'deep_support'),
          'pincer': ('pincer.urdf', 'deep_support'), 'grasp': ('grasp.urdf', 'deep_support'), 'slider': ('slider.urdf', 'deep_support')}
for name, (urdf, rep) in MODELS.items():
    if ONLY and name not in ONLY:
        continue
    g = np.load(os.path.join(REPO, 'tests', 'golden', name + '_literal.npz'))
    for dtype in (torch.float32, torch.float64):
        s = MultibodyLearnableSystem({name: os.path.join(REPO, 'assets', urdf)}, float(g['dt']), dtype=dtype, device='cuda:0',
                                     mesh_representation=rep)
        if os.environ.get('DPLL_SOLVER'):  # e.g. DPLL_SOLVER="{'f64_refine': 0}"
            import ast
            s.set_solver(**ast.literal_eval(os.environ['DPLL_SOLVER']))
        reps = -(-4096 // g['x'].shape[0])
        x = torch.tensor(np.tile(g['x'], (reps, 1))[:4096], dtype=dtype, device='cuda:0')
        xp = torch.tensor(np.tile(g['x_plus'], (reps, 1))[:4096], dtype=dtype, device='cuda:0')
        for _ in range(3):
            s.contactnets_loss_and_grad(x, xp)
        best = float('inf')
        for _ in range(3):  # (the calls are enqueued faster than they run only when the queue is deep enough: 100 per timing)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                s.contactnets_loss_and_grad(x, xp)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 10)
        print(f'{name:10s} {str(dtype):14s} loss+grad at B=4096: {best:.0f} us per call')
