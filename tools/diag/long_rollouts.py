"""Diagnostic: long fused rollouts of every model family from the fixtures' states (repeated to a batch): non-finite states, and
whether bodies stay above the ground.   python tools/diag/long_rollouts.py [steps]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
from dair_pll_amd import MultibodyLearnableSystem
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
GOLD = os.path.join(REPO, 'tests', 'golden'); ASSETS = os.path.join(REPO, 'assets')
SOURCES = {'polycube': ('cube_mesh.urdf', 'polygon'), 'wedge': ('wedge.urdf', 'polygon'), 'clasp_ball': ('clasp_ball.urdf', 'polygon')}
general = ['chain3', 'vee', 'ballcube', 'mace', 'polycube', 'wedge', 'clasp', 'clasp_ball', 'vee_pair', 'gripper', 'crank', 'pincer', 'grasp', 'slider']
cases = [(m, SOURCES.get(m, (m + '.urdf', 'deep_support')), m + '_literal') for m in general]
cases += [('cube_mesh', ('cube_mesh.urdf', 'deep_support'), 'cube_mesh_literal'), ('elbow_mesh', ('elbow_mesh.urdf', 'deep_support'), 'elbow_mesh_literal'),
          ('clasp_mesh', ('clasp_mesh.urdf', 'deep_support'), 'clasp_mesh_literal'), ('elbow', ('elbow.urdf', 'deep_support'), 'elbow_box_4096')]
for name, (urdf, rep), fixture in cases:
    g = np.load(os.path.join(GOLD, fixture + '.npz'))
    for dtype in (torch.float32, torch.float64):
        try:
            s = MultibodyLearnableSystem({'m': os.path.join(ASSETS, urdf)}, float(g['dt']), dtype=dtype, device='cuda:0', mesh_representation=rep)
            x = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
            reps = max(1, 512 // x.shape[0])
            x0 = x.repeat(reps, 1)[:512].unsqueeze(-2)
            carry = torch.zeros((x0.shape[0], 1), device='cuda:0')
            with torch.no_grad():
                traj, _ = s.simulate(x0, carry, steps)
            bad = ~torch.isfinite(traj).all(-1)
            n_bad = int(bad.any(-1).sum())
            first = int(bad.float().argmax(-1)[bad.any(-1)].min()) if n_bad else -1
            big = float(traj[torch.isfinite(traj)].abs().max())
            print(f'{name:12s} {str(dtype):14s} batch {x0.shape[0]} steps {steps}: non-finite trajectories {n_bad} (first at step {first}), largest |state| {big:.3g}', flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f'{name:12s} {dtype}: {type(exc).__name__}: {str(exc)[:200]}', flush=True)
