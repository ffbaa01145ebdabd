"""Step time of the forest build (loss + every gradient, 4096 items; rollout step) per system and dtype.
DPLL_HIP_LIBRARY selects another build of the library for A/B runs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from dair_pll_amd import MultibodyLearnableSystem  # noqa: E402

SYSTEMS = {'cube': ({'cube': 'cube.urdf'}, 'cube_box_literal'), 'two_cubes': ({'cube_a': 'cube.urdf', 'cube_b': 'cube.urdf'}, 'two_cubes_literal'),
           'rake': ({'rake': 'rake.urdf'}, 'rake_literal'), 'gripper': ({'gripper': 'gripper.urdf'}, 'gripper_literal'),
           'pendulum_cube': ({'pendulum': 'pendulum.urdf', 'cube': 'cube.urdf'}, 'pendulum_cube_literal'), 'chain6': ({'chain6': 'chain6.urdf'}, 'chain6_literal')}
batch = int(os.environ.get('BATCH', '4096'))
print('library', os.environ.get('DPLL_HIP_LIBRARY', 'default'), 'batch', batch)
for name in (sys.argv[1:] or SYSTEMS):
    urdfs, fixture = SYSTEMS[name]
    g = np.load(os.path.join('tests', 'golden', fixture + '.npz'))
    line = f'{name:14s}'
    for dtype in (torch.float32, torch.float64):
        system = MultibodyLearnableSystem({k: os.path.join('assets', v) for k, v in urdfs.items()}, float(g['dt']), dtype=dtype, device='cuda:0', build='forest')
        pick = torch.randint(0, g['x'].shape[0], (batch,), generator=torch.Generator().manual_seed(0))
        x = torch.tensor(g['x'], dtype=dtype)[pick].cuda()
        xp = torch.tensor(g['x_plus'], dtype=dtype)[pick].cuda()
        iters = torch.zeros(batch, dtype=torch.int32, device='cuda:0')
        system._launch_loss(x, xp, None, 1.0, False, iters=iters)
        for _ in range(3):
            system.contactnets_loss_and_grad(x, xp)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        loss_us = float('inf')
        for _ in range(3):  # (best of three timings of ten calls, after the three warm-up calls above: round 4's table carried a
            torch.cuda.synchronize()  # 7.1 ms outlier for chain6 f32 from a single timing)
            start.record()
            for _ in range(10):
                system.contactnets_loss_and_grad(x, xp)
            end.record()
            torch.cuda.synchronize()
            loss_us = min(loss_us, start.elapsed_time(end) / 10 * 1e3)
        with torch.no_grad():
            system.simulate(x.unsqueeze(-2), torch.zeros((batch, 1), device='cuda:0'), 8)
            torch.cuda.synchronize()
            start.record()
            system.simulate(x.unsqueeze(-2), torch.zeros((batch, 1), device='cuda:0'), 8)
            end.record()
            torch.cuda.synchronize()
        sim_us = start.elapsed_time(end) / 8 * 1e3
        line += f' | {"f32" if dtype == torch.float32 else "f64"} loss+grad {loss_us:8.1f} us  rollout {sim_us:8.1f} us/step  iters mean {iters.float().mean().item():.1f} max {iters.max().item()}'
    print(line, flush=True)
