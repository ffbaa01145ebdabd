"""Loss + every gradient of 4096 items, the general build against the forest build, per model of tests/test_general_models.py."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from dair_pll_amd import MultibodyLearnableSystem  # noqa: E402
from test_general_models import MODELS, source  # noqa: E402

batch = 4096
for name in (sys.argv[1:] or MODELS):
    urdf, representation = source(name)
    g = np.load(os.path.join('tests', 'golden', name + '_literal.npz'))
    line = f'{name:11s}'
    for dtype in (torch.float32, torch.float64):
        for build in ('auto', 'forest'):
            system = MultibodyLearnableSystem({name: urdf}, float(g['dt']), dtype=dtype, device='cuda:0', build=build, mesh_representation=representation)
            pick = torch.randint(0, g['x'].shape[0], (batch,), generator=torch.Generator().manual_seed(0))
            x = torch.tensor(g['x'], dtype=dtype)[pick].cuda()
            xp = torch.tensor(g['x_plus'], dtype=dtype)[pick].cuda()
            for _ in range(3):
                system.contactnets_loss_and_grad(x, xp)
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            start.record()
            for _ in range(10):
                system.contactnets_loss_and_grad(x, xp)
            end.record()
            torch.cuda.synchronize()
            loss_us = start.elapsed_time(end) / 10 * 1e3
            with torch.no_grad():
                system.simulate(x.unsqueeze(-2), torch.zeros((batch, 1), device='cuda:0'), 8)
                torch.cuda.synchronize()
                start.record()
                system.simulate(x.unsqueeze(-2), torch.zeros((batch, 1), device='cuda:0'), 8)
                end.record()
                torch.cuda.synchronize()
            sim_us = start.elapsed_time(end) / 8 * 1e3
            line += f' | {"f32" if dtype == torch.float32 else "f64"} {("general" if build == "auto" else build):7s} loss {loss_us:7.1f} sim {sim_us:7.1f}'
    print(line, flush=True)
