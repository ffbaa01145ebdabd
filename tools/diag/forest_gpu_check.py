"""The forest build on the MI355X against the reference-run fixtures (through MultibodyLearnableSystem and the C ABI): loss, every
gradient, next state, terms, the step's backward against the host build, float64 and float32; timing of 4096 items."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, 'tests')
sys.path.insert(0, '.')
from dair_pll_amd import MultibodyLearnableSystem  # noqa: E402

P = 'multibody_terms.'
SOURCES = {'polycube': ('cube_mesh.urdf', 'polygon'), 'wedge': ('wedge.urdf', 'polygon'), 'clasp_ball': ('clasp_ball.urdf', 'polygon'),
           'cube_box': ('cube.urdf', 'deep_support'), 'elbow_box': ('elbow.urdf', 'deep_support')}
FOREST = {'chain6': {'chain6': 'chain6.urdf'}, 'rake': {'rake': 'rake.urdf'}, 'two_cubes': {'cube_a': 'cube.urdf', 'cube_b': 'cube.urdf'},
          'pendulum_cube': {'pendulum': 'pendulum.urdf', 'cube': 'cube.urdf'}}
MODELS = list(FOREST) + ['cube_box', 'elbow_box', 'chain3', 'vee', 'ballcube', 'mace', 'polycube', 'wedge', 'clasp', 'clasp_ball', 'vee_pair', 'gripper',
                         'crank', 'pincer', 'grasp', 'slider']

for name in (sys.argv[1:] or MODELS):
    urdf, representation = SOURCES.get(name, (name + '.urdf', 'deep_support'))
    g = np.load(os.path.join('tests', 'golden', name + '_literal.npz'))
    urdfs = {key: os.path.join('assets', value) for key, value in FOREST.get(name, {name: urdf}).items()}
    line = f'{name:13s}'
    for dtype in (torch.float64, torch.float32):
        system = MultibodyLearnableSystem(urdfs, float(g['dt']), dtype=dtype, device='cuda:0', mesh_representation=representation, build='forest')
        system.load_state_dict({k: torch.tensor(g['param/' + k]) for k, _ in system.named_parameters()})
        x, xp = (torch.tensor(g[k], dtype=dtype, device='cuda:0') for k in ('x', 'x_plus'))
        u = torch.zeros((x.shape[0], 0), device='cuda:0')
        loss = system.contactnets_loss(x, u, xp)
        loss.mean().backward()
        e_loss = np.abs(loss.detach().cpu().double().numpy() - g['loss']).max()
        e_grad = max(np.abs(p.grad.cpu().double().numpy() - g['grad/' + k]).max() / max(1.0, np.abs(g['grad/' + k]).max()) for k, p in system.named_parameters())
        with torch.no_grad():
            nxt = system.step(x)
        e_step = np.abs(nxt.cpu().double().numpy() - g['dynamics/x_next']).max() / max(1.0, np.abs(g['dynamics/x_next']).max())
        rows = g['simulate/rows']
        with torch.no_grad():
            traj, _ = system.simulate(x[rows].unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), int(g['simulate/steps']))
        e_sim = np.abs(traj.cpu().double().numpy() - g['simulate/traj']).max()
        q, v = system.space.q_v(xp)
        D, M, J, phi, a = system.multibody_terms(q, v, u)
        e_M = np.abs(M.cpu().double().numpy() - g['terms/M']).max()
        e_phi = np.abs(np.sort(phi.cpu().double().numpy(), -1) - np.sort(g['terms/phi'], -1)).max()
        e_D = np.abs(np.sort(np.diagonal(D.cpu().double().numpy(), axis1=-2, axis2=-1), -1) - np.sort(np.diagonal(g['terms/D'], axis1=-2, axis2=-1), -1)).max()
        # timing: 4096 items
        pick = torch.randint(0, x.shape[0], (4096,), generator=torch.Generator().manual_seed(0)).cuda()
        xb, xpb = x[pick].contiguous(), xp[pick].contiguous()
        system.contactnets_loss_and_grad(xb, xpb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            system.contactnets_loss_and_grad(xb, xpb)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 5 * 1e6
        line += f' | {"f64" if dtype == torch.float64 else "f32"} loss {e_loss:.1e} grad {e_grad:.1e} step {e_step:.1e} sim {e_sim:.1e} M {e_M:.1e} phi {e_phi:.1e} D {e_D:.1e} {us:7.0f} us/4096'
        if dtype == torch.float64:  # step backward against the host build
            from hostsim import forest
            from dair_pll_amd import _capi
            xg = x[:16].clone().requires_grad_(True)
            system.zero_grad()
            w = torch.randn((16, x.shape[1]), generator=torch.Generator().manual_seed(1), dtype=dtype).cuda()
            (system.step(xg) * w).sum().backward()
            flat = system._packed().detach().cpu().numpy()
            nb, ng = system.spec.n_bodies, len(system.spec.geoms())
            th, fr, le = flat[:10 * nb].reshape(nb, 10), flat[10 * nb:10 * nb + 1 + ng], flat[10 * nb + 1 + ng:].reshape(ng, 24)
            gh, xh = forest.step_backward(system._desc, th, fr, le, x[:16].cpu().numpy(), w.cpu().numpy(), want_state=True)
            gd = torch.cat([p.grad.reshape(-1) for p in system._param_list()]).cpu().numpy()
            layout = np.concatenate([gh[off:off + p.numel()] for p, off in system._layout()[0]])
            line += f' bwd p {np.abs(gd - layout).max() / max(1.0, np.abs(layout).max()):.1e} x {np.abs(xg.grad.cpu().numpy() - xh).max() / max(1.0, np.abs(xh).max()):.1e}'
    print(line, flush=True)
