"""Diagnostic: the one-probe line search on a wide synthetic distribution (65,536 random cube / elbow states, random
parameters): iteration counts stay far from the cap, forces are cone-feasible, float32 agrees with float64."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
for urdf in ('cube.urdf', 'elbow.urdf'):
    gen = torch.Generator().manual_seed(11)
    systems = {dt: MultibodyLearnableSystem({'m': os.path.join(REPO, 'assets', urdf)}, 0.0068, dtype=dt, device='cuda:0')
               for dt in (torch.float64, torch.float32)}
    s64 = systems[torch.float64]
    with torch.no_grad():
        for p in s64.parameters():
            p.add_(0.2 * torch.randn(p.shape, generator=gen, dtype=torch.float64).to(p.device) * (p.abs() + 0.05))
    systems[torch.float32].load_state_dict(s64.state_dict())
    n_j, batch = s64.spec.n_joints, 65536
    quat = torch.randn((batch, 4), generator=gen, dtype=torch.float64)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    pos = torch.cat((0.3 * torch.randn((batch, 2), generator=gen, dtype=torch.float64), 0.02 + 0.15 * torch.rand((batch, 1), generator=gen, dtype=torch.float64)), -1)
    joints = 1.0 * torch.randn((batch, n_j), generator=gen, dtype=torch.float64)
    vel = torch.cat((6.0 * torch.randn((batch, 3), generator=gen, dtype=torch.float64), 1.5 * torch.randn((batch, 3), generator=gen, dtype=torch.float64),
                     4.0 * torch.randn((batch, n_j), generator=gen, dtype=torch.float64)), -1)
    x = torch.cat((quat, pos, joints, vel), -1).cuda()
    with torch.no_grad():
        xp = s64.step(x)
        xp = xp + 2e-3 * torch.randn(xp.shape, generator=gen, dtype=torch.float64).cuda()
        xp[:, :4] = xp[:, :4] / xp[:, :4].norm(dim=-1, keepdim=True)
        out = {}
        for dt, s in systems.items():
            loss, force, iters = s.contact_forces(x.to(dt), xp.to(dt))
            k = s.spec.n_contacts
            normal, tang = force[:, :k], force[:, k:].reshape(-1, k, 2)
            viol = (tang.norm(dim=-1) - normal).clamp(min=0).max().item()
            out[dt] = loss.double()
            print(urdf, dt, 'iters max', iters.max().item(), 'mean %.2f' % iters.float().mean().item(), 'cone violation %.1e' % viol,
                  'loss max %.3e' % loss.max().item(), 'finite', bool(torch.isfinite(loss).all()))
        steps = systems[torch.float32].step(x.float())
        print(urdf, 'float32 loss vs float64: max abs diff %.2e (rel to max loss %.1e)' % ((out[torch.float32] - out[torch.float64]).abs().max().item(),
              ((out[torch.float32] - out[torch.float64]).abs().max() / out[torch.float64].abs().max()).item()),
              ' step f32 vs f64 max diff %.2e' % (steps.double() - s64.step(x)).abs().max().item())
