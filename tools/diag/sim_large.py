import os, sys, time, numpy as np, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem
g = np.load(os.path.join(REPO, 'tests', 'golden', 'cube_box_4096.npz'))
s = MultibodyLearnableSystem({'cube': os.path.join(REPO, 'assets', 'cube.urdf')}, float(g['dt']), dtype=torch.float32, device='cuda:0')
x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0')
for B in (4096, 65536):
    xb = x[torch.randint(0, 4096, (B,), device='cuda:0')]
    with torch.no_grad():
        for steps in (40,):
            s.simulate(xb.unsqueeze(-2), torch.zeros((B, 1), device='cuda:0'), steps); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): s.simulate(xb.unsqueeze(-2), torch.zeros((B, 1), device='cuda:0'), steps)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print('B', B, 'steps', steps, 'ms %.2f' % (dt * 1e3), 'traj-steps/s %.3e' % (B * steps / dt))
