"""Diagnostic: step-backward parameter gradients of a general model under the library in DPLL_HIP_LIBRARY against the oracle's
autograd, entry by entry (which parameters a poisoned local reaches)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_general_models as tg
name = sys.argv[1] if len(sys.argv) > 1 else 'gripper'
g = dict(np.load(f'tests/golden/{name}_literal.npz'))
system = tg.gpu_system(g, name, torch.float64)
rows = np.linspace(0, g['x'].shape[0] - 1, 24).astype(int)
x_np = g['x'][rows]
w = torch.rand((len(rows), 2, x_np.shape[1]), generator=torch.Generator().manual_seed(5), dtype=torch.float64) - 0.5
oracle = tg.oracle_from(g, name).requires_grad_()
x_ref = torch.tensor(x_np).requires_grad_(True)
(oracle.simulate(x_ref, 2)[:, 1:] * w).sum().backward()
x = torch.tensor(x_np, device='cuda:0').requires_grad_(True)
traj, _ = system.simulate(x.unsqueeze(-2), torch.zeros((len(rows), 1), device='cuda:0'), 2)
(traj[:, 1:] * w.cuda()).sum().backward()
ref_named = oracle.named_parameters()
np.set_printoptions(precision=4, linewidth=220)
for key, param in system.named_parameters():
    ref = ref_named[key].grad.numpy(); mine = param.grad.cpu().numpy()
    err = np.abs(mine - ref).max()
    flag = 'BAD' if err > 1e-7 * max(np.abs(ref).max(), 1e-3) else 'ok '
    print(flag, key, 'err %.3e' % err)
    if flag == 'BAD':
        print('   mine', mine.reshape(-1)); print('   ref ', ref.reshape(-1))
print('x grad err', (x.grad.cpu() - x_ref.grad).abs().max().item())
