"""Two URDFs in one system (the reference's ``init_urdfs`` with two entries: ``drake_utils.py:309-335`` puts both models into one
plant, the state is the ``ProductSpace`` of their spaces): two cubes that collide with the ground and with each other, on the
forest build (``csrc/dpll_forest.hip``).  Tosses are simulated with the true parameters; a model whose second cube starts 15 %
too large learns its size back from the ContactNets loss.

    python examples/two_cubes.py [--epochs 100] [--fused-adam] [--out DIR]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dair_pll_amd import MultibodyLearnableSystem  # noqa: E402
from dair_pll_amd.trainer import ContactNetsTrainer  # noqa: E402


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--epochs', type=int, default=100)
    parser.add_argument('--fused-adam', action='store_true', help='Adam inside the finalize kernel (dpll_contactnets_train_step)')
    parser.add_argument('--out', default=None, help='directory for the learned URDFs (one per model)')
    args = parser.parse_args()
    cube = os.path.join(REPO, 'assets', 'cube.urdf')
    urdfs = {'cube_a': cube, 'cube_b': cube}
    dt = 0.0068
    truth = MultibodyLearnableSystem(urdfs, dt, dtype=torch.float64, device='cuda:0')
    print('system:', truth.spec.n_bodies, 'bodies,', truth.space.n_q, '+', truth.space.n_v, 'coordinates,', truth.spec.n_contacts,
          'contacts (', len(truth.spec.pairs), 'body-body candidate ), build:', 'forest' if truth.forest else 'register-resident')
    # initial states: the recorded two-cube states of the test fixture (cubes above the ground, moving towards each other)
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'two_cubes_literal.npz'))
    x0 = torch.tensor(g['x'], device='cuda:0')
    with torch.no_grad():
        traj, _ = truth.simulate(x0.unsqueeze(-2), torch.zeros((x0.shape[0], 1), device='cuda:0'), 20)
    n_x = truth.space.n_x
    x, xp = traj[:, :-1].reshape(-1, n_x), traj[:, 1:].reshape(-1, n_x)
    model = MultibodyLearnableSystem(urdfs, dt, dtype=torch.float64, device='cuda:0', output_urdfs_dir=args.out)
    with torch.no_grad():
        model.multibody_terms.contact_terms.geometries[2].length_params.mul_(1.15)
    u = torch.zeros((x.shape[0], 0), device='cuda:0')
    with torch.no_grad():
        first = model.contactnets_loss(x, u, xp).mean().item()
    trainer = ContactNetsTrainer(model, lr=2e-3, batch_size=128, fused_adam=args.fused_adam)
    start = time.time()
    trainer.fit(x, xp, epochs=args.epochs)
    torch.cuda.synchronize()
    elapsed = time.time() - start
    with torch.no_grad():
        last = model.contactnets_loss(x, u, xp).mean().item()
    half = model.multibody_terms.contact_terms.geometries[2].length_params.abs().reshape(-1).tolist()
    print(f'{x.shape[0]} transitions, {args.epochs} epochs in {elapsed:.2f} s: loss {first:.3e} -> {last:.3e}')
    print('half lengths of cube_b:', [round(h, 4) for h in half], '(true 0.0524, started at', round(1.15 * 0.0524, 4), ')')
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        print('written:', model.generate_updated_urdfs())


if __name__ == '__main__':
    main()
