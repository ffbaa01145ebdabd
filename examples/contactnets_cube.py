"""Learn the cube's shape and friction from toss data with the ContactNets loss -- the small-scale counterpart of
``dair_pll/examples/contactnets_simple.py`` (cube system, Adam lr 1e-3, ContactNets loss) on the MI355X kernels.

    python examples/contactnets_cube.py [--epochs 10] [--batch 4096] [--graph] [--fused-adam] [--out /tmp/learned_urdfs]

Data: the reference's whole cube-toss data set (``assets/contactnets_cube_tosses.npz``: the 550 real tosses of
``assets/contactnets_cube/*.pt``, 57,812 (x, x+) pairs by the reference's slice rule), resident on the device.  The model
starts from box lengths that are 25 % too long and a friction coefficient of 0.6 (true: 0.1048 m, 0.15) and is trained
on all pairs in shuffled batches; the learned parameters are printed with the
reference's scalar names and, with ``--out``, written back as a URDF.
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
from dair_pll_amd.trainer import ContactNetsTrainer, load_tosses, slice_pairs  # noqa: E402


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--epochs', type=int, default=10)
    parser.add_argument('--batch', type=int, default=4096)
    parser.add_argument('--graph', action='store_true', help='replay the training step as a hipGraph')
    parser.add_argument('--fused-adam', action='store_true', help='Adam inside the finalize kernel (two launches per step)')
    parser.add_argument('--out', default=None, help='directory for the learned URDF')
    args = parser.parse_args()
    path = os.path.join(REPO, 'assets', 'contactnets_cube_tosses.npz')
    pairs_x, pairs_xp = slice_pairs(load_tosses(path))  # dataset_management.py:43-59
    data = {'dt': float(np.load(path)['dt'])}
    x = pairs_x.to(device='cuda:0', dtype=torch.float32)
    x_plus = pairs_xp.to(device='cuda:0', dtype=torch.float32)
    print(f'{x.shape[0]} pairs of {len(np.load(path)["lengths"])} tosses')
    system = MultibodyLearnableSystem({'cube': os.path.join(REPO, 'assets', 'cube.urdf')}, float(data['dt']),
                                      output_urdfs_dir=args.out, dtype=torch.float32, device='cuda:0')
    with torch.no_grad():
        system.multibody_terms.contact_terms.geometries[1].length_params.mul_(1.25)
        system.multibody_terms.contact_terms.friction_params[1] = 0.6
    trainer = ContactNetsTrainer(system, lr=1e-3, batch_size=args.batch, use_graph=args.graph, fused_adam=args.fused_adam)
    start = time.perf_counter()
    for epoch in range(args.epochs):
        loss = trainer.train_epoch(x, x_plus)
        if epoch % 2 == 0 or epoch == args.epochs - 1:
            scalars = system.scalars()
            print(f'epoch {epoch:3d}  loss {loss:.3e}  len_x {scalars["body_len_x"]:.4f}  mu {scalars["body_mu"]:.3f}')
    torch.cuda.synchronize()
    steps = args.epochs * -(-x.shape[0] // args.batch)
    print(f'{steps} optimizer steps in {time.perf_counter() - start:.2f} s')
    if args.out:
        print('wrote', system.generate_updated_urdfs())


if __name__ == '__main__':
    main()
