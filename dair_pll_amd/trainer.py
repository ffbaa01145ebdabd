"""Minimal training loop around the fused loss: the caller side of the hot path.

Counterpart of what ``dair_pll/experiment.py:332-363`` (``train_epoch``), ``:213-228`` (Adam) and
``dair_pll/dataset_management.py:43-59`` (``TrajectorySliceDataset`` with the default one-step slices)
do around ``contactnets_loss`` -- only the parts the hot path needs: slicing ``(T, n_x)`` trajectories
into ``(x, x+)`` pairs, shuffled mini-batches resident on the device, one fused forward+backward
launch per batch, ``torch.optim.Adam`` on the parameters (whose ``.grad`` the kernels write in place),
and the single gradient all-reduce when a process group is active.  Everything else of the
reference's experiment machinery (W&B, checkpoints, evaluation, hyper-parameter search) is out of
scope.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Tuple

import torch
from torch import Tensor

from .distributed import GradientAllReduce, shard_bounds


def slice_pairs(trajectories: Iterable[Tensor]) -> Tuple[Tensor, Tensor]:
    """``TrajectorySliceDataset.add_slices_from_trajectory`` with ``t_skip=0, t_history=1,
    t_prediction=1`` (``dataset_management.py:43-59``, ``data_config.py:8-13``): every state but the
    last is an ``x``, its successor the ``x+``."""
    xs, xps = [], []
    for trajectory in trajectories:
        assert trajectory.dim() == 2 and trajectory.shape[0] >= 2
        xs.append(trajectory[:-1])
        xps.append(trajectory[1:])
    return torch.cat(xs), torch.cat(xps)


@dataclass
class TrainLog:
    epoch_losses: List[float]


class ContactNetsTrainer:
    """Adam on ``system.parameters()`` with the fused ContactNets loss
    (``lr=1e-3``, ``weight_decay`` as in ``examples/contactnets_simple.py:78-86``)."""

    def __init__(self, system, lr: float = 1e-3, weight_decay: float = 0.0, batch_size: int = 4096,
                 seed: int = 0) -> None:
        self.system = system
        self.batch_size = batch_size
        self.optimizer = torch.optim.Adam(system.parameters(), lr=lr, weight_decay=weight_decay)
        self.generator = torch.Generator().manual_seed(seed)
        self.reducer: Optional[GradientAllReduce] = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1:
            self.reducer = GradientAllReduce(system)

    def train_epoch(self, x: Tensor, x_plus: Tensor) -> float:
        """One pass over the (device-resident) pairs in shuffled mini-batches; returns the mean of the
        batch losses (``experiment.py:348-363``).  With a process group every rank must hold the SAME
        pairs: each batch is sharded by contiguous rows and gradients are all-reduced."""
        n = x.shape[0]
        order = torch.randperm(n, generator=self.generator).to(x.device)
        losses = []
        for start in range(0, n, self.batch_size):
            idx = order[start:start + self.batch_size]
            if self.reducer is not None:
                lo, hi = shard_bounds(idx.numel(), torch.distributed.get_rank(), self.reducer.world)
                self.system.global_batch = idx.numel()
                idx = idx[lo:hi]
            total = self.system.contactnets_loss_and_grad(x[idx], x_plus[idx])
            if self.reducer is not None:
                self.reducer.all_reduce_mean()
            self.optimizer.step()
            losses.append(total.clone())
        return torch.stack(losses).mean().item()

    def fit(self, x: Tensor, x_plus: Tensor, epochs: int) -> TrainLog:
        return TrainLog([self.train_epoch(x, x_plus) for _ in range(epochs)])
