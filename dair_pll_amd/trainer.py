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


def load_tosses(path: str) -> List[Tensor]:
    """The trajectories of a toss data set stored as one array file (``states`` (sum T, n_x), ``lengths`` (n,)): the layout
    ``assets/contactnets_cube_tosses.npz`` keeps the reference's 550 real cube tosses in (``assets/contactnets_cube/N.pt``
    there, ``file_utils.py``: one ``(T, 13)`` tensor per toss)."""
    import numpy as np
    data = np.load(path)
    states, lengths = torch.tensor(data['states']), data['lengths']
    assert int(lengths.sum()) == states.shape[0]
    return list(torch.split(states, [int(n) for n in lengths]))


def slice_windows(trajectories: Iterable[Tensor], t_prediction: int) -> Tuple[Tensor, Tensor]:
    """The same slicing with a prediction horizon (``dataset_management.py:43-59`` with ``t_skip=0,
    t_history=1``): ``x_past (S, 1, n_x)`` and ``x_future (S, t_prediction, n_x)`` for every start index
    ``i`` in ``[0, T - t_prediction)``."""
    past, future = [], []
    for trajectory in trajectories:
        assert trajectory.dim() == 2 and trajectory.shape[0] > t_prediction >= 1
        for i in range(trajectory.shape[0] - t_prediction):
            past.append(trajectory[i:i + 1])
            future.append(trajectory[i + 1:i + 1 + t_prediction])
    return torch.stack(past), torch.stack(future)


def prediction_loss(system, x_past: Tensor, x_future: Tensor) -> Tensor:
    """``SupervisedLearningExperiment.prediction_loss`` (``experiment.py:292-320``): roll the model out from
    the last state of ``x_past (B, T_0, n_x)`` for ``x_future.shape[-2]`` steps and average the squared
    velocity error over batch, horizon and velocity components.  Differentiable with respect to the system's
    parameters through every step (``dpll_step_backward``: parameter gradient + state adjoint)."""
    steps = x_future.shape[-2]
    carry = torch.zeros(x_past.shape[:-2] + (1,), device=x_past.device)
    predicted, _ = system.simulate(x_past, carry, steps)
    v_predicted = system.space.v(predicted[..., 1:, :])
    v_future = system.space.v(x_future)
    return ((v_future - v_predicted)**2).sum() / v_predicted.numel()


@dataclass
class TrainLog:
    epoch_losses: List[float]


class ContactNetsTrainer:
    """Adam on ``system.parameters()`` with the fused ContactNets loss
    (``lr=1e-3``, ``weight_decay`` as in ``examples/contactnets_simple.py:78-86``)."""

    def __init__(self, system, lr: float = 1e-3, weight_decay: float = 0.0, batch_size: int = 4096,
                 seed: int = 0, loss: str = 'contactnets', use_graph: bool = False, fused_adam: bool = False) -> None:
        """``use_graph``: one training step (packing the parameters, fused loss + gradients, gradient exchange, Adam)
        is captured once as a hipGraph and replayed per full batch -- the step is a dozen launch-bound small
        kernels, most of them the optimizer's.  ``fused_adam`` (box geometry, cube / elbow topology): the Adam update is
        done by the finalize kernel of the loss launch (``dpll_contactnets_train_step``): a training step is two
        launches; ``optimizer`` is then a :class:`FusedAdamState`."""
        assert loss in ('contactnets', 'prediction')  # MultibodyLosses of drake_experiment.py:47-52
        self.loss = loss
        self.system = system
        self.batch_size = batch_size
        self.use_graph = use_graph and loss == 'contactnets'
        self.fused_adam = fused_adam and loss == 'contactnets'
        if self.fused_adam:
            from .system import FusedAdamState
            self.optimizer = FusedAdamState(lr=lr, weight_decay=weight_decay)
        else:
            self.optimizer = torch.optim.Adam(system.parameters(), lr=lr, weight_decay=weight_decay,
                                              capturable=self.use_graph)
        self._graph = None
        self._static = None
        # how the steps reached the device: `enqueued` = training steps whose kernels this process enqueued one by one
        # (eager batches, the warm-up and the capture of a graph), `replayed` = steps that were one hipGraph replay
        self.step_counts = {'enqueued': 0, 'replayed': 0}
        self.generator = torch.Generator().manual_seed(seed)
        self.health_every = 16  # steps between looks at the peer exchange's error word (each look synchronises)
        self.reducer: Optional[GradientAllReduce] = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1:
            # (the fused update rides in the finalize kernel, and so must the exchange: the peer transport)
            self.reducer = GradientAllReduce(system, transport='peer' if self.fused_adam else 'auto')

    def train_epoch(self, x: Tensor, x_plus: Tensor) -> float:
        """One pass over the (device-resident) pairs in shuffled mini-batches; returns the mean of the
        batch losses (``experiment.py:348-363``).  With a process group every rank must hold the SAME
        pairs: each batch is sharded by contiguous rows and gradients are all-reduced.
        ``loss='prediction'``: ``x`` is ``x_past (S, T_0, n_x)``, ``x_plus`` is ``x_future (S, T, n_x)``
        (:func:`slice_windows`) and the batch loss is :func:`prediction_loss` through autograd."""
        if self.loss == 'prediction':
            return self._train_epoch_prediction(x, x_plus)
        n = x.shape[0]
        order = torch.randperm(n, generator=self.generator).to(x.device)
        losses = []
        for start in range(0, n, self.batch_size):
            idx = order[start:start + self.batch_size]
            if self.reducer is not None:
                lo, hi = shard_bounds(idx.numel(), torch.distributed.get_rank(), self.reducer.world)
                self.system.global_batch = idx.numel()
                idx = idx[lo:hi]
            if self.reducer is not None and len(losses) % self.health_every == self.health_every - 1:
                # a timed-out peer exchange turns the gradient rows into NaN and the optimizer would apply them: look at
                # the error word every few steps (it synchronises), so that the abort leaves parameters at most
                # `health_every` steps old instead of an epoch of NaN updates
                self.reducer.check_healthy()
            if self.use_graph and idx.numel() == self._graph_rows():
                losses.append(self._graph_step(x, x_plus, idx))
                continue
            # a ragged tail batch may leave some ranks an empty shard: they launch a zero row and still take part in
            # the exchange (dpll_contactnets_loss accepts batch = 0 together with a gradient buffer)
            losses.append(self._step_body(x[idx], x_plus[idx]).clone())
        if self.reducer is not None:
            self.reducer.check_healthy()  # once per epoch: a timed-out exchange must not train on silently
        return torch.stack(losses).mean().item()

    # ---- hipGraph replay of the training step ----------------------------------------------------------
    def _graph_rows(self) -> int:
        """Rows of a full batch on this rank (the captured step has static shapes; a ragged tail runs eagerly)."""
        if self.reducer is None:
            return self.batch_size
        lo, hi = shard_bounds(self.batch_size, torch.distributed.get_rank(), self.reducer.world)
        return hi - lo

    def _step_body(self, x: Tensor, x_plus: Tensor) -> Tensor:
        self.step_counts['enqueued'] += 1
        if self.fused_adam:
            return self.system.contactnets_train_step(x, x_plus, self.optimizer)
        total = self.system.contactnets_loss_and_grad(x, x_plus)
        if self.reducer is not None:
            self.reducer.all_reduce_mean()
        self.optimizer.step()
        return total

    def _graph_step(self, x: Tensor, x_plus: Tensor, idx: Tensor) -> Tensor:
        if self._graph is None:
            if self.reducer is not None and self.reducer.transport != 'peer' and \
                    torch.distributed.get_backend() != 'nccl':
                raise NotImplementedError('graph capture needs a capturable gradient exchange (peer kernel or RCCL)')
            rows = self._graph_rows()
            xs = torch.empty((rows, x.shape[1]), dtype=x.dtype, device=x.device)
            xps = torch.empty_like(xs)
            xs.copy_(x[idx])
            xps.copy_(x_plus[idx])
            # the state an eager warm-up changes (parameters, Adam moments and step count) is restored afterwards,
            # so that captured training is step for step the eager training
            params = [p.detach().clone() for p in self.system.parameters()]
            if self.fused_adam:
                self.optimizer.bind(self.system._packed())
                fused_saved = {n: v.clone() for n, v in self.optimizer.state_dict().items()}
                saved_state = {}
            else:
                saved_state = {id(p): {n: v.clone() for n, v in st.items() if torch.is_tensor(v)}
                               for p, st in self.optimizer.state.items()}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._step_body(xs, xps)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                total = self._step_body(xs, xps)
            with torch.no_grad():
                for p, saved in zip(self.system.parameters(), params):
                    p.copy_(saved)
                # optimizer state back to what it was before the warm-up (moments of earlier eager steps, a loaded
                # checkpoint, or zeros when no step had been taken), IN PLACE: the graph holds these tensors' addresses
                if self.fused_adam:
                    for n, v in self.optimizer.state_dict().items():
                        v.copy_(fused_saved[n])
                for k, st in ({} if self.fused_adam else self.optimizer.state).items():
                    for n, v in st.items():
                        if torch.is_tensor(v):
                            before = saved_state.get(id(k), {}).get(n)
                            if before is not None:
                                v.copy_(before)
                            else:
                                v.zero_()
            self._graph, self._static = graph, (xs, xps, total)
        xs, xps, total = self._static
        torch.index_select(x, 0, idx, out=xs)
        torch.index_select(x_plus, 0, idx, out=xps)
        self._graph.replay()
        self.step_counts['replayed'] += 1
        return total.clone()

    def _train_epoch_prediction(self, x_past: Tensor, x_future: Tensor) -> float:
        if self.reducer is not None:
            raise NotImplementedError('prediction-loss training is single process (its gradients live in .grad)')
        order = torch.randperm(x_past.shape[0], generator=self.generator).to(x_past.device)
        losses = []
        for start in range(0, x_past.shape[0], self.batch_size):
            idx = order[start:start + self.batch_size]
            self.optimizer.zero_grad(set_to_none=True)
            loss = prediction_loss(self.system, x_past[idx], x_future[idx])
            loss.backward()
            self.optimizer.step()
            losses.append(loss.detach())
        return torch.stack(losses).mean().item()

    def fit(self, x: Tensor, x_plus: Tensor, epochs: int) -> TrainLog:
        return TrainLog([self.train_epoch(x, x_plus) for _ in range(epochs)])
