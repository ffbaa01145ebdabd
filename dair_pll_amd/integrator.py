"""``Integrator`` / ``VelocityIntegrator`` with the reference's calling convention
(``dair_pll/integrator.py:40-99, 149-166``).

``simulate`` has a fast path: when the integrator belongs to a
:class:`~dair_pll_amd.system.MultibodyLearnableSystem` (``fused_simulate`` set), the whole rollout is
one ``dpll_simulate`` kernel launch instead of ``steps`` Python iterations.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
from torch import Tensor
from torch.nn import Module

PartialStepCallback = Callable[[Tensor, Tensor], Tuple[Tensor, Tensor]]


class Integrator(Module):
    def __init__(self, space, partial_step_callback: PartialStepCallback, dt: float) -> None:
        super().__init__()
        self.partial_step_callback = partial_step_callback
        self.space = space
        self.dt = dt
        self.out_size = type(self).calc_out_size(space)
        self.fused_simulate: Optional[Callable[[Tensor, int], Tensor]] = None

    def partial_step(self, x: Tensor, carry: Tensor) -> Tuple[Tensor, Tensor]:
        assert self.partial_step_callback is not None
        return self.partial_step_callback(x, carry)

    def simulate(self, x_0: Tensor, carry_0: Tensor, steps: int) -> Tuple[Tensor, Tensor]:
        """``(*, n_x)`` initial states -> ``(*, steps + 1, n_x)`` trajectory (and tiled carry)."""
        assert steps >= 0
        assert x_0.shape[-1] == self.space.n_x
        carry_trajectory = carry_0.unsqueeze(-2).repeat((1,) * (carry_0.dim() - 1) + (steps + 1, 1))
        if self.fused_simulate is not None:
            return self.fused_simulate(x_0, steps), carry_trajectory
        x_trajectory = x_0.unsqueeze(-2).repeat((1,) * (x_0.dim() - 1) + (steps + 1, 1))
        x, carry = x_0, carry_0
        for step in range(steps):
            x, carry = self.step(x, carry)
            x_trajectory[..., step + 1, :] = x
            carry_trajectory[..., step + 1, :] = carry
        return x_trajectory, carry_trajectory

    def step(self, x: Tensor, carry: Tensor) -> Tuple[Tensor, Tensor]:
        raise NotImplementedError

    @staticmethod
    def calc_out_size(space) -> int:
        return space.n_x


class VelocityIntegrator(Integrator):
    """``partial_step`` returns the next velocity; the configuration follows by the Lie-group
    Euler step (``dair_pll/integrator.py:149-166``)."""

    def step(self, x: Tensor, carry: Tensor) -> Tuple[Tensor, Tensor]:
        space = self.space
        q = space.q(x)
        v_next, carry = self.partial_step(x, carry)
        return space.x(space.euler_step(q, v_next, self.dt), v_next), carry

    @staticmethod
    def calc_out_size(space) -> int:
        return space.n_v
